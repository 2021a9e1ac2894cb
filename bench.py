"""bench.py — fusion-level DynMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic batch resident in HBM: zero grads, forward (dual
ResNet-34 NonBottleneck1D encoders, SE fusion, global gate with soft DiffSoftmax gates tau=1, PPM, ESANet
decoder, 4 training outputs), weighted multi-scale CE + FLOP regulariser, full backward, [gradient all-reduce],
fused SGD-Nesterov update — batch 32/GPU at 480x640 (BASELINE.json configs[2], the configuration `metric` is
quoted on).  N>1 = pure data parallel, weak scaling, gradient buckets all-reduced over RCCL while backward is
still running.  Prints ONE JSON line on rank 0; besides the contract's fields it carries
  roofline      dominant implicit-GEMM kernel vs the fp32 MFMA peak (HIP events on the launch stream),
  cpu_baseline  the CPU oracle (a port of the reference's PyTorch CPU path) on this host's cores,
  parity        the HIP step vs that same oracle run (logits, losses, gradient cosine) — same inputs,
  extra         configs[1] (fwd-only, batch 16, gate forced on) and configs[3]-style hard-gate lines measured
                in the same process.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dynmm_amd import dp, engine, ops, synth                # noqa: E402
from dynmm_amd.nn.net import SkipGateESANet                 # noqa: E402
from dynmm_amd.nn.net_skip import SkipESANet                # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
GFLOP_PER_IMG_FWD_BWD = {'P': 222.98, 'S': 300.8}   # BASELINE.md §2 (conv MACs x2, fwd+bwd)
GFLOP_PER_IMG_FWD = {'P': 74.67, 'S': 100.62}
TRAFFIC_NOTE = ('HBM bytes per launch from profiles/pmc_dominant_kernel.json (separate rocprofv3 --pmc passes for FETCH_SIZE '
                'and WRITE_SIZE): fetch_factor x FETCH_SIZE + WRITE_SIZE, fetch_factor = 2 (the guide\'s gfx950 correction) '
                'for kernels that stream 16 B/lane (conv_wgrad_v4, conv_igemm_v5), 1 = raw for the 4 B/lane gathers of the implicit-GEMM '
                'kernels (uncalibrated width: lower bound).  For the split weight gradient the figure includes its slab '
                'writes; the slab reduction kernel that follows is listed separately in the JSON.')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=None, help='ranks (= GPUs) of this node; default: WORLD_SIZE or 1')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--config', default='P', choices=['P', 'S'])
    ap.add_argument('--model', default='gate', choices=['gate', 'skip'],
                    help="gate: SkipGateESANet (global gate, the north-star line); skip: SkipESANet (per-stage Gumbel "
                         "gates, block_rule 2222) — a side measurement, SURVEY.md §8f-3")
    ap.add_argument('--mode', default='train', choices=['train', 'fwd'],
                    help="train: fwd+bwd+update (configs[2]; with --hard: configs[3]); fwd: eval forward (configs[1])")
    ap.add_argument('--hard', action='store_true',
                    help='--mode train: hard one-hot gates with a FIXED synthetic branch distribution (--branches), '
                         'the per-GPU part of BASELINE configs[3]')
    ap.add_argument('--branches', default='all4', choices=['all4', 'uniform', 'all0'],
                    help='per-sample gate branch (hard one-hot): all4 = every sample fuses at every stage, uniform = '
                         'k = n %% 5, all0 = every sample skips depth after the stem.  With compaction depth stage j '
                         'runs on the samples with k >= j only')
    ap.add_argument('--compact', action='store_true',
                    help='--mode train --hard: gate-decision compaction in TRAINING (approximate: BatchNorm statistics '
                         'of depth stage j are taken over the samples that run it; see DESIGN.md §K16)')
    ap.add_argument('--no-compact', action='store_true', help='--mode fwd: disable K16 compaction (dense reference semantics)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step as one hipGraph.  Default is eager multi-stream launches: the step is '
                         'GPU-bound at batch 32 (eager == graph on one stream) and the 3-stream schedule '
                         '(RGB encoder | depth encoder | weight gradients) overlaps better un-captured')
    ap.add_argument('--single-stream', action='store_true', help='disable the depth-encoder and wgrad side streams')
    ap.add_argument('--dp-exchange', default='wgrad', choices=['wgrad', 'depth', 'comm'],
                    help="N > 1 over RCCL: the stream the gradient-bucket all-reduces are enqueued on (dp.GradBucketReducer): the last "
                         "weight-gradient stream (default), the depth-encoder stream, or asynchronous collectives from a "
                         "communication stream of their own")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=4, help='batch of the thread sweep + parity sample')
    ap.add_argument('--no-cpu-full-batch', dest='cpu_full_batch', action='store_false',
                    help='skip the single full-batch oracle step (report the small-sample rate)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the secondary lines (configs[1] fwd-only, hard-gate)')
    ap.add_argument('--affect-only', action='store_true',
                    help='print the ModalityDynMM (configs[4]) secondary line alone: what the main run starts as a child process')
    return ap.parse_args()


def make_model(cfg, h, w, device, kind='gate'):
    block = 'NonBottleneck1D' if cfg == 'P' else 'BasicBlock'
    if kind == 'skip':
        m = SkipESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
                       encoder_block=block, nr_decoder_blocks=[3, 3, 3], fuse_depth_in_rgb_encoder='SE-add',
                       block_rule=[2, 2, 2, 2])
    else:
        m = SkipGateESANet(height=h, width=w, encoder_block=block, fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), seed=0)
    return m.to(device)


def make_batch(n, h, w, device, seed):
    """Synthetic batch generated on the host (so the CPU oracle can see the same values) and made resident."""
    g = torch.Generator().manual_seed(seed)
    rgb = torch.randn(n, 3, h, w, generator=g)
    depth = torch.randn(n, 1, h, w, generator=g)
    labels = [torch.randint(0, 41, (n, h // s, w // s), generator=g, dtype=torch.uint8) for s in (1, 8, 16, 32)]
    return rgb.to(device), depth.to(device), [t.to(device) for t in labels]


def branches_for(kind, n):
    return [(i % 5) if kind == 'uniform' else (4 if kind == 'all4' else 0) for i in range(n)]


# ---------------------------------------------------------------------------------------------------
# CPU baseline (the oracle on this host's cores) + parity of the HIP step against that same run
# ---------------------------------------------------------------------------------------------------
def host_cpu():
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name') and model == 'unknown':
                model = ln.split(':', 1)[1].strip()
            elif ln.startswith('physical id'):
                phys = ln.split(':', 1)[1].strip()
            elif ln.startswith('core id'):
                core = ln.split(':', 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1)), (os.cpu_count() or 1)


def cpu_allowance():
    """What this process may actually use: the scheduler affinity mask and the cgroup CPU quota (v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`) — a container can see 128 cores in /proc/cpuinfo and be throttled to 16."""
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count() or 1
    cpu_max, quota = None, None
    try:
        cpu_max = open('/sys/fs/cgroup/cpu.max').read().strip()
        q, per = cpu_max.split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            cpu_max = f'{q} {per}'
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    eff = aff if quota is None else min(aff, max(1, int(quota + 0.999)))
    return {'affinity': aff, 'cpu_max': cpu_max, 'quota_cpus': quota, 'effective': eff}


def cpu_baseline_and_parity(args, device):
    """Times the oracle's step on a bounded sample (batch args.cpu_batch of the same workload) for several thread
    counts and keeps the best; the last oracle run is also the checker for one HIP step on identical inputs."""
    from oracle import dynmm_oracle as O
    n = args.cpu_batch
    cfg = O.Config(encoder_block='NonBottleneck1D' if args.config == 'P' else 'BasicBlock', fuse='SE-add')
    m = make_model(args.config, args.height, args.width, 'cpu', args.model)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    rgb, depth, labels = make_batch(n, args.height, args.width, 'cpu', 4321)
    cw = torch.linspace(0.5, 2.0, 40)
    noise = [torch.empty(n, 2).exponential_() for _ in range(4)]
    train = args.mode == 'train'
    keep = {}

    def one(batch=None):
        """one oracle step on `batch` (default: the batch-n sample whose results the parity check uses)"""
        rgb_, depth_, labels_, noise_, out = (rgb, depth, labels, noise, keep) if batch is None else batch + ({},)
        sd_run = {k: (v.detach().clone() if 'running_' in k or not v.dtype.is_floating_point else v) for k, v in sd.items()}
        if not train:
            with torch.no_grad():
                if args.model == 'skip':
                    out['out'] = O.forward_skip(sd_run, rgb_, depth_, cfg, noise_, test=True)
                else:
                    out['out'] = O.forward(sd_run, rgb_, depth_, cfg, test=True, baseline=True)
            return
        for p in params.values():
            p.grad = None
        if args.model == 'skip':
            outs, lf = O.forward_skip(sd_run, rgb_, depth_, cfg, noise_, training=True), torch.zeros(())
        else:
            outs, lf = O.forward(sd_run, rgb_, depth_, cfg, training=True, temp=1.0)
        losses = O.cross_entropy_2d(outs, labels_, cw)
        total = sum(losses) + torch.clamp(lf, min=0.0)
        total.backward()
        out.update(out=outs[0].detach(), losses=torch.stack([l.detach() for l in losses]), lf=lf.detach(),
                   total=total.detach())

    model_name, phys, logical = host_cpu()
    avail = cpu_allowance()
    default_threads = torch.get_num_threads()
    cap = min(logical, avail['affinity'])
    # widths from "every physical core" down to 4; the cgroup quota (if any) is a candidate of its own: on a host that
    # shows 128 cores to a container allowed ~16 CPUs' worth of time, more threads than the quota only add contention
    cands = {phys, 64, 32, 16, 8, 4, default_threads}
    if avail['quota_cpus'] is not None:
        cands.add(max(1, int(round(avail['quota_cpus']))))
    cands = sorted({t for t in cands if 1 <= t <= cap}, reverse=True)
    sweep, best = {}, None
    one()                                              # warm-up (allocator, oneDNN primitive caches)
    for t in cands:
        torch.set_num_threads(t)
        one()                                          # settle the thread pool at this width
        t0 = time.perf_counter()
        one()
        dt = time.perf_counter() - t0
        sweep[t] = round(n / dt, 4)
        if best is None or dt < best[1]:
            best = (t, dt)
        elif t < best[0] and dt > 1.25 * best[1]:
            break                                      # the curve has turned: narrower widths only get slower
    torch.set_num_threads(best[0])

    def median_rate(batch, nb, max_s):
        """1 warm-up + >= 3 timed steps, median (fewer timed steps only if one step exceeds max_s / 4)"""
        t0 = time.perf_counter()
        one(batch)                                     # warm-up at THIS batch size: first touch of its activations
        warm = time.perf_counter() - t0
        reps = 3 if warm * 4 <= max_s else 1
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            one(batch)
            ts.append(time.perf_counter() - t0)
        return round(nb / sorted(ts)[len(ts) // 2], 4), reps

    small, small_reps = median_rate(None, n, 60.0)
    # the workload's own batch at the winning width; the line reports the BEST rate the host reaches over the two samples
    full_n, value, sample_n, reps_used, why, full_rate = args.batch if train else 16, small, n, small_reps, None, None
    if args.cpu_full_batch and full_n > n:
        try:
            gb = torch.Generator().manual_seed(99)
            big = make_batch(full_n, args.height, args.width, 'cpu', 4322) + \
                ([torch.empty(full_n, 2).exponential_(generator=gb) for _ in range(4)],)
            full_rate, full_reps = median_rate(big, full_n, 120.0)
            if full_rate > value:
                value, sample_n, reps_used = full_rate, full_n, full_reps
            del big
        except (RuntimeError, MemoryError) as e:       # host RAM: a batch-32 training step keeps ~45 GB of activations
            why = f'batch {full_n} step failed on this host ({type(e).__name__}); batch {n} figure reported'
    one()                                              # the parity reference below is THIS run (batch n, winning width)
    torch.set_num_threads(default_threads)
    cpu = {'value': value, 'unit': 'images/s', 'cores': best[0], 'kind': 'port',
           'cpu_model': model_name, 'physical_cores': phys, 'logical_cpus': logical,
           'cores_available': avail['effective'], 'sched_affinity': avail['affinity'], 'cgroup_cpu_max': avail['cpu_max'],
           'batch': sample_n, f'batch{n}_images_per_s': small,
           f'batch{full_n}_images_per_s': full_rate,
           'threads_sweep_images_per_s': {str(k): v for k, v in sweep.items()},
           'sample': f'oracle (PyTorch CPU fp32): {args.height}x{args.width} {args.mode} steps (fwd + weighted 4-scale CE + '
                     f'flop loss + bwd) at batch {sample_n} with {best[0]} threads, 1 warm-up + {reps_used} timed, median; the best '
                     f'of the batch-{n} and batch-{full_n} rates is reported; thread count chosen by a sweep {list(sweep)} '
                     f'over batch-{n} steps; the host allows {avail["effective"]} CPUs (affinity {avail["affinity"]}, cgroup '
                     f'cpu.max {avail["cpu_max"]})' + (f'; {why}' if why else '')}

    # ---- parity: the HIP path on the same inputs, against the oracle run above ----
    parity = None
    if args.model == 'gate':
        mh = make_model(args.config, args.height, args.width, device, args.model)
        if train:
            mh.train()
            mh.temp, mh.hard_gate = 1.0, False
            st = engine.TrainStep(mh, cw.numpy(), lr=0.0, loss_ratio=1.0, flop_budget=0.0, multi_stream=not args.single_stream)
            got = {}
            fwd = mh.forward

            def spy(*a, **k):
                r = fwd(*a, **k)
                got['out'] = r[0][0].detach()
                return r
            mh.forward = spy
            st._body(rgb.to(device), depth.to(device), [t.to(device) for t in labels])
            torch.cuda.synchronize()
            hp = dict(mh.named_parameters())
            names = [k for k in params if params[k].grad is not None and k in hp]
            a = torch.cat([hp[k].grad.detach().cpu().double().flatten() for k in names])
            b = torch.cat([params[k].grad.double().flatten() for k in names])
            parity = {
                'logits_rel': float(((got['out'].cpu() - keep['out']).abs().max() / keep['out'].abs().max()).item()),
                'loss_abs': float((st.last['losses'].cpu() - keep['losses']).abs().max().item()),
                'loss_flop_abs': float(abs(st.last['loss_flop'].item() - keep['lf'].item())),
                'total_rel': float(abs(st.last['total'].item() - keep['total'].item()) / abs(keep['total'].item())),
                'grad_cos': float(torch.nn.functional.cosine_similarity(a, b, dim=0).item()),
                'grad_norm_ratio': float((a.norm() / b.norm()).item()),
                'batch': n,
                'note': 'HIP TrainStep body vs the fp32 CPU oracle (the cpu_baseline run) on identical inputs/weights; '
                        'fp32 gradients of this net carry ~1e-2 conditioning noise (fp32 vs fp64 oracle, DESIGN.md §1), '
                        'per-tensor bars are in tests/test_hip_model.py'}
            del st
        else:
            mh.eval()
            mh.baseline = True
            with torch.no_grad():
                out = mh(rgb.to(device), depth.to(device), test=True)
            parity = {'logits_rel': float(((out.cpu() - keep['out']).abs().max() / keep['out'].abs().max()).item()),
                      'batch': n}
        del mh
        torch.cuda.empty_cache()
    return cpu, parity


# ---------------------------------------------------------------------------------------------------
# per-kernel timing (roofline leg)
# ---------------------------------------------------------------------------------------------------
def kernel_timing(step_fn, model):
    """One instrumented EAGER step: HIP events around every implicit-GEMM launch on the stream it is
    launched on.  Returns per-kernel-variant totals (launches, ms, algorithmic GFLOP)."""
    # isolated per-kernel durations: the instrumented pass runs on ONE stream (in the timed region the
    # RGB / depth / wgrad streams overlap, which inflates every individual kernel's wall duration)
    torch.cuda.synchronize()
    ops.PROFILE = []
    step_fn()
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    agg, shapes = {}, {}
    for name, flops, e0, e1, shape, (kind, extra) in rec:
        ms = e0.elapsed_time(e1)
        n_, ci, h, w, co, kh, kw, sh, sw, nprob = shape
        # algorithmic bytes of the launch: every operand once (input + output/gradient tensor + weights), fp32, plus the
        # epilogue's own operands — the ReLU-mask source and the accumulated residual gradient of an input-gradient launch,
        # the residual of a forward launch — each a tensor of the launch's OUTPUT size;
        # nprob > 1: a grouped weight-gradient launch (dynmm_conv2d_wgrad_group) of that many same-shape convolutions
        ho, wo = -(-h // sh), -(-w // sw)
        out_elems = n_ * ci * h * w if kind == 'dgrad' else n_ * co * ho * wo
        abytes = 4.0 * nprob * (n_ * ci * h * w + n_ * co * ho * wo + co * ci * kh * kw) + 4.0 * extra * out_elems
        for d, key in ((agg, name), (shapes, (name, shape))):
            a = d.setdefault(key, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += flops
            a[3] += abytes
    if os.environ.get('DYNMM_BENCH_SHAPES'):
        with open(os.environ['DYNMM_BENCH_SHAPES'], 'w') as f:
            f.write('kernel | N,Ci,H,W,Co,KH,KW,SH,SW | launches | total ms | avg us | TFLOP/s\n')
            for (name, shape), (n, ms, fl, _) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                f.write(f'{name} | {shape} | {n} | {ms:.3f} | {1000 * ms / n:.1f} | {fl / (ms * 1e-3) / 1e12:.1f}\n')
    return agg


def executed_fraction(name):
    """matrix-core FLOP executed per algorithmic (direct-convolution) FLOP of a launch"""
    if 's2' in name:                 # stride 2 — polyphase input gradients on the pair kernel, conv_wgrad_s2: no Winograd arithmetic
        return 1.0
    if 'wino2d' in name:
        return 4.0 / 9.0
    if 'wino43' in name:
        return 0.5
    if 'wino' in name or name.startswith('conv_wgrad_v6'):
        return 2.0 / 3.0
    return 1.0


def kernel_instance(name):
    """The compiled kernel (template instance) a launch label runs on: the Winograd kernels serve every channel count from one
    instance per tap axis (1x3 / 3x1: conv_wino.hip, conv_wino43.hip) or one per direction (3x3: conv_wino2d.hip) — the unit
    rocprofv3 reports and the roofline is quoted for."""
    import re
    m = re.match(r'conv_wino2d_(fwd|dgrad)<co\d+,3x3>', name)
    if m:
        return f'conv_wino2d_{m.group(1)}<3x3>'
    m = re.match(r'conv_wino(43)?_(fwd|dgrad)<co\d+,(\d)x(\d)(s2)?>', name)
    if not m:
        return name
    f43, kind, kh, kw, s2 = m.groups()
    return f"conv_wino{f43 or ''}_{kind}<{'vertical' if kw == '1' else 'horizontal'}{',s2' if s2 else ''}>"


def roofline_of(agg):
    if not agg:
        return None
    inst = {}
    for k, v in agg.items():
        a = inst.setdefault(kernel_instance(k), [0, 0.0, 0.0, 0.0, []])
        for i in range(4):
            a[i] += v[i]
        a[4].append(k)
    name, (launches, ms, flops, abytes, members) = max(inst.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM traffic per launch: rocprofv3 PMC record of the SAME step (profiles/pmc_dominant_kernel.json, written by
    # profiles/r03_recipe.sh).  It is a like-for-like figure only if it was taken over the same launch set: same kernel
    # label, same number of launches per step as this run's instrumented step — otherwise it is withheld.
    traffic, traffic_why = None, None
    pmc = os.path.join(ROOT, 'profiles', 'pmc_dominant_kernel.json')
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(name)
            if rec is None:
                traffic_why = f'no PMC record for {name}'
            elif abs(rec.get('launches_per_step', -1) - launches) > 0.01:
                traffic_why = (f"PMC record covers {rec.get('launches_per_step')} launches per step, this step has {launches}: "
                               'not the same launch set')
            else:
                traffic = rec.get('hbm_bytes_per_launch')
        except Exception as e:
            traffic_why = f'unreadable PMC record: {e}'
    share = executed_fraction(name)
    return {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
            # a Winograd kernel executes `share` of its algorithmic FLOP, so the ceiling of its ALGORITHMIC rate is peak / share
            # (F(2,3) 236, F(4,3) 315, F(2x2,3x3) 354 TF/s); frac = achieved / attainable = the share of the fp32 matrix peak the
            # launch keeps busy — never above 1.  algorithmic_frac = achieved / peak is kept beside it.
            'attainable': round(FP32_MFMA_PEAK_TFLOPS / share, 1),
            'unit': 'TFLOP/s', 'frac': round(achieved * share / FP32_MFMA_PEAK_TFLOPS, 4),
            'algorithmic_frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'executed_share': round(share, 4), 'traffic': traffic,
            'traffic_note': TRAFFIC_NOTE if traffic is not None else traffic_why,
            'kernel': name, 'launch_labels': sorted(members), 'launches_per_step': launches,
            'avg_launch_us': round(1000.0 * ms / launches, 2),
            'algorithmic_gflop_per_launch': round(flops / launches / 1e9, 3),
            'executed_gflop_per_launch': round(flops / launches / 1e9 * executed_fraction(name), 3),
            'executed_frac': round(achieved * executed_fraction(name) / FP32_MFMA_PEAK_TFLOPS, 4),
            'algorithmic_bytes_per_launch': round(abytes / launches),
            'note': ('achieved = ALGORITHMIC (direct-convolution) FLOP / time (SURVEY 8d).  conv_wino_* and the three-tap weight '
                     'gradients (conv_wgrad_v6<..,1x3> in the Winograd form, conv_wgrad_v6<..,3x1> = conv_wgrad_wino_vt) execute '
                     '2/3 of their algorithmic FLOP on the matrix cores (1-D Winograd F(2,3), fp32), conv_wino43_* 1/2 (F(4,3)), '
                     'conv_wino2d_* 4/9 (F(2x2,3x3)): attainable = peak / that share, frac = achieved / attainable (= executed_frac, '
                     'the share of the fp32 MFMA peak the launch keeps busy); algorithmic_frac = achieved / peak can exceed what '
                     'a direct kernel could reach'),
            'all_igemm_kernels': {k: {'launches': v[0], 'ms': round(v[1], 3),
                                      'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                                  for k, v in sorted(agg.items())}}


def dependent_kernel_interval_us(device, n=2000):
    """Box diagnostic: the time one DEPENDENT, empty kernel costs on a stream (n one-element adds back to back): dispatch + completion
    latency of the box, the floor under each of the step's ~890 launches.  The same tree measured 62.6 ... 65.2 ms per step on boxes
    whose isolated per-kernel rates were identical; this figure travels with the line so a reader can tell."""
    x = torch.zeros(1, device=device)
    for _ in range(200):
        x.add_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1.0)
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e6, 2)


def timed(step, steps, warmup, world, device):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed.per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        timed.per_rank = [float(x.item()) for x in every]
        elapsed = max(timed.per_rank)                 # MAX over ranks
    return elapsed


def timed_local(step, steps, warmup):
    """Every rank's own loop time WITHOUT the closing barrier (the spread between ranks that the barrier hides:
    hard-gate compaction gives ranks different amounts of work)."""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


# ---------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------
def train_workload(args, device, rank, world, hard=False, branches='all4', compact=False):
    """(step callable, TrainStep, model).  The step is the product's own engine.TrainStep."""
    model = make_model(args.config, args.height, args.width, device, args.model)
    dp.broadcast_parameters(model)
    rgb, depth, labels = make_batch(args.batch, args.height, args.width, device, 1234 + rank)
    cw = np.linspace(0.5, 2.0, 40)
    model.train()
    model.hard_gate, model.temp = bool(hard), 1.0
    if hard and args.model == 'gate':
        # fixed synthetic branch distribution: the gate network is evaluated and trained, its hard
        # decision is replaced by the given branch per sample
        model.branch_override = branches_for(branches, args.batch)
        model.compact_train = bool(compact)
    ts = engine.TrainStep(model, cw, lr=1e-4, momentum=0.9, weight_decay=1e-4, loss_ratio=1.0, flop_budget=0.0,
                          use_graph=args.graph and not hard, multi_stream=not args.single_stream, overlap=True,
                          exchange=args.dp_exchange)

    def step():
        ts(rgb, depth, labels)

    def body_only():
        """forward + loss + backward of THIS rank only — no collective, no update: what the per-kernel timing pass runs
        on rank 0 while the other ranks wait at the final barrier (a collective here would dead-lock the job)."""
        ts.reducer.active = False
        try:
            ts._body(rgb, depth, [t if t.dtype == torch.uint8 else t.to(torch.uint8) for t in labels])
        finally:
            ts.reducer.active = True
    step.body_only = body_only
    return step, ts, model


def fwd_workload(args, device, batch, branches='all4', compact=True, graph=False):
    model = make_model(args.config, args.height, args.width, device, args.model)
    rgb, depth, _ = make_batch(batch, args.height, args.width, device, 1234)
    model.eval()
    model.compact = compact
    model.dual_stream = not args.single_stream      # depth-encoder stages on a second HIP stream
    if args.model == 'gate':
        if branches == 'all4':
            model.baseline = True                 # configs[1]: static fuse, gate forced on
        else:
            model.ini_stage = True
            model.ini_branches = branches_for(branches, batch)

    infer = engine.InferStep(model, policy='auto') if (graph and args.model == 'gate') else None

    def step():
        if infer is not None:
            return infer(rgb, depth)                      # hipGraph replays (engine.InferStep)
        with torch.no_grad():
            return model(rgb, depth, test=True)
    step.infer = infer
    return step, model


def measure_fwd(args, device, batch, branches, compact, steps, warmup, with_kernels=True, graph=True):
    """graph: the forward as hipGraph replays (engine.InferStep) — the deployment-side default since round 6; the eager
    rate (one launch per kernel from Python) is reported beside it as `eager`."""
    step, model = fwd_workload(args, device, batch, branches, compact, graph=False)
    for _ in range(2):
        step()
    el_eager = timed(step, steps, warmup, 1, device)
    el, launch, auto = el_eager, 'eager', None
    if graph and args.model == 'gate':
        gstep, gmodel = fwd_workload(args, device, batch, branches, compact, graph=True)
        for _ in range(3):
            gstep()
        el = timed(gstep, steps, warmup, 1, device)
        launch = gstep.infer.launch
        auto = getattr(gstep.infer, 'auto_timing', None)
        del gstep, gmodel
    val = batch * steps / el
    out = {'metric': 'images/sec fwd-only, 480x640 RGB-D', 'value': round(val, 2), 'unit': 'images/s',
           'ms_per_step': round(1000 * el / steps, 3), 'launch': launch, 'launch_policy': 'auto (the faster of hipGraph replay and eager launches on this host)',
           'auto_timing': auto if (graph and args.model == 'gate') else None,
           'eager': {'value': round(batch * steps / el_eager, 2), 'ms_per_step': round(1000 * el_eager / steps, 3)},
           'batch': batch, 'branches': branches, 'compaction': bool(compact),
           'stage_batch': getattr(model, 'last_stage_batch', None),
           'model_tflops': round(val * GFLOP_PER_IMG_FWD[args.config] / 1e3, 2) if branches == 'all4' else None}
    if with_kernels:
        saved = model.dual_stream
        model.dual_stream = False
        r = roofline_of(kernel_timing(step, model))
        model.dual_stream = saved
        if r:
            r.pop('all_igemm_kernels', None)
            r['traffic'] = r['traffic_note'] = None
            out['roofline'] = r
    del model
    torch.cuda.empty_cache()
    return out


def measure_affect(device, steps, warmup=3, batch=128, T=50):
    """BASELINE configs[4] (per GPU): ModalityDynMM CMU-MOSEI DynMMNetV2 — text-transformer expert + 3-modality
    late-fusion transformer expert + transformer gate — one optimisation step (fwd, L1 + gate regulariser, bwd,
    clip_grad_norm_, AdamW) at batch 128 on synthetic MOSEI-shaped features.  Parity of this path is UNPINNED
    (MultiBench is not vendored by the reference): tests/test_affect.py checks it against a self-written oracle."""
    from dynmm_amd.nn import affect as A
    torch.manual_seed(0)
    model = A.DynMMNetV2(1.0, False, freeze=False).to(device)
    g = torch.Generator().manual_seed(7)
    xs = [torch.randn(batch, T, f, generator=g).to(device) for f in (35, 74, 300)]
    inputs = [xs, [torch.full((batch,), T, dtype=torch.long)] * 3]
    y = torch.randn(batch, 1, generator=g).to(device)
    step = A.AffectTrainStep(model, lr=1e-5, weight_decay=1e-4, lossw=0.1, use_graph=True)
    res = {}
    model.train()                                     # dropout p = 0.1 at the 100 sites, as the reference trains
    for name, fn in (('train_step', lambda: step(inputs, y)), ('train_step_no_dropout', lambda: step(inputs, y)),
                     ('forward', None)):
        if name == 'train_step_no_dropout':
            model.eval()                                  # same step with the eval-mode arithmetic (what rounds 1-2 measured)
        if fn is None:
            model.eval()

            def fn():
                with torch.no_grad():
                    return model(inputs)
        for _ in range(2):
            fn()
        el = timed(fn, steps, warmup, 1, device)
        res[name] = {'value': round(batch * steps / el, 1), 'unit': 'samples/s', 'ms_per_step': round(1000 * el / steps, 3)}
    mmac = 135.13226 + 320.03205                      # affect_dyn.py:126 (thop MACs per sample, experts 1 + 2)
    res['forward']['model_tflops'] = round(res['forward']['value'] * 2 * mmac * 1e6 / 1e12, 2)
    res['train_step']['model_tflops'] = round(res['train_step']['value'] * 6 * mmac * 1e6 / 1e12, 2)
    res['train_step']['launch'] = 'hipGraph replay'
    res['train_step']['dropout'] = 'p = 0.1, Philox masks regenerated in the backward, new masks every replay'
    res['train_step_no_dropout']['model_tflops'] = round(res['train_step_no_dropout']['value'] * 6 * mmac * 1e6 / 1e12, 2)
    res['workload'] = ('configs[4] per GPU: ModalityDynMM DynMMNetV2 on CMU-MOSEI-shaped synthetic features '
                       f'(T={T}: visual 35, audio 74, text 300), batch {batch}; experts trainable (freeze=False); '
                       'PARITY UNPINNED (MultiBench not vendored)')
    del step, model
    torch.cuda.empty_cache()
    # BASELINE configs[4] words the workload as "3-expert transformer late-fusion + 3-way gating net" = DynMMNet
    # (affect_dyn.py:31-104: three uni-modal transformer experts, 3-way gate): the same step on that model
    torch.manual_seed(0)
    m3 = A.DynMMNet(1.0, False, freeze=False).to(device)
    step3 = A.AffectTrainStep(m3, lr=1e-5, weight_decay=1e-4, lossw=0.1, use_graph=True)
    m3.train()
    for _ in range(2):
        step3(inputs, y)
    el = timed(lambda: step3(inputs, y), steps, warmup, 1, device)
    m3.eval()

    def fwd3():
        with torch.no_grad():
            return m3(inputs)
    for _ in range(2):
        fwd3()
    elf = timed(fwd3, steps, warmup, 1, device)
    res['dynmmnet_3expert'] = {'train_step': {'value': round(batch * steps / el, 1), 'unit': 'samples/s',
                                              'ms_per_step': round(1000 * el / steps, 3), 'launch': 'hipGraph replay'},
                               'forward': {'value': round(batch * steps / elf, 1), 'unit': 'samples/s',
                                           'ms_per_step': round(1000 * elf / steps, 3)},
                               'workload': 'DynMMNet (affect_dyn.py:31-104): visual / audio / text transformer experts + 3-way '
                                           'gate, experts trainable, dropout p = 0.1; PARITY UNPINNED'}
    del step3, m3
    torch.cuda.empty_cache()
    return res


def measure_affect_isolated(steps):
    """measure_affect() in a process of its own.  It is the one leg that captures and replays hipGraphs (five branch streams,
    ~500 nodes); a fault inside the runtime's capture path is not a Python exception, and a secondary line must never take the
    headline line down with it."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--affect-only', '--steps', str(int(steps))]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            return {'error': f'child exited with {r.returncode}: {(r.stderr or r.stdout)[-300:]}'}
        return json.loads(lines[-1])
    except Exception as e:
        return {'error': f'{type(e).__name__}: {e}'}


def sub_soft(args):
    sub = argparse.Namespace(**vars(args))
    sub.graph = False
    return sub


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: re-run this script as N ranks of one node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and hand its exit code back.  Under a launcher
    (WORLD_SIZE set — the driver's `python -m torch.distributed.run … bench.py --gpus N`) this is a no-op."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: the only mode the host driver supports (RCCL)
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(1, args.gpus))))
    return subprocess.call(cmd, env=env)


def rccl_self_description():
    """What the first SCALE record should say about the collective library it ran on: the RCCL version torch was built against /
    loaded, the HIP runtime, the devices, and RCCL's own NCCL_DEBUG=VERSION banner (captured through NCCL_DEBUG_FILE, which
    main() points at a per-process file before the process group is created, so nothing but the JSON line reaches stdout)."""
    info = {'hip': torch.version.hip, 'devices': [torch.cuda.get_device_name(i) for i in range(torch.cuda.device_count())]}
    try:
        info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:                                    # noqa: BLE001 - informational only
        info['rccl_version'] = f'unavailable ({type(e).__name__})'
    path = os.environ.get('NCCL_DEBUG_FILE', '')
    if path and os.path.exists(path):
        try:
            with open(path) as f:
                info['rccl_debug'] = [ln.strip()[:200] for ln in f.read().splitlines() if ln.strip()][:6]
        except OSError:
            pass
    return info


def main():
    args = parse()
    if args.gpus is None:
        args.gpus = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    rank = int(os.environ.get('RANK', '0'))
    if os.environ.get('DYNMM_BENCH_LAUNCH_PROBE'):
        # tests/test_bench_launch.py (no GPU needed): the launcher started `world` ranks and they can talk
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', str(_free_port())), ('RANK', '0'), ('WORLD_SIZE', '1')):
            os.environ.setdefault(k, v)
        dist.init_process_group('gloo')
        ones = torch.ones(1)
        dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({'probe': True, 'n_gpus': world, 'ranks_seen': int(ones.item()), 'steps': args.steps,
                              'omp_num_threads': os.environ.get('OMP_NUM_THREADS'),
                              'master': f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                              'ipc_mode_legacy': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}))
        dist.destroy_process_group()
        return
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)           # before the process group: RCCL binds the communicator to the current device
    device = torch.device('cuda', local_rank)
    if args.affect_only:
        print(json.dumps(measure_affect(device, max(10, args.steps))))
        return
    if world > 1:
        # nccl == RCCL on ROCm.  DYNMM_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a
        # single-GPU box (ranks then share device 0).
        os.environ.setdefault('NCCL_DEBUG', 'VERSION')        # RCCL's version / build banner, into a file (rccl_self_description)
        os.environ.setdefault('NCCL_DEBUG_FILE', f'/tmp/dynmm_rccl_{os.getpid()}.log')
        dist.init_process_group(os.environ.get('DYNMM_DIST_BACKEND', 'nccl'))
    train = args.mode == 'train'

    ts = None
    if train:
        step, ts, model = train_workload(args, device, rank, world, args.hard, args.branches, args.compact)
    else:
        step, model = fwd_workload(args, device, args.batch, args.branches, not args.no_compact)
    for _ in range(2):
        step()
    elapsed = timed(step, args.steps, args.warmup, world, device)
    ms_per_step = 1000.0 * elapsed / args.steps
    value = args.batch * world * args.steps / elapsed

    dp_info = None
    if world > 1:
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                          # every rank of the job answers: the line proves N ranks ran
        per_rank_ms = [1000.0 * t / args.steps for t in timed.per_rank]
        dp_info = {'backend': dist.get_backend(), 'ranks_seen': int(round(ones.item())),
                   'devices_visible': torch.cuda.device_count(), 'collective_library': rccl_self_description(),
                   'ms_per_step_rank_min_mean_max': [round(min(per_rank_ms), 3), round(sum(per_rank_ms) / world, 3),
                                                     round(max(per_rank_ms), 3)]}
        if dp_info['ranks_seen'] != args.gpus:
            raise SystemExit(f"bench.py: {dp_info['ranks_seen']} ranks answered the all-reduce, --gpus {args.gpus}")
    if ts is not None and world > 1:
        red = ts.reducer
        dp_info.update({'buckets': len(red.buckets), 'launched_during_backward': red.launched_in_backward,
                        'bucket_mb': round(4 * max(e - s for s, e in red.buckets) / 2 ** 20, 1)})
        # what the exchange costs: the same step with the reducer switched off (no collective, local gradients), and
        # each rank's own loop time with no barrier at the end (rank skew)
        k2 = max(3, args.steps // 2)
        red.enabled = False
        el_nc = timed(step, k2, 1, world, device)
        red.enabled = True
        local = timed_local(step, k2, 1)
        t = torch.tensor([local], device=device, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        loc = [1000.0 * float(x.item()) / k2 for x in every]
        # the four-stream plan, per rank: how many streams each rank's step enqueued on (ops.stream_census; 4 = the plan)
        cz = torch.tensor([float((ts.census or {}).get('streams', -1))], device=device, dtype=torch.float64)
        every_c = [torch.zeros_like(cz) for _ in range(world)]
        dist.all_gather(every_c, cz)
        rep = ops.stream_plan(device).report
        cl = torch.tensor([1.0 if rep.get('clean') else 0.0, float(rep.get('replaced', 0))], device=device, dtype=torch.float64)
        every_p = [torch.zeros_like(cl) for _ in range(world)]
        dist.all_gather(every_p, cl)
        dp_info.update({'stream_plan_clean_per_rank': [bool(x[0].item()) for x in every_p],
                        'stream_plan_replaced_per_rank': [int(x[1].item()) for x in every_p]})
        dp_info.update({'exchange': red.exchange, 'stream_ordered_collectives': bool(red._stream_ordered),
                        'busy_streams_per_rank': [int(x.item()) for x in every_c],
                        'ms_per_step_no_collectives': round(1000.0 * el_nc / k2, 3),
                        'comm_exposed_ms': round(ms_per_step - 1000.0 * el_nc / k2, 3),
                        'ms_per_step_unbarriered_rank_min_mean_max': [round(min(loc), 3), round(sum(loc) / world, 3),
                                                                      round(max(loc), 3)]})

    census = ts.census if ts is not None else None        # of the timed loop's last step (the instrumented pass below is single-stream)
    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        if ts is not None:
            saved = (ts.multi_stream, getattr(model, 'dual_stream', False), ts.use_graph)
            ts.multi_stream, model.dual_stream, ts.use_graph = False, False, False
            roofline = roofline_of(kernel_timing(step.body_only if world > 1 else step, model))
            ts.multi_stream, model.dual_stream, ts.use_graph = saved
        else:
            saved = getattr(model, 'dual_stream', False)
            model.dual_stream = False
            roofline = roofline_of(kernel_timing(step, model))
            model.dual_stream = saved
    stage_batch = getattr(model, 'last_stage_batch', None)
    del step, ts, model
    torch.cuda.empty_cache()

    extra = None
    if rank == 0 and world == 1 and not args.no_extra and train and not args.hard and args.model == 'gate':
        extra = {}
        # configs[1]: fwd-only, batch 16, static fuse (gate forced on) — measured by the same driver run
        extra['fwd_only'] = measure_fwd(args, device, 16, 'all4', True, max(10, args.steps), 3)
        extra['fwd_only']['workload'] = 'configs[1]: fwd-only eval, batch 16, static fuse (gate forced on), BN folded'
        # configs[3] (per-GPU part): hard gates, fixed uniform branch distribution, with / without compaction
        d = measure_fwd(args, device, 32, 'uniform', False, max(10, args.steps), 3, with_kernels=False)
        c = measure_fwd(args, device, 32, 'uniform', True, max(10, args.steps), 3, with_kernels=False)
        extra['fwd_hard_uniform'] = {'dense': d, 'compacted': c, 'gain': round(c['value'] / d['value'], 4),
                                     'workload': 'fwd-only eval, batch 32, hard one-hot gates, branch k = n % 5 per sample; '
                                                 'compacted = depth stage j runs on the samples with k >= j only (exact)'}
        sub = argparse.Namespace(**vars(args))
        sub.graph = False
        res = {}
        for name, (br, cp) in {'dense_uniform': ('uniform', False), 'compact_uniform': ('uniform', True),
                               'compact_all0': ('all0', True)}.items():
            st2, ts2, m2 = train_workload(sub, device, rank, 1, True, br, cp)
            for _ in range(2):
                st2()
            el = timed(st2, max(5, args.steps // 2), 2, 1, device)
            k = max(5, args.steps // 2)
            res[name] = {'value': round(args.batch * k / el, 2), 'ms_per_step': round(1000 * el / k, 3), 'branches': br,
                         'compaction': cp, 'stage_batch': getattr(m2, 'last_stage_batch', None)}
            del st2, ts2, m2
            torch.cuda.empty_cache()
        res['gain_uniform'] = round(res['compact_uniform']['value'] / res['dense_uniform']['value'], 4)
        res['workload'] = ('configs[3] per GPU: fwd+bwd+update, batch 32, hard one-hot gates with a fixed branch '
                           'distribution; compaction in training is APPROXIMATE (depth-stage BatchNorm statistics over the '
                           'taken subset, straight-through gate gradient from the taken stages only) — DESIGN.md')
        res['unit'] = 'images/s'
        extra['train_hard'] = res
        extra['affect_mosei'] = measure_affect_isolated(max(10, args.steps))
        if args.config == 'P':
            # SURVEY.md section 8 "Config S (secondary)": the BasicBlock (3x3) variant of the same step, 300.8 GFLOP per image
            subs = argparse.Namespace(**vars(args))
            subs.config, subs.graph = 'S', False
            st3, ts3, m3 = train_workload(subs, device, rank, 1)
            for _ in range(2):
                st3()
            k = max(5, args.steps // 2)
            el = timed(st3, k, 2, 1, device)
            val = args.batch * k / el
            saved = (ts3.multi_stream, m3.dual_stream)
            ts3.multi_stream, m3.dual_stream = False, False
            rs = roofline_of(kernel_timing(st3, m3))
            ts3.multi_stream, m3.dual_stream = saved
            extra['config_S'] = {'value': round(val, 2), 'unit': 'images/s', 'ms_per_step': round(1000 * el / k, 3),
                                 'algorithmic_gflop_per_image': GFLOP_PER_IMG_FWD_BWD['S'],
                                 'tflops': round(val * GFLOP_PER_IMG_FWD_BWD['S'] / 1e3, 2),
                                 'frac_of_fp32_mfma_peak': round(val * GFLOP_PER_IMG_FWD_BWD['S'] / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4),
                                 'dominant_kernel': None if not rs else {kk: rs[kk] for kk in (
                                     'kernel', 'launches_per_step', 'avg_launch_us', 'achieved', 'frac', 'executed_frac') if kk in rs},
                                 'workload': 'configs[2] on config S (ResNet-34 BasicBlock encoders): the same step, batch 32, 480x640'}
            del st3, ts3, m3
            torch.cuda.empty_cache()

    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, parity = cpu_baseline_and_parity(args, device)

    if rank == 0:
        gflop_img = GFLOP_PER_IMG_FWD_BWD[args.config] if train else GFLOP_PER_IMG_FWD[args.config]
        dense_work = (not args.hard or args.branches == 'all4' or not args.compact) if train else \
            (args.branches == 'all4' or args.no_compact)
        if args.model == 'gate':
            if train and not args.hard:
                workload = ('configs[2]: zero-grad + fwd + weighted 4-scale CE + FLOP loss + bwd + fused SGD-Nesterov update, '
                            '--dynamic --global-gate soft DiffSoftmax gates tau=1')
            elif train:
                workload = (f'configs[3] per GPU: fwd+bwd+update with hard one-hot gates, branches={args.branches}, '
                            f'compaction={"on (approximate in training)" if args.compact else "off"}')
            else:
                workload = 'configs[1]: fwd-only eval, static fuse (gate forced on)' if args.branches == 'all4' else \
                    f'fwd-only eval, hard gates, branches={args.branches}'
        else:
            workload = ('fwd+bwd --dynamic (per-stage Gumbel-softmax gates, block_rule 2222, soft tau=1), weighted 4-scale CE'
                        if train else 'fwd-only eval test=True (hard Gumbel gates; depth stages run on the still-fusing '
                                      'samples only unless --no-compact)')
        line = {
            'metric': 'images/sec fwd+bwd, 480x640 RGB-D, batch 32/GPU' if train else
                      'images/sec fwd-only, 480x640 RGB-D, gate forced on',
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            # N > 1: the same step with the reducer switched off, beside ms_per_step (what the exchange costs)
            'step_without_collectives_ms': (dp_info or {}).get('ms_per_step_no_collectives'),
            'config': {'workload': workload,
                       'branches': args.branches if (args.hard or not train) else None,
                       'compaction': (args.compact if args.hard else None) if train else (not args.no_compact),
                       'stage_batch': stage_batch,
                       'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                       'height': args.height, 'width': args.width,
                       'parallelism': f'dp{world}', 'launch': 'hipGraph replay' if (args.graph and train and not args.hard) else 'eager',
                       'streams': 1 if args.single_stream else ((2 + ops.WGRAD_STREAMS) if train else 2),
                       'stream_census': census,
                       'stream_plan': ops.stream_plan(device).report,      # do the plan's streams run side by side (pair probe)?
                       'optimizer_in_step': 'fused SGD-Nesterov' if train else None,
                       'winograd': {'passes': ops.WINO, 'input_gradients': ops.WINO_DGRAD, 'weight_gradients': True},
                       'dependent_kernel_interval_us': dependent_kernel_interval_us(device),
                       'dp': dp_info},
            'whole_step': {'net': f'{"SkipGateESANet" if args.model == "gate" else "SkipESANet"} R34-'
                                  f'{"NBt1D" if args.config == "P" else "BasicBlock"} SE-add (config {args.config})',
                           'algorithmic_gflop_per_image': gflop_img,
                           'tflops': round(value * gflop_img / 1e3, 2) if dense_work else None,
                           'frac_of_fp32_mfma_peak': round(value * gflop_img / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 4)
                           if dense_work else None},
            'roofline': roofline, 'cpu_baseline': cpu, 'parity': parity, 'extra': extra,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()          # rank 0 finishes its instrumented pass before anyone tears NCCL down
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
