"""bench.py — fusion-level DynMM hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic batch: forward (dual ResNet-34
NonBottleneck1D encoders, SE fusion, global gate with soft DiffSoftmax gates tau=1, PPM, ESANet
decoder, 4 training outputs) + weighted multi-scale CE + FLOP regulariser + full backward, batch
32/GPU at 480x640 (BASELINE.json configs[2], the configuration `metric` is quoted on).  N>1 = pure
data parallel, weak scaling, gradients all-reduced over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dynmm_amd import dp, ops, synth                        # noqa: E402
from dynmm_amd.nn.net import SkipGateESANet                 # noqa: E402
from dynmm_amd.nn.net_skip import SkipESANet                # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
GFLOP_PER_IMG_FWD_BWD = {'P': 222.98, 'S': 300.8}   # BASELINE.md §2 (conv MACs x2, fwd+bwd)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
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
                    help="train: fwd+bwd soft gates (configs[2]); fwd: eval forward, gate forced on (configs[1])")
    ap.add_argument('--branches', default='all4', choices=['all4', 'uniform', 'all0'],
                    help='--mode fwd only: per-sample gate branch (hard one-hot); uniform = k = n %% 5. '
                         'With K16 compaction depth stage j runs on the samples with k >= j only')
    ap.add_argument('--no-compact', action='store_true', help='--mode fwd: disable K16 compaction (dense reference semantics)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step as one hipGraph.  Default is eager multi-stream launches: the step is '
                         'GPU-bound at batch 32 (eager == graph on one stream) and the 3-stream schedule '
                         '(RGB encoder | depth encoder | weight gradients) overlaps better un-captured')
    ap.add_argument('--no-graph', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--single-stream', action='store_true', help='disable the depth-encoder and wgrad side streams')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=2)
    ap.add_argument('--no-kernel-timing', action='store_true')
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
    g = torch.Generator(device=device).manual_seed(seed)
    rgb = torch.randn(n, 3, h, w, device=device, generator=g)
    depth = torch.randn(n, 1, h, w, device=device, generator=g)
    labels = [torch.randint(0, 41, (n, h // s, w // s), device=device, generator=g, dtype=torch.uint8)
              for s in (1, 8, 16, 32)]
    return rgb, depth, labels


def cpu_baseline(args):
    """The CPU oracle (a port of the reference's PyTorch CPU path, pinned to it by tests/golden) timed on
    this host's cores on a bounded sample of the same workload."""
    from oracle import dynmm_oracle as O
    n = args.cpu_batch
    cfg = O.Config(encoder_block='NonBottleneck1D' if args.config == 'P' else 'BasicBlock', fuse='SE-add')
    m = make_model(args.config, args.height, args.width, 'cpu', args.model)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k]
    rgb, depth, labels = make_batch(n, args.height, args.width, 'cpu', 1234)
    cw = torch.linspace(0.5, 2.0, 40)
    cores = torch.get_num_threads()

    noise = [torch.empty(n, 2).exponential_() for _ in range(4)]

    def one():
        if args.mode == 'fwd':
            with torch.no_grad():
                if args.model == 'skip':
                    O.forward_skip(sd, rgb, depth, cfg, noise, test=True)
                else:
                    O.forward(sd, rgb, depth, cfg, test=True, baseline=True)
            return
        for p in params:
            p.grad = None
        if args.model == 'skip':
            outs, lf = O.forward_skip(sd, rgb, depth, cfg, noise, training=True), 0.0
        else:
            outs, lf = O.forward(sd, rgb, depth, cfg, training=True, temp=1.0)
        losses = O.cross_entropy_2d(outs, labels, cw)
        (sum(losses) + lf).backward()

    one()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    return {'value': round(n / med, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle (PyTorch CPU fp32, {cores} threads), batch {n} of the same {args.height}x{args.width} '
                      f'{args.mode} step, median of 3 after 1 warm-up'}


model_ref = [None]


def kernel_timing(step_fn):
    """One instrumented EAGER step: HIP events around every implicit-GEMM launch on the stream it is
    launched on.  Returns per-kernel-variant totals (launches, ms, algorithmic GFLOP)."""
    # isolated per-kernel durations: the instrumented pass runs on ONE stream (in the timed region the
    # RGB / depth / wgrad streams overlap, which inflates every individual kernel's wall duration)
    saved = (ops.ASYNC_WGRAD, getattr(model_ref[0], 'dual_stream', False))
    ops.ASYNC_WGRAD = False
    model_ref[0].dual_stream = False
    torch.cuda.synchronize()
    ops.PROFILE = []
    step_fn()
    torch.cuda.synchronize()
    rec, ops.PROFILE = ops.PROFILE, None
    ops.ASYNC_WGRAD, model_ref[0].dual_stream = saved
    agg, shapes = {}, {}
    for name, flops, e0, e1, shape in rec:
        ms = e0.elapsed_time(e1)
        for d, key in ((agg, name), (shapes, (name, shape))):
            a = d.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += ms
            a[2] += flops
    if os.environ.get('DYNMM_BENCH_SHAPES'):
        with open(os.environ['DYNMM_BENCH_SHAPES'], 'w') as f:
            f.write('kernel | N,Ci,H,W,Co,KH,KW,SH,SW | launches | total ms | avg us | TFLOP/s\n')
            for (name, shape), (n, ms, fl) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                f.write(f'{name} | {shape} | {n} | {ms:.3f} | {1000 * ms / n:.1f} | {fl / (ms * 1e-3) / 1e12:.1f}\n')
    return agg


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        # nccl == RCCL on ROCm.  DYNMM_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a
        # single-GPU box (ranks then share device 0).
        dist.init_process_group(os.environ.get('DYNMM_DIST_BACKEND', 'nccl'))
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    model = make_model(args.config, args.height, args.width, device, args.model)
    model_ref[0] = model
    dp.broadcast_parameters(model)
    rgb, depth, labels = make_batch(args.batch, args.height, args.width, device, 1234 + rank)
    cw = torch.linspace(0.5, 2.0, 40, device=device)
    train = args.mode == 'train'
    reducer = None
    if train:
        model.train()
        model.hard_gate, model.temp = False, 1.0
        model.dual_stream = not args.single_stream
        ops.ASYNC_WGRAD = not args.single_stream
        reducer = dp.GradBucketReducer(model.parameters(), bucket_mb=32, overlap=False)
        ops.DIRECT_GRAD = True      # kernels write parameter gradients straight into the flat buffer views
    else:
        model.eval()
        model.compact = not args.no_compact
        model.dual_stream = not args.single_stream      # depth-encoder stages on a second HIP stream
        if args.branches == 'all4':
            model.baseline = True                 # configs[1]: static fuse, gate forced on
        else:
            model.ini_stage = True
            model.ini_branches = [(i % 5) if args.branches == 'uniform' else 0 for i in range(args.batch)]

    def fwd_bwd():
        if not train:
            with torch.no_grad():
                return model(rgb, depth, test=True)
        reducer.zero()
        if args.model == 'skip':
            outs, total = model(rgb, depth), 0.0
        else:
            outs, total = model(rgb, depth)
        for o, t in zip(outs, labels):
            total = total + ops.cross_entropy_2d(o, t, cw)
        total.backward()
        ops.join_async()
        return total

    graph = None
    use_graph = args.graph and (train or args.no_compact)   # compaction reads the branch on the host
    # warm-up (also initialises lazily-created buffers before capture)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if use_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fwd_bwd()
        except Exception as e:   # capture is an optimisation of launch overhead, never a different compute path
            print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); using eager launches', file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
        else:
            fwd_bwd()
        if reducer is not None and world > 1:
            reducer.finish()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    ms_per_step = 1000.0 * elapsed / args.steps
    value = args.batch * world * args.steps / elapsed

    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        agg = kernel_timing(fwd_bwd)
        if agg:
            name, (launches, ms, flops) = max(agg.items(), key=lambda kv: kv[1][1])
            achieved = flops / (ms * 1e-3) / 1e12
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_dominant_kernel.json')
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(name, {}).get('hbm_bytes_per_launch')
                except Exception:
                    traffic = None
            roofline = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': FP32_MFMA_PEAK_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), 'traffic': traffic,
                        'kernel': name, 'launches_per_step': launches,
                        'avg_launch_us': round(1000.0 * ms / launches, 2),
                        'algorithmic_gflop_per_launch': round(flops / launches / 1e9, 3),
                        'all_igemm_kernels': {k: {'launches': v[0], 'ms': round(v[1], 3),
                                                  'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2)}
                                              for k, v in sorted(agg.items())}}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        gflop_img = GFLOP_PER_IMG_FWD_BWD[args.config] if train else {'P': 74.67, 'S': 100.62}[args.config]
        line = {
            'metric': 'images/sec fwd+bwd, 480x640 RGB-D, batch 32/GPU' if train else
                      'images/sec fwd-only, 480x640 RGB-D, gate forced on',
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (('configs[2]: fwd+bwd --dynamic --global-gate soft DiffSoftmax gates tau=1, '
                                     'weighted 4-scale CE + FLOP loss' if train else
                                     'configs[1]: fwd-only eval, static fuse (gate forced on)') if args.model == 'gate' else
                                    ('fwd+bwd --dynamic (per-stage Gumbel-softmax gates, block_rule 2222, soft tau=1), '
                                     'weighted 4-scale CE' if train else 'fwd-only eval test=True (hard Gumbel gates; depth stages run on the still-fusing samples only '
                                     'unless --no-compact)')),
                       'net': f'{"SkipGateESANet" if args.model == "gate" else "SkipESANet"} R34-'
                              f'{"NBt1D" if args.config == "P" else "BasicBlock"} SE-add (config {args.config})',
                       'branches': None if train else args.branches, 'compaction': None if train else (not args.no_compact),
                       'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                       'height': args.height, 'width': args.width,
                       'parallelism': f'dp{world}', 'launch': 'hipGraph replay' if graph is not None else 'eager',
                       'streams': 1 if args.single_stream else (3 if train else 2),
                       'model_tflops': round(value * gflop_img / 1e3, 2),
                       'model_frac_of_fp32_mfma_peak': round(value * gflop_img / 1e3 / (FP32_MFMA_PEAK_TFLOPS * world), 4)},
            'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()          # rank 0 finishes its instrumented pass before anyone tears NCCL down
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
