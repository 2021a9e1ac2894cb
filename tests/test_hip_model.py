"""GPU: whole-model parity of the HIP path (SkipGateESANet through the C ABI) against
  (1) the CPU oracle on the same seeded weights/inputs, forward AND backward, and
  (2) the committed golden fixtures produced by the reference itself.
Tolerance: north_star's 1e-3 relative on logits; the fp32-MFMA path is held to 2e-4 here."""
import os

import numpy as np
import pytest
import torch

from dynmm_amd import synth
from tests import helpers as Hh

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-4        # eval-mode logits (fp32 oracle vs fp64 oracle differ by ~4e-5 themselves)
TRAIN_OUT_TOL = 1e-3    # train-mode outputs incl. 3x4 side maps: batch-stat BN amplifies rounding; north_star bar
# Whole-model fp32 gradients: two comparisons.
#  (1) test_model_gradients_vs_fp64_oracle_at_equal_decisions — the parity gate proper: against the fp64 oracle with the HIP pass's
#      ReLU / arg-max decisions imposed at its near-ties, every per-parameter gradient norm to EQUAL_DECISION_NORM_TOL = 0.02, every
EQUAL_DECISION_COS_TOL = 0.999      # per gradient tensor, same construction (measured: see the test's printout)
#      fixture and mode (train_hard included).
#  (2) against the reference's own fp32 gradients in the fixtures (its decisions, its summation order): bars CALIBRATED in
#      tests/golden/grad_noise.npz (tests/golden/make_grad_noise.py: how far the fp32 CPU oracle itself moves between thread
#      counts, oneDNN / native and direct / Winograd-form convolutions — one run of a fixed draw list, no ratchet), x 1.25 with a
#      floor of 0.05.  A fixture whose calibrated bar exceeds GOLDEN_GRAD_CAP = 0.10 (one flipped decision between correct fp32
#      evaluations: P_se train_hard, R50_se) is NOT compared with the fixture's gradients — a bar that wide tests nothing —
#      and must be covered by (1).
GOLDEN_GRAD_CAP = 0.10
NOISE_MARGIN = 1.25
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def grad_noise():
    return np.load(os.path.join(GOLDEN, 'grad_noise.npz'))


def calibrated_tols(cfg, h, w, mode):
    g = grad_noise()
    tags = [str(t) for t in g['small_fixtures']]
    tag = f'{cfg} {h}x{w} {mode}'
    if tag not in tags:
        return None
    nrm, full = g['small'][tags.index(tag)]
    tols = max(NOISE_MARGIN * float(nrm), 0.05), max(NOISE_MARGIN * float(full), 0.05)
    return tols if max(tols) <= GOLDEN_GRAD_CAP else None


def hip_model(cfg_name, h, w, seed=0):
    from dynmm_amd.nn.net import SkipGateESANet
    cfg = Hh.CFGS[cfg_name]
    m = SkipGateESANet(height=h, width=w, encoder_rgb=cfg.encoder, encoder_depth=cfg.encoder,
                       encoder_block=cfg.encoder_block, fuse_depth_in_rgb_encoder=cfg.fuse,
                       nr_decoder_blocks=cfg.nr_decoder_blocks)
    synth.fill_state_dict(m.state_dict(), seed)
    return m.cuda()


def set_mode(m, mode, n):
    m.train(mode.startswith('train'))
    m.baseline = mode == 'eval_baseline'
    m.ini_stage = mode == 'eval_ini'
    m.hard_gate = mode in ('eval_hard', 'train_hard')
    m.temp = 0.5 if mode == 'train_hard' else 1.0


class fixed_randint:
    """ini_stage draws with the CPU RNG (…globalgate.py:269); pin it like the golden generator did."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.real = torch.randint
        torch.randint = lambda *a, **k: Hh.ini_index(self.n)

    def __exit__(self, *a):
        torch.randint = self.real


MODEL_FIXTURES = [('P_se', 96, 128), ('P_add', 96, 128), ('S_se', 96, 128), ('S_add', 96, 128), ('P_se', 160, 192),
                  ('R18_se', 96, 128), ('R50_se', 96, 128)]


@pytest.mark.parametrize('cfg,h,w', MODEL_FIXTURES)
def test_model_matches_reference_goldens(golden_dir, cfg, h, w):
    g = np.load(os.path.join(golden_dir, f'model_{cfg}_{h}x{w}.npz'))
    hh, ww, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, hh, ww, seed=1234, device='cuda')
    for mode in sorted({k.split('/')[0] for k in g.files if '/' in k}):
        m = hip_model(cfg, hh, ww)
        set_mode(m, mode, n)
        if mode.startswith('train'):
            outs, lf = m(rgb, depth)
            loss = Hh.train_loss(outs, lf)
            loss.backward()
            out = outs[0].detach()
            for i, o in enumerate(outs[1:]):
                assert Hh.rel_err(o.detach().cpu(), g[f'{mode}/side{i}']) < TRAIN_OUT_TOL, (mode, 'side', i)
            assert abs(loss.item() - float(g[f'{mode}/loss'])) < 1e-3 * max(1, abs(float(g[f'{mode}/loss'])))
            params = dict(m.named_parameters())
            names = [str(s) for s in g[f'{mode}/grad_names']]
            norms = np.array([0.0 if params[nm].grad is None else params[nm].grad.norm().item() for nm in names])
            ref = g[f'{mode}/grad_norms']
            tols = calibrated_tols(cfg, hh, ww, mode)
            dev = np.abs(norms - ref) / np.maximum(ref, 1e-2 * ref.max())
            if tols is None:
                # decisions differ between correct fp32 evaluations of this fixture: gradient parity is (1) above
                assert (cfg, hh, ww, mode) in EQUAL_DECISION_CASES, (cfg, hh, ww, mode)
                print(f'{cfg} {hh}x{ww} {mode}: gradient-norm deviation from the fixture {dev.max():.4f} (not a bar: see '
                      'test_model_gradients_vs_fp64_oracle_at_equal_decisions)')
            else:
                norm_tol, full_tol = tols
                print(f'{cfg} {hh}x{ww} {mode}: worst gradient-norm deviation {dev.max():.4f} (bar {norm_tol:.4f})')
                bad = dev > norm_tol
                assert not bad.any(), [(names[i], norms[i], ref[i]) for i in np.nonzero(bad)[0][:8]]
            sd = m.state_dict()
            for k in g.files:
                if tols is not None and k.startswith(f'{mode}/grad:') and np.abs(g[k]).max() > 1e-6:   # skip analytically-zero grads
                    assert Hh.rel_err(params[k.split('grad:')[1]].grad.cpu(), g[k]) < tols[1], k
                if k.startswith(f'{mode}/rm:'):
                    assert Hh.rel_err(sd[k.split('rm:')[1] + '.running_mean'].cpu(), g[k]) < 1e-4, k
                if k.startswith(f'{mode}/rv:'):
                    assert Hh.rel_err(sd[k.split('rv:')[1] + '.running_var'].cpu(), g[k]) < 1e-4, k
        else:
            with torch.no_grad(), fixed_randint(n):
                out, weight = m(rgb, depth, test=True, return_weight=True)
            with torch.no_grad(), fixed_randint(n):
                _, lf = m(rgb, depth)
            assert Hh.rel_err(weight.cpu(), g[f'{mode}/weight']) < 1e-4, mode
        assert abs(lf.item() - float(g[f'{mode}/loss_flop'])) < 1e-4, mode
        out = out.cpu()
        tol = TRAIN_OUT_TOL if mode.startswith('train') else LOGIT_TOL
        assert Hh.rel_err(out[:, :, ::stride, ::stride], g[f'{mode}/strided']) < tol, mode
        assert Hh.rel_err(out.sum(dim=(2, 3)), g[f'{mode}/csum']) < 1e-3, mode
        assert Hh.rel_err(out.abs().sum(dim=(2, 3)), g[f'{mode}/cabs']) < 1e-3, mode


class hip_gate_decisions:
    """Run the oracle's DiffSoftmax (…globalgate.py:20-30) with the hard arg-max the HIP pass took wherever the oracle's own two
    largest probabilities lie within `tau` of each other — the same construction as tests/test_hip_blocks.hip_relu_decisions for
    the one other DECISION of the forward pass.  Away from such a near-tie the HIP decision must equal the oracle's (`outside`)."""

    def __init__(self, hip_weight, tau=1e-5):
        self.idx = None if hip_weight is None else hip_weight.argmax(1)
        self.tau = tau
        self.imposed = self.outside = self.calls = 0

    def __enter__(self):
        from oracle import dynmm_oracle as O
        self.O, self.orig = O, O.diff_softmax

        def diff_softmax(logits, tau=1.0, hard=False, dim=-1):
            y_soft = (logits / tau).softmax(dim)
            if not hard or self.idx is None:
                return self.orig(logits, tau, hard, dim)
            self.calls += 1
            top = y_soft.detach().flatten(1).topk(2, dim=1).values
            band = (top[:, 0] - top[:, 1]) <= self.tau
            own = y_soft.detach().flatten(1).argmax(1)
            dis = own != self.idx
            self.imposed += int((dis & band).sum())
            self.outside += int((dis & ~band).sum())
            idx = torch.where(band, self.idx, own).view(-1, *([1] * (logits.dim() - 1)))
            y_hard = torch.zeros_like(logits).scatter_(dim, idx, 1.0)
            return y_hard - y_soft.detach() + y_soft
        O.diff_softmax = diff_softmax
        return self

    def __exit__(self, *a):
        self.O.diff_softmax = self.orig


DECISION_BAND = TRAIN_OUT_TOL     # a pre-activation within the forward bar of zero is a legitimately open decision
EQUAL_DECISION_NORM_TOL = 0.02      # per-parameter gradient norm, HIP fp32 vs the fp64 oracle at equal decisions
EQUAL_DECISION_CASES = [('P_se', 96, 128, 'train_soft'), ('P_se', 96, 128, 'train_hard'), ('P_add', 96, 128, 'train_soft'),
                        ('S_se', 96, 128, 'train_soft'), ('R50_se', 96, 128, 'train_soft'), ('R18_se', 96, 128, 'train_soft'),
                        ('P_se', 160, 192, 'train_soft'), ('S_add', 96, 128, 'train_soft')]


@pytest.mark.parametrize('cfg,h,w,mode', EQUAL_DECISION_CASES)
def test_model_gradients_vs_fp64_oracle_at_equal_decisions(golden_dir, cfg, h, w, mode):
    """Whole-model parameter gradients against the fp64 oracle with the DECISIONS of the HIP pass — every ReLU that conv2d /
    batch_norm_act evaluate, and the hard gate's arg-max — imposed on the oracle where its own pre-activation (its two largest
    gate probabilities) lies within DECISION_BAND of a tie, and REQUIRED to equal the oracle's everywhere else.  A gradient is a function
    of the decisions taken in the forward pass; two correct fp32 evaluations of the same sums take different ones at
    rounding-level pre-activations (one flipped ReLU / arg-max moves a single parameter's gradient norm of
    these small fixtures by up to 0.37), so the fixtures' fp32 gradients — the reference's own decisions — are comparable only
    up to that, while at EQUAL decisions the HIP gradients are held to 0.02 on every per-parameter norm (measured: <= 0.006), `train_hard` included.
    Same inputs, weights, modes and loss as the reference fixtures (tests/golden/make_goldens.py)."""
    from oracle import dynmm_oracle as O
    from dynmm_amd import ops
    from tests.test_hip_blocks import hip_relu_decisions
    g = np.load(os.path.join(golden_dir, f'model_{cfg}_{h}x{w}.npz'))
    hh, ww, n, _ = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, hh, ww, seed=1234)
    m = hip_model(cfg, hh, ww)
    set_mode(m, mode, n)
    m.dual_stream = False                    # the oracle evaluates the RGB stage before the depth stage: same ReLU order
    m.start_weight()
    ops.ACT_TRACE = []
    try:
        outs, lf = m(rgb.cuda(), depth.cuda())
    finally:
        trace, ops.ACT_TRACE = ops.ACT_TRACE, None
    hip_weight = m.weight_list.clone()
    m.save_weight_info = False
    Hh.train_loss(outs, lf).backward()
    torch.cuda.synchronize()

    sd = Hh.filled_state_dict(Hh.CFGS[cfg], seed=0)
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    kw = dict(Hh.MODE_KW[mode])
    # band: the forward's own tolerance — pre-activations of the deep layers differ from fp64 by up to ~1e-4 of the tensor's
    # maximum after ~100 layers with batch-statistic BatchNorms (TRAIN_OUT_TOL is 1e-3), against 1e-5 for a single block
    with hip_relu_decisions(trace, tau=DECISION_BAND) as census, \
            hip_gate_decisions(hip_weight if kw.get('hard_gate') else None, tau=DECISION_BAND) as gate:
        outs64, lf64 = O.forward(sd, rgb.double(), depth.double(), Hh.CFGS[cfg], **kw)
    assert census['outside_band'] == 0, f'{census["outside_band"]} ReLU decisions differ from the fp64 oracle away from zero'
    assert not any(census['queues'].values()), 'traced HIP ReLU outputs the oracle never matched'
    assert gate.outside == 0 and (gate.calls == 1) == bool(kw.get('hard_gate')), (gate.outside, gate.calls)
    tot = 3.0 * lf64
    for i, o in enumerate(outs64):
        tot = tot + (o * Hh.grad_probe(tuple(o.shape), f's{i}').double()).mean()
    tot.backward()
    for a, b in zip(outs, outs64):
        assert Hh.rel_err(a.detach().cpu(), b.detach()) < TRAIN_OUT_TOL
    hp = dict(m.named_parameters())
    names = [k for k in params if params[k].grad is not None]
    ref = np.array([params[k].grad.norm().item() for k in names])
    got = np.array([0.0 if hp[k].grad is None else hp[k].grad.double().norm().item() for k in names])
    dev = np.abs(got - ref) / np.maximum(ref, 1e-2 * ref.max())
    worst = int(dev.argmax())
    print(f'{cfg} {hh}x{ww} {mode}: {census["imposed"]} ReLU + {gate.imposed} gate decisions imposed of {census["sites"]} traced '
          f'activations; worst gradient-norm deviation {dev.max():.4f} ({names[worst]})')
    assert dev.max() < EQUAL_DECISION_NORM_TOL, [(names[i], got[i], ref[i]) for i in np.argsort(-dev)[:8]]
    # ... and on every gradient TENSOR's direction (a sign / permutation bug confined to one small tensor leaves its
    # norm alone): cosine with the fp64 gradient, for every parameter whose gradient is not rounding noise
    cos = {}
    for k, r in zip(names, ref):
        if r < 1e-3 * ref.max() or hp[k].grad is None:
            continue
        a_, b_ = hp[k].grad.detach().double().cpu().flatten(), params[k].grad.flatten()
        cos[k] = float(torch.dot(a_, b_) / (a_.norm() * b_.norm()).clamp_min(1e-300))
    wk = min(cos, key=cos.get)
    print(f'   lowest gradient-tensor cosine {cos[wk]:.6f} ({wk}) over {len(cos)} tensors')
    assert cos[wk] > EQUAL_DECISION_COS_TOL, sorted(cos.items(), key=lambda kv: kv[1])[:8]


def _oracle_train_step(cfg, rgb, depth, dtype, seed, temp):
    from oracle import dynmm_oracle as O
    sd = Hh.filled_state_dict(Hh.CFGS[cfg], seed=seed)
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    outs, lf = O.forward(sd, rgb.to(dtype), depth.to(dtype), Hh.CFGS[cfg], training=True, temp=temp)
    tot = 3.0 * lf
    for i, o in enumerate(outs):
        tot = tot + (o * Hh.grad_probe(tuple(o.shape), f's{i}').to(dtype)).mean()
    tot.backward()
    return outs, lf, {k: p.grad for k, p in params.items()}, sd


def _rl2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize('cfg,dual', [('P_se', False), ('S_add', False), ('P_se', True)])
def test_model_vs_oracle_fwd_bwd_full_tensors(cfg, dual):
    """Every output and EVERY parameter-gradient tensor of one train step, HIP vs oracle.

    Train-mode gradients of this ~100-layer net are ill-conditioned in fp32 (stacked BatchNorm
    projections cancel most of each upstream gradient): the CPU oracle in fp32 differs from the same
    oracle in fp64 by ~1e-2 (median over tensors) and up to ~1e-1 on single tensors [measured in the
    build container], while the logits agree to ~4e-5.  So the bar is stated against the fp64 truth:
    the HIP fp32 path must be as close to fp64 as the reference-equivalent fp32 CPU path is (x3),
    tensor by tensor and in aggregate.  Tight per-block gradient parity (5e-4) is in
    tests/test_hip_blocks.py, where the conditioning is benign."""
    h, w, n = 96, 128, 3
    rgb, depth = synth.synth_inputs(n, h, w, seed=99)
    outs32, lf32, g32, _ = _oracle_train_step(cfg, rgb, depth, torch.float32, 5, 0.7)
    outs64, lf64, g64, sd64 = _oracle_train_step(cfg, rgb, depth, torch.float64, 5, 0.7)

    m = hip_model(cfg, h, w, seed=5)
    m.train()
    m.temp = 0.7
    m.dual_stream = dual          # depth encoder on a second HIP stream (fwd and, via autograd, bwd)
    outs, lf = m(rgb.cuda(), depth.cuda())
    Hh.train_loss(outs, lf).backward()
    torch.cuda.synchronize()
    for a, b32, b64 in zip(outs, outs32, outs64):
        e_ref = Hh.rel_err(b32.detach(), b64.detach())
        assert Hh.rel_err(a.detach().cpu(), b64.detach()) < max(3 * e_ref, 1e-5) and \
            Hh.rel_err(a.detach().cpu(), b32.detach()) < LOGIT_TOL
    assert abs(lf.item() - lf64.item()) < 1e-5
    gmax = max(v.abs().max().item() for v in g64.values())
    e_hip, e_ref, names = [], [], []
    for name, p in m.named_parameters():
        if g64[name].abs().max().item() < 1e-5 * gmax:
            continue        # analytically-zero gradients (conv bias in front of a train-mode BN)
        names.append(name)
        e_hip.append(_rl2(p.grad.cpu(), g64[name]))
        e_ref.append(_rl2(g32[name], g64[name]))
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    assert np.median(e_hip) <= 3 * np.median(e_ref) + 1e-5, (np.median(e_hip), np.median(e_ref))
    assert e_hip.max() <= 3 * e_ref.max() + 1e-3, (names[int(e_hip.argmax())], e_hip.max(), e_ref.max())
    cat = lambda d, src: torch.cat([src(nm).double().flatten() for nm in names])   # noqa: E731
    params = dict(m.named_parameters())
    all_hip, all_32, all_64 = cat(0, lambda nm: params[nm].grad.cpu()), cat(0, lambda nm: g32[nm]), cat(0, lambda nm: g64[nm])
    assert _rl2(all_hip, all_64) <= 3 * _rl2(all_32, all_64) + 1e-5
    new_sd = m.state_dict()
    for k, v in sd64.items():
        if 'running_' in k:
            assert Hh.rel_err(new_sd[k].cpu(), v.detach()) < 1e-4, k


def test_nyu8_baseline_config0(golden_dir):
    """BASELINE.json configs[0] at full 480x640: strided logits, argmax histogram, CM and mIoU vs the
    reference's own outputs on the 8 synthetic NYUv2-like pairs."""
    from oracle import dynmm_oracle as O
    g = np.load(os.path.join(golden_dir, 'nyu8_P_se.npz'))
    m = hip_model('P_se', 480, 640)
    m.eval()
    m.baseline = True
    rgb, depth = synth.synth_inputs(8, 480, 640, seed=77, nyu_like=True, device='cuda')
    label = synth.synth_labels(8, 480, 640, seed=78)
    with torch.no_grad():
        out = m(rgb, depth, test=True).cpu()
    assert Hh.rel_err(out[:, :, ::32, ::32], g['strided']) < LOGIT_TOL
    assert Hh.rel_err(out.sum(dim=(2, 3)), g['csum']) < 1e-3
    hist = torch.stack([torch.bincount(out[i].argmax(0).flatten(), minlength=40) for i in range(8)])
    assert (hist.numpy() - g['hist']).__abs__().sum() <= 64          # fp32-rounding argmax flips only
    lab, prd = O.eval_postprocess(out, label)
    cm = O.confusion_matrix(lab, prd, 40)
    assert np.abs(cm.numpy() - g['cm']).sum() <= 64
    _, miou = O.iou_from_cm(cm)
    assert abs(miou.item() - float(g['miou'])) < 1e-4               # north_star: mIoU within 1e-3 relative


def test_full_size_properties():
    """At BASELINE's full size (480x640) the oracle is too slow to run per test; check
    size-independent properties instead: batch-composition invariance in eval mode, linearity of the
    gate blend (baseline one-hot == ESANet static fusion), determinism."""
    m = hip_model('P_se', 480, 640)
    m.eval()
    rgb, depth = synth.synth_inputs(4, 480, 640, seed=3, device='cuda')
    with torch.no_grad():
        m.hard_gate = True
        full, wfull = m(rgb, depth, test=True, return_weight=True)
        again = m(rgb, depth, test=True)
        halves = torch.cat([m(rgb[:2], depth[:2], test=True), m(rgb[2:], depth[2:], test=True)])
    assert torch.equal(full, again)                                   # deterministic
    assert Hh.rel_err(halves.cpu(), full.cpu()) < 1e-5                # samples independent in eval
    assert torch.all((wfull.sum(1) - 1).abs() < 1e-6)
    assert full.shape == (4, 40, 480, 640) and torch.isfinite(full).all()


@pytest.mark.parametrize('cfg', ['P_se', 'S_add'])
def test_hard_gate_compaction_is_exact(cfg):
    """K16: per-sample branch compaction in inference (depth stage j runs only on samples with branch >= j)
    must reproduce the dense reference semantics; checked against the dense HIP path and the oracle."""
    from oracle import dynmm_oracle as O
    h, w, n = 96, 128, 6
    branches = [1, 4, 0, 3, 2, 4]
    rgb, depth = synth.synth_inputs(n, h, w, seed=21)
    m = hip_model(cfg, h, w, seed=2)
    m.eval()
    m.ini_stage = True
    real = torch.randint
    torch.randint = lambda *a, **k: torch.tensor(branches)
    try:
        with torch.no_grad():
            m.compact = True
            out_c, w_c = m(rgb.cuda(), depth.cuda(), test=True, return_weight=True)
            assert m.last_stage_batch == [5, 4, 3, 2]         # samples still needing depth at stages 1..4
            m.compact = False
            out_d, w_d = m(rgb.cuda(), depth.cuda(), test=True, return_weight=True)
            assert m.last_stage_batch is None
    finally:
        torch.randint = real
    assert torch.equal(w_c, w_d)
    assert Hh.rel_err(out_c.cpu(), out_d.cpu()) < 1e-5
    sd = Hh.filled_state_dict(Hh.CFGS[cfg], seed=2)
    iw = torch.zeros(n, 5)
    iw[range(n), branches] = 1
    with torch.no_grad():
        ref = O.forward(sd, rgb, depth, Hh.CFGS[cfg], test=True, ini_stage=True, ini_weight=iw)
    assert Hh.rel_err(out_c.cpu(), ref) < LOGIT_TOL
    # all-skip and all-fuse extremes
    for br, expect in (([0] * n, [0, 0, 0, 0]), ([4] * n, [n] * 4)):
        torch.randint = lambda *a, **k: torch.tensor(br)
        try:
            with torch.no_grad():
                m.compact = True
                oc = m(rgb.cuda(), depth.cuda(), test=True)
                assert m.last_stage_batch == expect
                m.compact = False
                od = m(rgb.cuda(), depth.cuda(), test=True)
        finally:
            torch.randint = real
        assert Hh.rel_err(oc.cpu(), od.cpu()) < 1e-5


def test_edge_shapes_and_errors():
    """Edge cases around the boundary: batch 1 inference, the smallest legal resolution, batch 1 in
    training mode (BatchNorm over one value -> the same ValueError PyTorch raises), inputs whose size is
    not a multiple of 32 (decoder skip shapes cannot match), mismatched rgb/depth."""
    from dynmm_amd.lib import DynmmHipError
    m = hip_model('P_se', 96, 128)
    m.eval()
    with torch.no_grad():
        rgb, depth = synth.synth_inputs(1, 96, 128, seed=1, device='cuda')
        out = m(rgb, depth, test=True)
        assert out.shape == (1, 40, 96, 128) and torch.isfinite(out).all()
        rgb2, depth2 = synth.synth_inputs(2, 96, 128, seed=1, device='cuda')
        out2 = m(torch.cat([rgb, rgb2[1:]]), torch.cat([depth, depth2[1:]]), test=True)
        assert Hh.rel_err(out2[:1].cpu(), out.cpu()) < 1e-5          # batch-composition invariance
        with pytest.raises(DynmmHipError):
            r, d = synth.synth_inputs(1, 100, 128, seed=1, device='cuda')     # 100 is not a multiple of 32
            m(r, d, test=True)
        with pytest.raises(DynmmHipError):
            m(rgb, depth2, test=True)                                          # batch mismatch rgb vs depth
    m.train()
    with pytest.raises(ValueError):
        m(rgb, depth)                                                          # PPM 1x1 branch: one value per channel


@pytest.mark.parametrize('dual', [False, True])
def test_all_parameter_gradients_are_run_to_run_reproducible(dual):
    """No fp32 atomics anywhere on the gradient path (conv slabs, SE / gate MLPs, depthwise upsample, bias sums are all reduced
    in a fixed order): two runs of the same train step give bit-identical gradients for EVERY parameter, single-stream and on
    the 3-stream schedule alike.  The one order-dependent accumulation is in FLOAT64: the per-channel BatchNorm sums that the
    convolution epilogues add with fp64 atomics (csrc/conv_wino.hip STATS / BNRED; bn_stats / bn_bwd_reduce finish their sums the
    same way).  Their order noise is ~1e-16 relative and is rounded away when mean / invstd / dgamma / dbeta are formed in fp32
    (a change needs the fp64 value within 1e-16 of an fp32 rounding boundary: ~1e-9 per value) — bit-identical here, and at the
    size where tiles number in the thousands in test_full_size_gradients_are_run_to_run_reproducible below."""
    from dynmm_amd import engine
    h, w, n = 96, 128, 4
    rgb, depth = synth.synth_inputs(n, h, w, seed=5, device='cuda')
    grads = []
    for _ in range(3):
        m = hip_model('P_se', h, w, seed=1)
        m.train()
        m.temp, m.dual_stream = 0.7, dual
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        with engine.direct_gradients(dual):
            outs, lf = m(rgb, depth)
            Hh.train_loss(outs, lf).backward()
            from dynmm_amd import ops
            ops.join_async()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
    for other in grads[1:]:
        diff = [k for k in grads[0] if not torch.equal(grads[0][k], other[k])]
        assert not diff, diff[:8]


def test_full_size_gradients_are_run_to_run_reproducible():
    """BASELINE configs[2]'s own size — batch 32, 480x640, the 3-stream schedule, BatchNorm statistics and backward reductions
    from the convolution epilogues (4800 pixel tiles per C = 64 launch adding fp64 atomics into 1-8 slabs): two runs of the same
    train step from the same state.  fp64-atomic order is the ONE tolerated source of run-to-run difference on this path (see
    above); it is bounded here: every parameter gradient within 1e-6 in relative L2 of the first run's — in practice
    bit-identical, the count of tensors that are not is printed."""
    from dynmm_amd import engine, ops
    h, w, n = 480, 640, 32
    rgb, depth = synth.synth_inputs(n, h, w, seed=5, device='cuda')
    m = hip_model('P_se', h, w, seed=1)
    m.train()
    m.temp, m.dual_stream = 0.7, True
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    grads = []
    for _ in range(2):
        for p in m.parameters():
            p.grad.zero_()
        with engine.direct_gradients(True):
            outs, lf = m(rgb, depth)
            Hh.train_loss(outs, lf).backward()
            ops.join_async()
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters()})
        del outs, lf
    diff = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    worst = max((_rl2(grads[1][k], grads[0][k]) for k in diff), default=0.0)
    print(f'batch 32, 480x640: {len(diff)} of {len(grads[0])} gradient tensors not bit-identical between two runs (worst relative L2 {worst:.1e})')
    assert worst < 1e-6, (len(diff), worst, diff[:8])


# ---------------------------------------------------------------------------------------------------
# train-mode parity with the real loss: reference-generated N=8 fixture, and the benchmark resolution
# ---------------------------------------------------------------------------------------------------
def _hip_train_step(cfg, h, w, rgb, depth, labels, cw, ratio, seed=0, temp=1.0):
    """One TrainStep body (forward, weighted 4-scale CE, total-loss rule, backward) on the HIP path; returns
    (model, outs-free dict of losses, {name: grad})."""
    from dynmm_amd import engine
    m = hip_model(cfg, h, w, seed=seed)
    m.train()
    m.temp, m.hard_gate = temp, False
    step = engine.TrainStep(m, cw, lr=0.0, loss_ratio=ratio, flop_budget=0.0)
    captured = {}
    real_forward = m.forward

    def spy(*a, **k):
        res = real_forward(*a, **k)
        captured['outs'] = [o.detach() for o in res[0]]
        return res
    m.forward = spy
    step._body(rgb, depth, [t.to(torch.uint8) for t in labels])
    torch.cuda.synchronize()
    m.forward = real_forward
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    return m, step.last, captured['outs'], grads


def _sample(t, limit=128):
    f = t.detach().reshape(-1)
    return f[::max(1, -(-f.numel() // limit))]


def test_train_step_matches_reference_n8_fixture(golden_dir):
    """Reference-generated (tests/golden/make_goldens.py::train_n8_fixture): config P, 160x192, N = 8, soft
    gates, the reference's weighted 4-scale CE + 0.5 * flop loss, run by the reference in fp32 AND fp64.
    Outputs / losses / running statistics are held to the reference's fp32 values; every parameter gradient
    (128-element samples) is held to the fp64 run within max(3 x the reference's own fp32 error, 1e-3), the
    full sampled gradient vector additionally by cosine and against the fp32 reference."""
    g = np.load(os.path.join(golden_dir, 'train_n8_P_se_160x192.npz'))
    h, w, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s, device='cuda') for s in (1, 8, 16, 32)]
    m, last, outs, grads = _hip_train_step('P_se', h, w, rgb, depth, labels, g['cw'], float(g['ratio']))
    assert np.allclose(last['losses'].cpu().numpy(), g['f32/losses'], rtol=2e-5), (last['losses'], g['f32/losses'])
    assert abs(last['loss_flop'].item() - float(g['f32/loss_flop'])) < 1e-5
    assert abs(last['total'].item() - float(g['f32/total'])) < 2e-5 * float(g['f32/total'])
    out = outs[0].cpu()
    assert Hh.rel_err(out[:, :, ::stride, ::stride], g['out/strided']) < TRAIN_OUT_TOL
    assert Hh.rel_err(out.sum(dim=(2, 3)), g['out/csum']) < 1e-3
    for i, st in enumerate((4, 2, 1)):
        assert Hh.rel_err(outs[1 + i].cpu()[:, :, ::st, ::st], g[f'out/side{i}']) < TRAIN_OUT_TOL, i
    sd = m.state_dict()
    for k in g.files:
        if k.startswith('f32/rm:'):
            assert Hh.rel_err(sd[k.split('rm:')[1] + '.running_mean'].cpu(), g[k]) < 1e-4, k
        if k.startswith('f32/rv:'):
            assert Hh.rel_err(sd[k.split('rv:')[1] + '.running_var'].cpu(), g[k]) < 1e-4, k
    names = [str(s) for s in g['grad_names']]
    gmax = max(np.abs(g['f64/g:' + nm]).max() for nm in names)
    e_hip, e_ref, e_h32, used, cat = [], [], [], [], {'hip': [], 'f32': [], 'f64': []}
    for nm in names:
        g64 = torch.from_numpy(g['f64/g:' + nm]).double()
        if g64.abs().max().item() < 1e-5 * gmax:
            continue                    # analytically-zero gradients (conv bias in front of a train-mode BN)
        g32 = torch.from_numpy(g['f32/g:' + nm]).double()
        gh = _sample(grads[nm]).cpu().double()
        used.append(nm)
        e_hip.append(_rl2(gh, g64))
        e_ref.append(_rl2(g32, g64))
        e_h32.append(_rl2(gh, g32))
        for key, v in (('hip', gh), ('f32', g32), ('f64', g64)):
            cat[key].append(v)
    e_hip, e_ref, e_h32 = np.array(e_hip), np.array(e_ref), np.array(e_h32)
    # e_hip and e_ref are two draws of the same fp32 noise (ReLU / max-pool decisions that flip between fp32 and
    # fp64 forward passes dominate it): compare the DISTRIBUTIONS tightly and each tensor against a bar that allows
    # for the draw-to-draw spread (a wrong kernel shows up as O(1) on its tensor, two orders above these bars)
    print(f'per-tensor grad err vs fp64: hip median {np.median(e_hip):.2e} p95 {np.percentile(e_hip, 95):.2e} max '
          f'{e_hip.max():.2e} | reference fp32 median {np.median(e_ref):.2e} p95 {np.percentile(e_ref, 95):.2e} max '
          f'{e_ref.max():.2e} | worst ratio {np.max(e_hip / np.maximum(e_ref, 1e-4)):.2f}')
    bad = e_hip > np.maximum(8 * e_ref, np.maximum(3 * np.median(e_ref), 1e-3))
    assert not bad.any(), [(used[i], e_hip[i], e_ref[i]) for i in np.nonzero(bad)[0][:8]]
    # The distribution is that of a chaotic process (which ReLU / max-pool decisions flip between an fp32 and an fp64
    # forward), not a precision figure: swapping the stem / gate convolutions between two kernels that are each at
    # least as accurate against fp64 as the CPU's own fp32 conv (scratch/stem_acc.py: 2.2e-7 / 5.1e-7 rms) moved the
    # median over 2.27e-2 .. 2.80e-2, the p95 over 2.66e-2 .. 3.48e-2 and the max over 0.92e-1 .. 1.58e-1 (fp32 oracle:
    # 1.96e-2 / 2.22e-2 / 1.02e-1).  The bars sit at 2x / 2x / 3x of the oracle's own fp32 error.
    # Round 3: the noise floor is CALIBRATED — the fp32 CPU oracle under 6 summation orders against the same fp64 truth at
    # this very point (grad_noise.npz 'n8': median 1.20e-2..1.58e-2, p95 1.61e-2..2.08e-2, max 4.0e-2..6.3e-2) — and HIP must sit inside the observed range x 1.25
    # (round 2: 2x / 2x / 3x of ONE draw).
    bars = NOISE_MARGIN * grad_noise()['n8'].max(axis=0)
    print(f'calibrated bars (median / p95 / max / cosine deficit): {bars[0]:.3e} {bars[1]:.3e} {bars[2]:.3e} {NOISE_MARGIN * bars[3]:.3e}')
    assert np.median(e_hip) <= bars[0], (np.median(e_hip), bars[0])
    assert np.percentile(e_hip, 95) <= bars[1], (np.percentile(e_hip, 95), bars[1])
    assert e_hip.max() <= bars[2], (e_hip.max(), bars[2])
    A, B32, B64 = (torch.cat(cat[k]) for k in ('hip', 'f32', 'f64'))
    cos64 = torch.nn.functional.cosine_similarity(A, B64, dim=0).item()
    cos_ref = torch.nn.functional.cosine_similarity(B32, B64, dim=0).item()
    print(f'n8 fixture: grad err vs fp64 median {np.median(e_hip):.2e} (reference fp32: {np.median(e_ref):.2e}), '
          f'max {e_hip.max():.2e} ({e_ref.max():.2e}); vs fp32 reference median {np.median(e_h32):.2e}; '
          f'cosine {cos64:.6f} (reference fp32: {cos_ref:.6f})')
    assert 1 - cos64 <= NOISE_MARGIN * bars[3], (cos64, bars[3])          # (quadratic in the error: margin^2 on the deficit)
    # against the reference's fp32 gradients directly: both sides carry ~1e-2 of fp32 conditioning noise
    assert np.median(e_h32) < 2.5e-2 and e_h32.max() < 8e-2, (np.median(e_h32), e_h32.max())


def test_train_step_parity_at_benchmark_resolution():
    """BASELINE configs[2] at its own resolution: 480x640, config P, soft gates tau = 1, weighted 4-scale CE +
    flop loss — batch 2 (the oracle's fp64 step takes ~20 s on the GPU box's host).  HIP TrainStep body vs the
    CPU oracle in fp32 and fp64: outputs, the five loss terms, running statistics, and EVERY parameter
    gradient (per-tensor relative L2 against fp64 within max(3 x the fp32 oracle's own error, 1e-3); full
    gradient cosine)."""
    from oracle import dynmm_oracle as O
    h, w, n, ratio = 480, 640, 2, 0.5
    cw = np.linspace(0.5, 2.0, 40).astype(np.float32)
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s) for s in (1, 8, 16, 32)]
    ref = {}
    for tag, dt in (('f32', torch.float32), ('f64', torch.float64)):
        sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
        sd = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
        params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
        outs, lf = O.forward(sd, rgb.to(dt), depth.to(dt), Hh.CFGS['P_se'], training=True, temp=1.0)
        losses = O.cross_entropy_2d(outs, labels, torch.from_numpy(cw).to(dt))
        total = sum(losses) + ratio * torch.clamp(lf, min=0.0)
        total.backward()
        ref[tag] = dict(outs=[o.detach() for o in outs], losses=torch.stack([l.detach() for l in losses]),
                        lf=lf.detach(), total=total.detach(), grads={k: p.grad for k, p in params.items()}, sd=sd)
    m, last, outs, grads = _hip_train_step('P_se', h, w, rgb.cuda(), depth.cuda(), [l.cuda() for l in labels], cw, ratio)
    r32, r64 = ref['f32'], ref['f64']
    for a, b32, b64 in zip(outs, r32['outs'], r64['outs']):
        assert Hh.rel_err(a.cpu(), b32) < TRAIN_OUT_TOL and Hh.rel_err(a.cpu(), b64) < TRAIN_OUT_TOL
    l_hip = last['losses'].cpu().double()
    assert torch.allclose(l_hip, r64['losses'], rtol=1e-5), (l_hip, r64['losses'])
    assert abs(last['loss_flop'].item() - r64['lf'].item()) < 1e-5 * max(1.0, abs(r64['lf'].item()))
    assert abs(last['total'].item() - r64['total'].item()) < 1e-5 * r64['total'].item()
    new_sd = m.state_dict()
    for k, v in r32['sd'].items():
        if 'running_' in k:
            assert Hh.rel_err(new_sd[k].cpu(), v.detach()) < 1e-4, k
    gmax = max(v.abs().max().item() for v in r64['grads'].values())
    names, e_hip, e_ref = [], [], []
    for name in grads:
        if r64['grads'][name].abs().max().item() < 1e-5 * gmax:
            continue
        names.append(name)
        e_hip.append(_rl2(grads[name].cpu(), r64['grads'][name]))
        e_ref.append(_rl2(r32['grads'][name], r64['grads'][name]))
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    print(f'per-tensor grad err vs fp64: hip median {np.median(e_hip):.2e} p95 {np.percentile(e_hip, 95):.2e} max '
          f'{e_hip.max():.2e} | fp32 oracle median {np.median(e_ref):.2e} p95 {np.percentile(e_ref, 95):.2e} max '
          f'{e_ref.max():.2e} | worst ratio {np.max(e_hip / np.maximum(e_ref, 1e-4)):.2f}')
    bad = e_hip > np.maximum(8 * e_ref, np.maximum(3 * np.median(e_ref), 1e-3))
    assert not bad.any(), [(names[i], e_hip[i], e_ref[i]) for i in np.nonzero(bad)[0][:8]]
    # The distribution is that of a chaotic process (which ReLU / max-pool decisions flip between an fp32 and an fp64
    # forward), not a precision figure: swapping the stem / gate convolutions between two kernels that are each at
    # least as accurate against fp64 as the CPU's own fp32 conv (scratch/stem_acc.py: 2.2e-7 / 5.1e-7 rms) moved the
    # median over 2.27e-2 .. 2.80e-2, the p95 over 2.66e-2 .. 3.48e-2 and the max over 0.92e-1 .. 1.58e-1 (fp32 oracle:
    # 1.96e-2 / 2.22e-2 / 1.02e-1).  The bars sit at 2x / 2x / 3x of the oracle's own fp32 error.
    # Round 3: the noise floor is CALIBRATED — the fp32 CPU oracle under 6 summation orders against the same fp64 truth at
    # this very point (grad_noise.npz 'b2_480x640': median 1.89e-2..2.71e-2, p95 2.23e-2..3.34e-2, max 6.8e-2..1.9e-1) — and HIP must sit inside the observed range x 1.25
    # (round 2: 2x / 2x / 3x of ONE draw).
    bars = NOISE_MARGIN * grad_noise()['b2_480x640'].max(axis=0)
    print(f'calibrated bars (median / p95 / max / cosine deficit): {bars[0]:.3e} {bars[1]:.3e} {bars[2]:.3e} {NOISE_MARGIN * bars[3]:.3e}')
    assert np.median(e_hip) <= bars[0], (np.median(e_hip), bars[0])
    assert np.percentile(e_hip, 95) <= bars[1], (np.percentile(e_hip, 95), bars[1])
    assert e_hip.max() <= bars[2], (e_hip.max(), bars[2])
    flat = lambda src: torch.cat([src(nm).double().flatten() for nm in names])   # noqa: E731
    A, B32, B64 = flat(lambda nm: grads[nm].cpu()), flat(lambda nm: r32['grads'][nm]), flat(lambda nm: r64['grads'][nm])
    cos64 = torch.nn.functional.cosine_similarity(A, B64, dim=0).item()
    cos_ref = torch.nn.functional.cosine_similarity(B32, B64, dim=0).item()
    print(f'480x640 batch 2: logits rel err {Hh.rel_err(outs[0].cpu(), r64["outs"][0]):.2e}; grad err vs fp64 median '
          f'{np.median(e_hip):.2e} (fp32 oracle {np.median(e_ref):.2e}), max {e_hip.max():.2e} ({e_ref.max():.2e}); '
          f'cosine {cos64:.6f} (fp32 oracle {cos_ref:.6f}); |g| ratio {(A.norm() / B64.norm()).item():.5f}')
    assert 1 - cos64 <= NOISE_MARGIN * bars[3], (cos64, bars[3])          # (quadratic in the error: margin^2 on the deficit)
    assert _rl2(A, B64) <= 3 * _rl2(B32, B64) + 1e-5


def test_compacted_training_backward_matches_dense():
    """K16 in training (`compact_train`, BASELINE configs[3]): with BatchNorm in eval mode (running statistics — the only
    per-batch coupling of the depth path) the compacted forward AND backward must equal the dense ones up to fp32
    rounding for every non-gate parameter: sorted-prefix stages, pass-through of the skipped samples, permutation and
    un-permutation are all differentiable plumbing.  (Gate parameters differ by construction: a stage's straight-through
    term comes only from the samples that run it — the documented approximation.)"""
    h, w, n = 96, 128, 7
    branches = [2, 4, 0, 1, 4, 3, 0]
    rgb, depth = synth.synth_inputs(n, h, w, seed=11, device='cuda')
    res = {}
    for compact in (False, True):
        m = hip_model('P_se', h, w, seed=4)
        m.eval()                                   # BN: running statistics; gradients are still recorded below
        m.hard_gate, m.temp = True, 0.7
        m.branch_override = branches
        m.compact_train = compact
        out, lf = m(rgb, depth)
        assert (m.last_stage_batch == [5, 4, 3, 2]) if compact else (m.last_stage_batch is None)
        (out * Hh.grad_probe(tuple(out.shape), 'c').cuda()).mean().backward()
        torch.cuda.synchronize()
        res[compact] = (out.detach(), lf.detach(), {k: p.grad.detach().clone() for k, p in m.named_parameters()
                                                    if p.grad is not None})
    (od, ld, gd), (oc, lc, gc) = res[False], res[True]
    assert Hh.rel_err(oc.cpu(), od.cpu()) < 1e-5 and abs(lc.item() - ld.item()) < 1e-6
    gmax = max(v.abs().max().item() for v in gd.values())
    # the gate's input is the pooled stem output, so the stems / stem fusion sit UPSTREAM of the approximated gate
    # gradient; every other parameter (encoder stages, fusion, skips, context module, decoder) must agree
    upstream = ('gate', 'encoder_rgb.conv1', 'encoder_rgb.bn1', 'encoder_depth.conv1', 'encoder_depth.bn1', 'se_layer0')
    checked = 0
    for k, g in gd.items():
        if any(u in k for u in upstream) or g.abs().max().item() < 1e-6 * gmax:
            continue
        assert _rl2(gc[k].cpu(), g.cpu()) < 2e-4, k
        checked += 1
    assert checked > 400
    assert any('gate' in k and v.abs().max() > 0 for k, v in gc.items())      # the gate still trains


def test_benchmark_config_train_step_invariants():
    """BASELINE configs[2] at its FULL size (batch 32, 480x640, config P, soft gates, weighted 4-scale CE + FLOP loss),
    where the oracle is too slow: size-independent properties of the whole step body instead —
      * the weighted CE normalises by the weight mass, so doubling every class weight (an exact power-of-two scaling)
        must leave all losses and every parameter gradient unchanged (exercises the fused up-sampling + CE tail, the
        loss head's per-scale seeds and everything downstream);
      * BatchNorm statistics, the FLOP loss and the mean-reduced losses are symmetric in the batch, so permuting the
        samples (inputs and labels alike) changes only summation orders: same losses to 1e-5, same gradient direction
        (the gradient itself carries the chaotic ReLU-flip noise of DESIGN.md section 1, hence a cosine bar);
      * finite everywhere, every trainable parameter receives a gradient."""
    from dynmm_amd import engine
    h, w, n = 480, 640, 32
    rgb, depth = synth.synth_inputs(n, h, w, seed=11, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=40 + s, device='cuda').to(torch.uint8) for s in (1, 8, 16, 32)]
    cw = np.linspace(0.5, 2.0, 40).astype(np.float32)

    def run(cwv, perm=None):
        m = hip_model('P_se', h, w, seed=2)
        m.train()
        m.temp, m.hard_gate = 1.0, False
        step = engine.TrainStep(m, cwv, lr=0.0, loss_ratio=0.5, flop_budget=0.0)
        r, d, ls = (rgb, depth, labels) if perm is None else (rgb[perm], depth[perm], [t[perm] for t in labels])
        step._body(r.contiguous(), d.contiguous(), [t.contiguous() for t in ls])
        torch.cuda.synchronize()
        grads = torch.cat([p.grad.detach().flatten() for p in m.parameters()]).double()
        touched = len(step._touched)
        out = (step.last['losses'].double().cpu(), step.last['total'].double().cpu(), grads, touched,
               sum(1 for _ in m.parameters()))
        del step, m
        torch.cuda.empty_cache()
        return out
    base = run(cw)
    assert torch.isfinite(base[0]).all() and torch.isfinite(base[2]).all()
    assert base[3] == base[4]                                         # every parameter took part in the step
    scaled = run(2.0 * cw)
    assert torch.allclose(scaled[0], base[0], rtol=1e-6, atol=0) and torch.allclose(scaled[1], base[1], rtol=1e-6, atol=0)
    assert ((scaled[2] - base[2]).norm() / base[2].norm()).item() < 1e-6
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(5)).cuda()
    shuffled = run(cw, perm)
    assert torch.allclose(shuffled[0], base[0], rtol=1e-5, atol=0) and torch.allclose(shuffled[1], base[1], rtol=1e-5, atol=0)
    cos = torch.nn.functional.cosine_similarity(shuffled[2], base[2], dim=0).item()
    print(f'batch 32 invariants: class-weight x2 gradient diff {((scaled[2] - base[2]).norm() / base[2].norm()).item():.2e}; '
          f'batch permutation: loss diff {((shuffled[1] - base[1]).abs() / base[1].abs()).item():.2e}, gradient cosine '
          f'{cos:.6f}, |g| ratio {(shuffled[2].norm() / base[2].norm()).item():.5f}')
    assert cos > 0.999, cos
    assert abs((shuffled[2].norm() / base[2].norm()).item() - 1) < 2e-2


def test_configs1_forward_only_batch16_properties():
    """BASELINE configs[1] at its OWN size — fwd-only eval, batch 16, 480x640, gate forced on (static fuse), both encoder
    streams, BatchNorm folded into the convolutions — where the oracle is too slow to run: size-independent properties.
      * eval mode is per-sample: the batch-16 forward equals two batch-8 forwards of its halves, sample by sample;
      * a batch permutation permutes the output rows and nothing else;
      * static fuse == the dynamic model with every gate decision "fuse" (ESANet.forward vs SkipGateESANet.baseline);
      * the first two samples agree with the CPU oracle (2e-4, the eval-logit bar)."""
    from dynmm_amd.nn.esanet import ESANet
    from oracle import dynmm_oracle as O
    h, w, n = 480, 640, 16
    rgb, depth = synth.synth_inputs(n, h, w, seed=2468, device='cuda')
    m = hip_model('P_se', h, w, seed=0)
    m.eval()
    m.baseline = True
    m.dual_stream = True
    with torch.no_grad():
        out = m(rgb, depth, test=True)
        halves = torch.cat([m(rgb[:8], depth[:8], test=True), m(rgb[8:], depth[8:], test=True)])
        perm = torch.tensor([5, 0, 11, 3, 15, 8, 1, 12, 7, 2, 14, 9, 4, 13, 6, 10], device='cuda')
        out_p = m(rgb[perm].contiguous(), depth[perm].contiguous(), test=True)
    assert out.shape == (n, 40, h, w) and bool(torch.isfinite(out).all())
    scale = out.abs().max().item()
    # (tile boundaries move with the batch size: the same sums in the same order per pixel, so these are exact or 1 ulp)
    assert (out - halves).abs().max().item() <= 2e-6 * scale
    assert (out_p - out[perm]).abs().max().item() <= 2e-6 * scale
    e = ESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
               encoder_block='NonBottleneck1D', channels_decoder=[128, 128, 128], nr_decoder_blocks=[3, 3, 3],
               pretrained_on_imagenet=False, fuse_depth_in_rgb_encoder='SE-add', upsampling='learned-3x3-zeropad')
    e.load_state_dict({k: v for k, v in m.state_dict().items() if 'gate' not in k})
    e = e.cuda().eval()
    with torch.no_grad():
        out_e = e(rgb, depth)
    assert (out_e - out).abs().max().item() <= 2e-6 * scale
    sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
    with torch.no_grad():
        ref = O.forward(sd, rgb[:2].cpu(), depth[:2].cpu(), Hh.CFGS['P_se'], test=True, baseline=True)
    assert Hh.rel_err(out[:2].cpu(), ref) < LOGIT_TOL


def test_configs3_hard_gate_compaction_at_its_own_size():
    """BASELINE configs[3] per GPU at its FULL size (batch 32, 480x640, hard gates, the uniform synthetic branch distribution
    k = n mod 5 the bench line is quoted on, `compact_train` on), where the oracle is too slow:
      * the real training step (BatchNorm batch statistics, weighted 4-scale CE + FLOP loss): stage batches 25/18/12/6,
        finite losses and gradients, EVERY parameter receives a gradient (also the depth stages that ran on a 6-sample
        prefix and the gate, through its straight-through term);
      * with BatchNorm in eval mode — the only per-batch coupling of the depth path — the compacted forward and backward
        equal the dense ones on every non-gate parameter (2e-4): sorted-prefix stages, pass-through of skipped samples,
        permutation and un-permutation at 480x640 tile counts, not only at 96x128."""
    from dynmm_amd import engine
    h, w, n = 480, 640, 32
    branches = [i % 5 for i in range(n)]
    rgb, depth = synth.synth_inputs(n, h, w, seed=21, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=50 + s, device='cuda').to(torch.uint8) for s in (1, 8, 16, 32)]
    m = hip_model('P_se', h, w, seed=3)
    m.train()
    m.hard_gate, m.temp = True, 0.7
    m.branch_override, m.compact_train = branches, True
    step = engine.TrainStep(m, np.linspace(0.5, 2.0, 40).astype(np.float32), lr=0.0, loss_ratio=0.5, flop_budget=0.0)
    step._body(rgb, depth, labels)
    torch.cuda.synchronize()
    assert m.last_stage_batch == [25, 18, 12, 6]
    assert torch.isfinite(step.last['losses']).all() and torch.isfinite(step.last['total']).all()
    grads = torch.cat([p.grad.detach().flatten() for p in m.parameters()])
    assert torch.isfinite(grads).all()
    assert len(step._touched) == sum(1 for _ in m.parameters())
    dead = [k for k, p in m.named_parameters() if p.grad.abs().max().item() == 0 and 'gate_layer.conv' not in k]
    assert not dead, dead[:8]
    del step, m, grads
    torch.cuda.empty_cache()

    res = {}
    probe = None
    for compact in (False, True):
        m = hip_model('P_se', h, w, seed=3)
        m.eval()                                   # BN: running statistics; gradients are still recorded below
        m.hard_gate, m.temp = True, 0.7
        m.branch_override, m.compact_train = branches, compact
        out, lf = m(rgb, depth)
        assert (m.last_stage_batch == [25, 18, 12, 6]) if compact else (m.last_stage_batch is None)
        if probe is None:
            probe = Hh.grad_probe((2, 40, 96, 128), 'c').cuda().repeat(16, 1, 5, 5)       # a fixed non-uniform seed gradient
        (out * probe).mean().backward()
        torch.cuda.synchronize()
        res[compact] = (out.detach()[:, :, ::8, ::8].clone(), lf.detach(),
                        {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
        del out, m
        torch.cuda.empty_cache()
    (od, ld, gd), (oc, lc, gc) = res[False], res[True]
    assert Hh.rel_err(oc.cpu(), od.cpu()) < 1e-5 and abs(lc.item() - ld.item()) < 1e-6
    gmax = max(v.abs().max().item() for v in gd.values())
    upstream = ('gate', 'encoder_rgb.conv1', 'encoder_rgb.bn1', 'encoder_depth.conv1', 'encoder_depth.bn1', 'se_layer0')
    checked, worst = 0, (0.0, None)
    for k, g in gd.items():
        if any(u in k for u in upstream) or g.abs().max().item() < 1e-6 * gmax:
            continue
        e = _rl2(gc[k].cpu(), g.cpu())
        worst = max(worst, (e, k))
        assert e < 2e-4, (k, e)
        checked += 1
    print(f'configs[3] batch 32: compacted vs dense non-gate gradients, worst rel-L2 {worst[0]:.2e} ({worst[1]}), {checked} tensors')
    assert checked > 400
