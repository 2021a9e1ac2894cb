"""SkipESANet — the per-stage Gumbel-gated variant (model_skip_mod.py).

CPU (-m "not gpu"): the oracle's restatement is pinned to the fixture the reference itself produced with its
Gumbel draws injected (tests/golden/make_goldens.py: skip_fixture), and the build's module reproduces the
reference's state_dict.
GPU (-m gpu): the fused gate/blend op against the oracle (forward, backward, the Philox draw), and the whole
model through the C ABI against the fixture and the oracle, with the same injected noise."""
import os

import numpy as np
import pytest
import torch

from dynmm_amd import synth
from oracle import dynmm_oracle as O
from tests import helpers as Hh

TOL = 2e-5
CFG = Hh.CFGS['P_se']


def skip_module(h=96, w=128, temp=1.0, rule=(2, 2, 2, 2)):
    from dynmm_amd.nn.net_skip import SkipESANet
    return SkipESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34',
                      encoder_block='NonBottleneck1D', nr_decoder_blocks=[3, 3, 3],
                      fuse_depth_in_rgb_encoder='SE-add', temp=temp, block_rule=list(rule))


def filled_sd(seed=0):
    sd = {k: v.clone() for k, v in skip_module().state_dict().items()}
    synth.fill_state_dict(sd, seed)
    return sd


def modes(g):
    return sorted({k.split('/')[0] for k in g.files if '/' in k})


def mode_cfg(g, mode):
    training, test, hard, *rule = [int(v) for v in g[f'{mode}/cfg']]
    return bool(training), bool(test), bool(hard), rule, float(g[f'{mode}/temp'])


# ------------------------------------------------------------------------------------------- CPU
def test_skip_state_dict_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'skip_P_96x128.npz'))
    sd = skip_module().state_dict()
    assert list(sd.keys()) == [str(k) for k in g['keys']]
    assert [','.join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g['shapes']]
    assert [str(v.dtype) for v in sd.values()] == [str(s) for s in g['dtypes']]


def test_skip_oracle_matches_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'skip_P_96x128.npz'))
    h, w, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    for mode in modes(g):
        training, test, hard, rule, temp = mode_cfg(g, mode)
        sd = filled_sd()
        noise = [torch.from_numpy(g[f'{mode}/noise{j}']) for j in range(4)]
        det = {}
        if training:
            params = {k: v.requires_grad_(True) for k, v in sd.items()
                      if v.dtype.is_floating_point and 'running_' not in k}
            outs = O.forward_skip(sd, rgb, depth, CFG, noise, training=True, test=test, hard_gate=hard,
                                  temp=temp, block_rule=rule, detail=det)
            loss = Hh.train_loss(outs, torch.zeros(()))
            loss.backward()
            out = outs[0].detach()
            assert abs(loss.item() - float(g[f'{mode}/loss'])) < 1e-4 * max(1, abs(float(g[f'{mode}/loss'])))
            names = [str(s) for s in g[f'{mode}/grad_names']]
            norms = np.array([0.0 if params[nm].grad is None else params[nm].grad.norm().item() for nm in names])
            ref = g[f'{mode}/grad_norms']
            assert np.all(np.abs(norms - ref) <= 5e-4 * np.maximum(ref, 1e-3) + 1e-6), \
                np.max(np.abs(norms - ref) / np.maximum(ref, 1e-3))
            for k in g.files:
                if k.startswith(f'{mode}/grad:'):
                    assert Hh.rel_err(params[k.split('grad:')[1]].grad, g[k]) < 5e-4, k
        else:
            with torch.no_grad():
                out = O.forward_skip(sd, rgb, depth, CFG, noise, test=test, hard_gate=hard, temp=temp,
                                     block_rule=rule, detail=det)
        for j in range(4):
            assert Hh.rel_err(det['weights'][j].detach(), g[f'{mode}/weight{j}']) < TOL, (mode, j)
        assert Hh.rel_err(out[:, :, ::stride, ::stride], g[f'{mode}/strided']) < TOL, mode
        assert Hh.rel_err(out.sum(dim=(2, 3)), g[f'{mode}/csum']) < 1e-4
        assert Hh.rel_err(out.abs().sum(dim=(2, 3)), g[f'{mode}/cabs']) < 1e-4


def test_skip_has_no_cpu_fallback():
    from dynmm_amd.lib import DynmmHipError
    m = skip_module().eval()
    with pytest.raises(DynmmHipError):
        with torch.no_grad():
            m(torch.randn(1, 3, 96, 128), torch.randn(1, 1, 96, 128), test=True)


# ------------------------------------------------------------------------------------------- GPU
def _gate_sd(C, seed):
    r = np.random.Generator(np.random.PCG64([seed, C]))
    C2, Hd = 2 * C, 2 * C // 16

    def t(*shape, scale=1.0):
        return torch.from_numpy((r.standard_normal(size=shape) * scale).astype(np.float32))
    return {'g.se.fc.0.weight': t(Hd, C2, 1, 1, scale=C2 ** -0.5), 'g.se.fc.0.bias': t(Hd, scale=0.1),
            'g.se.fc.2.weight': t(C2, Hd, 1, 1, scale=Hd ** -0.5), 'g.se.fc.2.bias': t(C2, scale=0.1)}


@pytest.mark.gpu
@pytest.mark.parametrize('C,H,W,mode,hard,use_prev', [
    (64, 12, 16, 2, False, True), (64, 12, 16, 2, True, True), (128, 6, 8, 1, False, False),
    (256, 3, 4, 0, True, True), (32, 5, 7, 2, False, False)])
def test_reweigh_fuse_matches_oracle(C, H, W, mode, hard, use_prev):
    from dynmm_amd import ops
    N, temp = 4, 0.6
    r = np.random.Generator(np.random.PCG64([7, C, H]))
    f = lambda *s: torch.from_numpy(r.standard_normal(size=s).astype(np.float32))  # noqa: E731
    rgb, depth = f(N, C, H, W), f(N, C, H, W) * 0.7 + 0.2
    wb = torch.softmax(f(N, 2), 1)
    prev = torch.sigmoid(f(N))
    noise = torch.from_numpy(r.exponential(size=(N, 2)).astype(np.float32))
    g_out, g_w = f(N, C, H, W), f(N, 2)
    sd = _gate_sd(C, 3)

    # oracle (fp64 for a clean reference)
    def run_oracle(dt):
        leaves = [x.to(dt).requires_grad_(True) for x in (rgb, depth, wb, prev)]
        sdo = {k: v.to(dt).requires_grad_(True) for k, v in sd.items()}
        ro, do, wo, po = leaves
        w_next = O.reweigh_gate(sdo, 'g', ro, do, temp, noise.to(dt), hard, po if use_prev else None)
        if mode == 0:
            fuse = ro * 1
        elif mode == 1:
            fuse = ro + do
        else:
            fuse = wo[:, 0].view(-1, 1, 1, 1) * ro + wo[:, 1].view(-1, 1, 1, 1) * (ro + do)
        ((fuse * g_out.to(dt)).sum() + (w_next * g_w.to(dt)).sum()).backward()
        return fuse.detach(), w_next.detach(), leaves, sdo
    fuse_o, w_o, leaves_o, sd_o = run_oracle(torch.float64)

    dev = 'cuda'
    leaves = [x.clone().to(dev).requires_grad_(True) for x in (rgb, depth, wb, prev)]
    params = [sd[k].clone().to(dev).requires_grad_(True) for k in
              ('g.se.fc.0.weight', 'g.se.fc.0.bias', 'g.se.fc.2.weight', 'g.se.fc.2.bias')]
    rg, dg, wg, pg = leaves
    fuse, w_next, aux = ops.reweigh_fuse(rg, dg, wg if mode == 2 else None, mode, params, temp, hard,
                                         pg if use_prev else None, noise.to(dev))
    ((fuse * g_out.to(dev)).sum() + (w_next * g_w.to(dev)).sum()).backward()
    assert Hh.rel_err(fuse.cpu(), fuse_o) < 2e-6
    assert Hh.rel_err(w_next.cpu(), w_o) < 5e-6
    assert torch.equal(aux[:, 4:].cpu(), noise)
    for got, ref, name in zip(leaves, leaves_o, ('rgb', 'depth', 'wblend', 'prev')):
        if ref.grad is None or (name == 'wblend' and mode != 2) or (name == 'prev' and not use_prev):
            assert got.grad is None or float(got.grad.abs().max()) == 0.0 or name in ('rgb', 'depth')
            continue
        assert Hh.rel_err(got.grad.cpu(), ref.grad) < 2e-5, name
    for got, k in zip(params, ('g.se.fc.0.weight', 'g.se.fc.0.bias', 'g.se.fc.2.weight', 'g.se.fc.2.bias')):
        assert Hh.rel_err(got.grad.cpu(), sd_o[k].grad) < 5e-5, k


@pytest.mark.gpu
def test_reweigh_philox_noise_is_exponential_and_reproducible():
    from dynmm_amd import ops
    N, C = 4096, 32
    x = torch.randn(N, C, 2, 2, device='cuda')
    params = [v.cuda() for v in _gate_sd(C, 5).values()]
    ops.manual_seed(123)
    _, w1, a1 = ops.reweigh_fuse(x, x, None, 1, params, 1.0, True)
    _, w2, a2 = ops.reweigh_fuse(x, x, None, 1, params, 1.0, True)
    ops.manual_seed(123)
    _, w3, a3 = ops.reweigh_fuse(x, x, None, 1, params, 1.0, True)
    e1, e2 = a1[:, 4:].double().cpu(), a2[:, 4:].double().cpu()
    assert torch.equal(a1, a3) and torch.equal(w1, w3)         # same seed + call index -> same draw
    assert not torch.equal(e1, e2)                              # the call counter advances the stream
    assert (e1 > 0).all() and torch.isfinite(e1).all()
    for e in (e1, e2):                                          # Exp(1): mean 1, var 1, median ln 2
        assert abs(e.mean().item() - 1.0) < 0.05 and abs(e.var().item() - 1.0) < 0.12
        assert abs(e.median().item() - np.log(2)) < 0.05
    assert abs(np.corrcoef(e1[:, 0], e1[:, 1])[0, 1]) < 0.05
    # hard one-hots; P(branch 0) = sigmoid(2w-1) for logits [w, 1-w] at temp 1 (Gumbel-max trick)
    w = a1[:, 0].double().cpu()
    assert set(torch.unique(w1).tolist()) <= {0.0, 1.0}
    assert abs(w1[:, 0].double().mean().item() - torch.sigmoid(2 * w - 1).mean().item()) < 0.03


def _hip_skip(temp, rule, seed=0):
    m = skip_module(temp=temp, rule=rule)
    synth.fill_state_dict(m.state_dict(), seed)
    return m.cuda()


@pytest.mark.gpu
def test_skip_model_matches_reference_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, 'skip_P_96x128.npz'))
    h, w, n, stride = [int(v) for v in g['meta']]
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    for mode in modes(g):
        training, test, hard, rule, temp = mode_cfg(g, mode)
        m = _hip_skip(temp, rule)
        m.train(training)
        m.compact = mode != 'eval_test'        # eval_test: dense reference semantics; the rest never compacts
        m.hard_gate = hard
        m.gumbel_noise = [torch.from_numpy(g[f'{mode}/noise{j}']).cuda() for j in range(4)]
        m.start_weight()
        if training:
            outs = m(rgb, depth, test=test)
            loss = Hh.train_loss(outs, torch.zeros((), device='cuda'))
            loss.backward()
            out = outs[0].detach()
            assert abs(loss.item() - float(g[f'{mode}/loss'])) < 2e-3 * max(1, abs(float(g[f'{mode}/loss'])))
            for i, o in enumerate(outs[1:]):
                assert Hh.rel_err(o.detach().cpu(), g[f'{mode}/side{i}']) < 1e-3
            names = [str(s) for s in g[f'{mode}/grad_names']]
            pr = dict(m.named_parameters())
            norms = np.array([0.0 if pr[nm].grad is None else pr[nm].grad.norm().item() for nm in names])
            ref = g[f'{mode}/grad_norms']
            big = ref > 1e-3 * ref.max()
            print(f'skip {mode}: worst gradient-norm deviation from the fixture {np.max(np.abs(norms - ref)[big] / ref[big]):.4f}')
            assert np.max(np.abs(norms - ref)[big] / ref[big]) < 0.05      # (measured 0.005; fp32 conditioning: see test_hip_model)
            assert np.all((ref == 0) == (norms == 0))                         # unused params get no gradient
        else:
            with torch.no_grad():
                out = m(rgb, depth, test=test)
        for j in range(4):
            assert Hh.rel_err(m.weight_list[j], g[f'{mode}/weight{j}']) < (1e-3 if training else 1e-4), (mode, j)
        tol = 1e-3 if training else 2e-4
        assert Hh.rel_err(out[:, :, ::stride, ::stride].cpu(), g[f'{mode}/strided']) < tol, mode
        assert Hh.rel_err(out.sum(dim=(2, 3)).cpu(), g[f'{mode}/csum']) < tol


@pytest.mark.gpu
def test_skip_model_compaction_is_exact():
    """Hard-gate inference: running the depth encoder only on the samples that still fuse (chained weights
    make a skip permanent) gives the dense result, with the same gate decisions."""
    h, w, n = 96, 128, 8
    rgb, depth = synth.synth_inputs(n, h, w, seed=4321, device='cuda')
    r = np.random.Generator(np.random.PCG64(17))
    m = _hip_skip(1.0, (2, 2, 2, 2)).eval()
    # strong noise contrast so that the eight samples take different exits
    m.gumbel_noise = [torch.from_numpy(np.exp(r.uniform(-2.5, 2.5, size=(n, 2))).astype(np.float32)).cuda()
                      for _ in range(4)]
    outs, wl = {}, {}
    for compact in (False, True):
        m.compact = compact
        m.start_weight()
        with torch.no_grad():
            outs[compact] = m(rgb, depth, test=True)
        wl[compact] = [t.clone() for t in m.weight_list]
        if compact:
            sizes = m.last_stage_batch
    assert sizes == sorted(sizes, reverse=True) and sizes[-1] < sizes[0] <= n and sizes[0] > 0, sizes
    for j in range(4):
        assert torch.allclose(wl[True][j], wl[False][j], atol=1e-6), j
    assert Hh.rel_err(outs[True].cpu(), outs[False].cpu()) < 1e-5


@pytest.mark.gpu
def test_skip_model_gate_gradients_vs_fp64_oracle():
    """Gate-parameter gradients of the whole model (frozen backbone case of train.py:139-141: only
    parameters with 'gate' in their name train) against the fp64 oracle with the same noise."""
    h, w, n, temp = 96, 128, 2, 0.7
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    r = np.random.Generator(np.random.PCG64(11))
    noise = [torch.from_numpy(r.exponential(size=(n, 2)).astype(np.float32)) for _ in range(4)]

    def oracle(dt):
        sd = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in filled_sd().items()}
        params = {k: v.requires_grad_(True) for k, v in sd.items() if 'gate' in k and 'linear' not in k}
        out = O.forward_skip(sd, rgb.to(dt), depth.to(dt), CFG, [e.to(dt) for e in noise], temp=temp)
        (out * Hh.grad_probe(tuple(out.shape), 's0').to(dt)).mean().backward()
        return out.detach(), params
    out64, p64 = oracle(torch.float64)
    out32, p32 = oracle(torch.float32)

    # The fp32 conditioning of this point is MEASURED, not assumed: the pass is run under both implicit-GEMM generations
    # (each pinned op by op against torch in tests/test_hip_ops.py; they sum a 1x3 convolution in different orders).
    # Where the two agree the bar is the oracle's own fp32-vs-fp64 error x 3; where they do not — ONE ReLU decision at a
    # pre-activation of 1e-6 differs between them in a decoder map of 6x8 pixels and moves the stage-1 gate gradient by
    # 17 % (scratch/v5_ab5.py) — two correct fp32 implementations disagree, and the bar is twice their disagreement.
    from dynmm_amd import lib as L
    runs = {}
    for gen in (0, 1):
        L.load().dynmm_debug_set_igemm_v5(gen)
        try:
            m = _hip_skip(temp, (2, 2, 2, 2)).eval()        # eval-mode BN: isolates the gate path
            m.freeze()
            m.gumbel_noise = [e.cuda() for e in noise]
            out = m(rgb.cuda(), depth.cuda())
            (out * Hh.grad_probe(tuple(out.shape), 's0').cuda()).mean().backward()
            torch.cuda.synchronize()
        finally:
            L.load().dynmm_debug_set_igemm_v5(-1)
        assert Hh.rel_err(out.detach().cpu(), out64) < 2e-4
        runs[gen] = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    strict = 0
    for k, v in p64.items():
        assert k in runs[0] and k in runs[1], k
        spread = Hh.rel_err(runs[0][k], runs[1][k])
        e_f32 = Hh.rel_err(p32[k].grad, v.grad)
        bar = max(3 * e_f32, 2e-4, 2 * spread)
        strict += bar == max(3 * e_f32, 2e-4)
        for gen in (0, 1):
            e_hip = Hh.rel_err(runs[gen][k], v.grad)
            assert e_hip < bar, (k, gen, e_hip, e_f32, spread)
    assert strict >= 4, strict          # at least one gate's four parameter tensors held to the fp32-oracle bar
    for k, prm in m.named_parameters():
        if 'gate' not in k:
            assert prm.grad is None, k
