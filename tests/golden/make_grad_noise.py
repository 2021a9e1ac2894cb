"""Calibration of the fp32 gradient-noise floor: how far apart are CORRECT fp32 evaluations of the same
train step?  The CPU oracle (a restatement of the reference, pinned to it by tests/test_oracle_golden.py) is run in fp32
under several summation orders — intra-op thread counts 1 / 4 / 8, the oneDNN vs native convolution back ends, and (round 4)
the three-tap / 3x3 convolutions evaluated in the 1-D Winograd F(2,3) form (tests/wino_eval.py: the form the HIP path
computes them in, forward and backward) — and each
draw's per-tensor gradient error against the fp64 ground truth is summarised (median / 95th percentile / maximum over the
parameter tensors, cosine deficit of the full gradient).  The GPU tests hold the HIP path to the observed RANGE x 1.25
instead of a multiple of one draw.

    python tests/golden/make_grad_noise.py           # writes tests/golden/grad_noise.npz   (oracle only; ~5 min on 8 cores)

Points: (a) the reference's own N = 8 fixture (config P, 160x192; fp64 gradients = the reference's, 128-element samples),
(b) BASELINE configs[2]'s resolution, 480x640, batch 2 (fp64 = the oracle's), (c) the small reference fixtures
(model_<cfg>_<HxW>.npz, train modes): worst deviation of a per-parameter gradient NORM and of a stored full gradient
tensor from the reference's fp32 values — what GRAD_NORM_TOL / GRAD_FULL_TOL of tests/test_hip_model.py are set from,
(d) the two-step training fixture (two_steps())."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from dynmm_amd import synth                     # noqa: E402
from oracle import dynmm_oracle as O            # noqa: E402
from tests import helpers as Hh                 # noqa: E402
from tests import wino_eval as WE               # noqa: E402
import contextlib                                # noqa: E402

# (threads, oneDNN convolutions, Winograd-form three-tap convolutions)
DRAWS = [(1, True, False), (4, True, False), (8, True, False), (1, False, False), (4, False, False), (8, False, False),
         (1, True, True), (8, True, True), (8, False, True)]


def _form(wino):
    return WE.winograd_convolutions() if wino else contextlib.nullcontext()


def sample(t, limit=128):
    f = t.detach().reshape(-1)
    return f[::max(1, -(-f.numel() // limit))]


def rl2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def step(h, w, n, cw, ratio, dtype, threads, mkldnn, wino=False):
    torch.set_num_threads(threads)
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s) for s in (1, 8, 16, 32)]
    sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
    sd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    with torch.backends.mkldnn.flags(enabled=mkldnn), _form(wino):
        outs, lf = O.forward(sd, rgb.to(dtype), depth.to(dtype), Hh.CFGS['P_se'], training=True, temp=1.0)
        losses = O.cross_entropy_2d(outs, labels, torch.from_numpy(cw).to(dtype))
        total = sum(losses) + ratio * torch.clamp(lf, min=0.0)
        total.backward()
    return {k: p.grad.detach() for k, p in params.items()}


def summarise(draws, g64, names):
    rows = []
    for grads in draws:
        e = np.array([rl2(grads[nm], g64[nm]) for nm in names])
        a = torch.cat([grads[nm].double().flatten() for nm in names])
        b = torch.cat([g64[nm].double().flatten() for nm in names])
        rows.append([np.median(e), np.percentile(e, 95), e.max(), 1.0 - torch.nn.functional.cosine_similarity(a, b, dim=0).item()])
    return np.array(rows)


SMALL = (('P_se', 96, 128, 'train_soft'), ('P_se', 96, 128, 'train_hard'), ('S_se', 96, 128, 'train_soft'),
         ('P_se', 160, 192, 'train_soft'), ('R18_se', 96, 128, 'train_soft'), ('P_add', 96, 128, 'train_soft'),
         ('R50_se', 96, 128, 'train_soft'))


def small_fixtures():
    rows, tags = [], []
    for cfg, h, w, mode in SMALL:
        g = np.load(os.path.join(HERE, f'model_{cfg}_{h}x{w}.npz'))
        hh, ww, n, _ = [int(v) for v in g['meta']]
        names = [str(s_) for s_ in g[f'{mode}/grad_names']]
        ref = g[f'{mode}/grad_norms']
        worst_norm = worst_full = 0.0
        for threads, mk, wino in ((1, True, False), (8, True, False), (1, False, False), (8, False, False),
                                  (1, True, True), (8, True, True), (8, False, True)):
            torch.set_num_threads(threads)
            rgb, depth = synth.synth_inputs(n, hh, ww, seed=1234)
            sd = Hh.filled_state_dict(Hh.CFGS[cfg], seed=0)
            params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
            with torch.backends.mkldnn.flags(enabled=mk), _form(wino):
                outs, lf = O.forward(sd, rgb, depth, Hh.CFGS[cfg], **Hh.MODE_KW[mode])
                Hh.train_loss(outs, lf).backward()
            norms = np.array([0.0 if params[k].grad is None else params[k].grad.norm().item() for k in names])
            worst_norm = max(worst_norm, float((np.abs(norms - ref) / np.maximum(ref, 1e-2 * ref.max())).max()))
            for k in g.files:
                if k.startswith(f'{mode}/grad:') and np.abs(g[k]).max() > 1e-6:
                    worst_full = max(worst_full, Hh.rel_err(params[k.split('grad:')[1]].grad, g[k]))
        rows.append([worst_norm, worst_full])
        tags.append(f'{cfg} {h}x{w} {mode}')
        print(tags[-1], rows[-1], flush=True)
    return np.array(tags), np.array(rows)


def two_steps():
    """(d) tests/test_engine.py::test_two_train_steps_match_reference: the reference's two SGD-Nesterov steps at batch 2, 96x128
    (train_steps_P_se.npz), replayed by the oracle under the evaluation orders above: worst relative deviation of a parameter
    NORM from the reference's after the second update, for the SE-layer tensors and for all others."""
    g = np.load(os.path.join(HERE, 'train_steps_P_se.npz'))
    h, w, n = [int(v) for v in g['meta']]
    lr, wd, mom, ratio, budget, temp = [float(v) for v in g['hyper']]
    names = [str(k) for k in g['param_names']]
    ref = g['param_norms']
    se = np.array(['se_layer' in nm for nm in names])
    worst = np.zeros(2)
    for threads, mk, wino in ((8, True, False), (1, True, False), (4, True, False), (8, False, False), (1, False, False),
                              (1, True, True), (8, True, True), (8, False, True)):
        torch.set_num_threads(threads)
        rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
        labels = [synth.synth_labels(n, h // s_, w // s_, seed=300 + s_) for s_ in (1, 8, 16, 32)]
        sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
        params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
        opt = torch.optim.SGD(list(params.values()), lr=lr, weight_decay=wd, momentum=mom, nesterov=True)
        with torch.backends.mkldnn.flags(enabled=mk), _form(wino):
            for _ in range(2):
                opt.zero_grad()
                outs, lf = O.forward(sd, rgb, depth, Hh.CFGS['P_se'], training=True, temp=temp)
                losses = O.cross_entropy_2d(outs, labels, torch.from_numpy(g['cw']))
                (sum(losses) + ratio * torch.clamp(lf - budget, min=0.0)).backward()
                opt.step()
        norms = np.array([sd[k].detach().double().norm().item() for k in names])
        rel = np.abs(norms - ref) / np.maximum(ref, 1e-3)
        worst = np.maximum(worst, [rel[~se].max(), rel[se].max()])
        print('two steps', threads, mk, wino, rel[~se].max(), rel[se].max(), flush=True)
    return worst


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'two_steps':  # add / refresh part (d) only
        blob = dict(np.load(os.path.join(HERE, 'grad_noise.npz')))
        blob['two_steps'] = two_steps()
        np.savez_compressed(os.path.join(HERE, 'grad_noise.npz'), **blob)
        print('two_steps (others, SE):', blob['two_steps'])
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'small':      # add / refresh part (c) only
        blob = dict(np.load(os.path.join(HERE, 'grad_noise.npz')))
        blob['small_fixtures'], blob['small'] = small_fixtures()
        np.savez_compressed(os.path.join(HERE, 'grad_noise.npz'), **blob)
        return
    blob = {'draws': np.array([f'{t} threads, {"oneDNN" if m else "native"} conv{", Winograd form" if wn else ""}' for t, m, wn in DRAWS]),
            'columns': np.array(['median', 'p95', 'max', 'cosine_deficit'])}
    # (a) the reference's N = 8 fixture: fp64 truth = the reference's own fp64 run
    g = np.load(os.path.join(HERE, 'train_n8_P_se_160x192.npz'))
    h, w, n, _ = [int(v) for v in g['meta']]
    names = [str(s) for s in g['grad_names']]
    g64 = {nm: torch.from_numpy(g['f64/g:' + nm]) for nm in names}
    gmax = max(v.abs().max().item() for v in g64.values())
    names = [nm for nm in names if g64[nm].abs().max().item() >= 1e-5 * gmax]
    draws = []
    for t, mk, wn in DRAWS:
        gr = step(h, w, n, g['cw'].astype(np.float32), float(g['ratio']), torch.float32, t, mk, wn)
        draws.append({nm: sample(gr[nm]) for nm in names})
        print('n8', t, mk, wn, flush=True)
    blob['n8'] = summarise(draws, g64, names)
    blob['n8_reference_fp32'] = summarise([{nm: torch.from_numpy(g['f32/g:' + nm]) for nm in names}], g64, names)[0]
    # (b) 480x640, batch 2
    cw = np.linspace(0.5, 2.0, 40).astype(np.float32)
    g64 = step(480, 640, 2, cw, 0.5, torch.float64, 8, True)
    gmax = max(v.abs().max().item() for v in g64.values())
    names = [nm for nm, v in g64.items() if v.abs().max().item() >= 1e-5 * gmax]
    draws = []
    for t, mk, wn in DRAWS:
        draws.append(step(480, 640, 2, cw, 0.5, torch.float32, t, mk, wn))
        print('480x640', t, mk, wn, flush=True)
    blob['b2_480x640'] = summarise(draws, g64, names)
    blob['two_steps'] = two_steps()
    blob['small_fixtures'], blob['small'] = small_fixtures()
    # (no ratchet: the table is what THIS run of the fixed draw list shows.  Fixtures whose spread exceeds 0.08 — a flipped
    # ReLU / arg-max decision between correct fp32 evaluations — are not compared with the reference's fp32 gradients at all;
    # tests/test_hip_model.py compares them with the fp64 oracle at equal decisions instead, at a fixed 0.02)
    np.savez_compressed(os.path.join(HERE, 'grad_noise.npz'), **blob)
    for k in ('n8', 'n8_reference_fp32', 'b2_480x640'):
        print(k, np.array2string(blob[k], precision=4))


if __name__ == '__main__':
    main()
