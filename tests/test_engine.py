"""Callers of the hot path: schedules on CPU; optimisation steps and the eval
protocol on the GPU against goldens produced by the reference's own loss / optimizer / model."""
import os
import warnings

import numpy as np
import pytest
import torch

from dynmm_amd import schedules, synth
from tests import helpers as Hh


def test_one_cycle_matches_torch():
    warnings.filterwarnings('ignore')
    for epochs, max_lr in ((500, 0.04), (30, 0.01)):
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=max_lr)
        sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=max_lr, total_steps=epochs, div_factor=25,
                                                  pct_start=0.1, anneal_strategy='cos', final_div_factor=1e4)
        for e in range(epochs):
            sch.step(e)                                         # train.py:267: stepped per epoch
            assert abs(opt.param_groups[0]['lr'] - schedules.one_cycle_lr(e, epochs, max_lr)) < 1e-12


def test_one_cycle_momentum_matches_torch():
    """OneCycleLR's default cycle_momentum=True rewrites SGD's momentum / Adam's beta1 at every scheduler step
    (train.py:120-128 uses the default): the driver must feed the same value to the fused optimizer."""
    warnings.filterwarnings('ignore')
    for epochs in (500, 30):
        for make, key in ((lambda p: torch.optim.SGD(p, lr=0.01, momentum=0.9, nesterov=True), 'momentum'),
                          (lambda p: torch.optim.Adam(p, lr=0.01, betas=(0.9, 0.999)), 'betas')):
            opt = make([torch.nn.Parameter(torch.zeros(1))])
            sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=0.01, total_steps=epochs, div_factor=25,
                                                      pct_start=0.1, anneal_strategy='cos', final_div_factor=1e4)
            for e in range(epochs):
                sch.step(e)
                mom = opt.param_groups[0][key]
                mom = mom[0] if key == 'betas' else mom
                assert abs(mom - schedules.one_cycle_momentum(e, epochs)) < 1e-12, (epochs, e)


def test_flat_optimizer_touched_ranges():
    """Host logic of the fused flat optimizers: only parameters that received a gradient are updated (torch.optim
    skips `.grad is None`), as merged element ranges per parameter group."""
    from dynmm_amd import engine
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 8, 3, 4)]
    fp = engine.FlatParameters(ps)
    assert fp.flat.numel() == 20 and fp.span[id(ps[3])] == (0, 4) and fp.span[id(ps[0])] == (15, 20)
    for q in ps:                                         # parameters are views of the flat buffer
        lo, hi = fp.span[id(q)]
        assert q.data_ptr() == fp.flat.data_ptr() + 4 * lo and torch.equal(q.detach().flatten(), fp.flat[lo:hi])

    class Dummy(engine._FlatOptimizer):
        pass
    opt = Dummy(fp, torch.zeros(20), {'gate': [ps[3]], 'rest': ps[:3]})
    assert opt.plan(None) == [(0, [(0, 4)]), (1, [(4, 20)])]
    assert opt.plan({id(ps[0]), id(ps[2])}) == [(1, [(4, 7), (15, 20)])]
    assert opt.plan({id(ps[3]), id(ps[2]), id(ps[1])}) == [(0, [(0, 4)]), (1, [(4, 15)])]
    assert opt.plan(set()) == []


def test_temperature_schedule_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'ops.npz'))
    t = schedules.ExpDecayTemp(1.0, 0.001, 300)
    for e, v in zip(g['temp/epochs'], g['temp/values']):
        assert abs(t.get_t(int(e)) - v) < 1e-12
    assert schedules.scaled_lr(0.01, 8) == 0.01 and abs(schedules.scaled_lr(0.01, 256) - 0.32) < 1e-12


@pytest.mark.gpu
def test_two_train_steps_vs_fp64_oracle_at_equal_decisions(golden_dir):
    """The two SGD-Nesterov steps of the reference fixture (train_steps_P_se.npz: batch 2, 96x128) against the fp64 oracle at EQUAL
    decisions (tests/test_hip_model.py: test_model_gradients_vs_fp64_oracle_at_equal_decisions): the HIP pass's ReLU decisions are
    imposed on the oracle at its near-ties and must equal the oracle's elsewhere.  Each step is checked from a COMMON state: the
    oracle takes step 0 from the fixture's weights and step 1 from the HIP path's own weights and momentum buffers after step 0
    (in fp64) — a gradient tensor of this net carries ~1e-2 of fp32 conditioning noise which one update turns
    into a weight perturbation that moves decisions far outside any rounding band, so the second forward of two free-running
    implementations is not comparable decision for decision.  (That free-running comparison is the golden test below, held to
    bands calibrated on the fp32 oracle's own spread.)  Per step: the four losses and the total to 1e-4, every parameter norm
    after the update to 1e-3 of the fp64 result and every parameter TENSOR to 2e-3 in relative L2 (measured 1.4e-4 / 3.2e-4; momentum, Nesterov look-ahead and weight decay of step 1 included)."""
    from dynmm_amd import engine, ops
    from dynmm_amd.nn.net import SkipGateESANet
    from oracle import dynmm_oracle as O
    from tests import helpers as Hh
    from tests.test_hip_blocks import hip_relu_decisions
    from tests.test_hip_model import DECISION_BAND
    g = np.load(os.path.join(golden_dir, 'train_steps_P_se.npz'))
    h, w, n = [int(v) for v in g['meta']]
    lr, wd, mom, ratio, budget, temp = [float(v) for v in g['hyper']]
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().train()
    m.temp, m.hard_gate = temp, False
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s) for s in (1, 8, 16, 32)]
    step = engine.TrainStep(m, g['cw'], lr=lr, momentum=mom, weight_decay=wd, loss_ratio=ratio, flop_budget=budget,
                            use_graph=False, multi_stream=False)
    hp = dict(m.named_parameters())
    sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
    sd = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    names = list(params)
    opt = torch.optim.SGD(list(params.values()), lr=lr, weight_decay=wd, momentum=mom, nesterov=True)
    se = np.array(['se_layer' in nm for nm in names])
    for s in range(2):
        if s:                                    # common state: the HIP path's weights, running statistics and momentum
            with torch.no_grad():
                for k, v in m.state_dict().items():
                    sd[k].copy_(v.cpu().to(sd[k].dtype))
                for k in names:
                    lo, hi = step.flatp.span[id(hp[k])]
                    opt.state[params[k]]['momentum_buffer'].copy_(step.opt.buf[lo:hi].view_as(hp[k]).cpu().double())
        ops.ACT_TRACE = []
        try:
            out = step(rgb.cuda(), depth.cuda(), [t.cuda() for t in labels])
        finally:
            trace, ops.ACT_TRACE = ops.ACT_TRACE, None
        opt.zero_grad()
        with hip_relu_decisions(trace, tau=DECISION_BAND) as census:
            outs, lf = O.forward(sd, rgb.double(), depth.double(), Hh.CFGS['P_se'], training=True, temp=temp)
        assert census['outside_band'] == 0 and not any(census['queues'].values()), (s, census['outside_band'])
        losses = O.cross_entropy_2d(outs, labels, torch.from_numpy(g['cw']).double())
        total = sum(losses) + ratio * torch.clamp(lf - budget, min=0.0)
        total.backward()
        opt.step()
        assert np.allclose(out['losses'].cpu().numpy(), [float(v.detach()) for v in losses], rtol=1e-4), (s, out['losses'], losses)
        assert abs(out['total'].item() - total.item()) < 1e-4 * abs(total.item()), s
        new_sd = m.state_dict()
        ref = np.array([sd[k].detach().norm().item() for k in names])
        got = np.array([new_sd[k].double().norm().item() for k in names])
        rel = np.abs(got - ref) / np.maximum(ref, 1e-3)
        full = max(((new_sd[k].double().cpu() - sd[k].detach()).norm() / sd[k].detach().norm().clamp_min(1e-3)).item() for k in names)
        print(f'step {s} at equal decisions ({census["imposed"]} imposed): worst parameter-norm deviation {rel[~se].max():.2e} '
              f'(SE layers {rel[se].max():.2e}); worst full-tensor relative L2 {full:.2e}')
        bad = np.nonzero(rel >= 1e-3)[0]
        assert bad.size == 0 and full < 2e-3, (s, full, [(names[i], got[i], ref[i], rel[i]) for i in bad[:8]])


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
def test_two_train_steps_match_reference(golden_dir, use_graph):
    from dynmm_amd import engine
    from dynmm_amd.nn.net import SkipGateESANet
    g = np.load(os.path.join(golden_dir, 'train_steps_P_se.npz'))
    h, w, n = [int(v) for v in g['meta']]
    lr, wd, mom, ratio, budget, temp = [float(v) for v in g['hyper']]
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().train()
    m.temp, m.hard_gate = temp, False
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s, device='cuda') for s in (1, 8, 16, 32)]
    step = engine.TrainStep(m, g['cw'], lr=lr, momentum=mom, weight_decay=wd, loss_ratio=ratio, flop_budget=budget,
                            use_graph=use_graph)
    for s, tol in ((0, 2e-3), (1, 4e-2)):      # step 1 sits behind one ill-conditioned update
        if not use_graph:
            m.start_weight()                     # host-side gate bookkeeping (a D2H copy): eager only
        out = step(rgb, depth, labels)
        losses = out['losses'].cpu().numpy()
        assert np.allclose(losses, g[f'step{s}/losses'], rtol=tol), (s, losses, g[f'step{s}/losses'])
        assert abs(out['loss_flop'].item() - float(g[f'step{s}/loss_flop'])) < tol
        assert abs(out['total'].item() - float(g[f'step{s}/total'])) < tol * float(g[f'step{s}/total'])
        if not use_graph:                        # weight_list is host-side bookkeeping, not captured in a graph
            assert np.allclose(m.weight_list.numpy(), g[f'step{s}/weight'], atol=10 * tol)
        m.end_weight()
    sd = m.state_dict()
    names = [str(k) for k in g['param_names']]
    norms = np.array([sd[k].double().norm().item() for k in names])
    rel = np.abs(norms - g['param_norms']) / np.maximum(g['param_norms'], 1e-3)
    worst = np.argsort(-rel)[:6]
    # two ill-conditioned updates + momentum at batch 2; the SE excitation biases are the most sensitive tensors.  Bands = 1.25 x
    # the worst deviation the CPU oracle itself shows from the reference under eight fp32 evaluation orders (threads, oneDNN /
    # native, direct / Winograd form — tests/golden/make_grad_noise.py part (d): 0.015 for `layer1.0.conv3x1_2.bias`, 0.067 for an
    # SE bias), never below the round-1 bands
    noise = np.load(os.path.join(golden_dir, 'grad_noise.npz'))['two_steps']
    tol_other, tol_se = max(1e-2, 1.25 * float(noise[0])), max(6e-2, 1.25 * float(noise[1]))
    tol = np.array([tol_se if 'se_layer' in nm else tol_other for nm in names])
    bad = np.nonzero(rel >= tol)[0]
    assert bad.size == 0, [(names[i], norms[i], g['param_norms'][i], rel[i]) for i in list(bad) + list(worst)]


@pytest.mark.gpu
def test_evaluate_protocol_config0(golden_dir):
    """eval.py protocol on BASELINE configs[0] (8 synthetic NYUv2-like pairs): mIoU*100 vs the reference."""
    from dynmm_amd import engine
    from dynmm_amd.nn.net import SkipGateESANet
    g = np.load(os.path.join(golden_dir, 'nyu8_P_se.npz'))
    m = SkipGateESANet(encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().eval()
    m.baseline = True
    rgb, depth = synth.synth_inputs(8, 480, 640, seed=77, nyu_like=True, device='cuda')
    label = synth.synth_labels(8, 480, 640, seed=78, device='cuda')
    batches = [(rgb[i:i + 4], depth[i:i + 4], label[i:i + 4]) for i in (0, 4)]
    miou, cm = engine.evaluate(m, batches)
    assert abs(miou - 100 * float(g['miou'])) < 1e-2
    assert np.abs(cm.numpy() - g['cm']).sum() <= 64


@pytest.mark.gpu
def test_train_driver_smoke(tmp_path):
    from dynmm_amd import train
    logs = train.train_main(['--dynamic', '--global-gate', '--encoder', 'resnet34', '--encoder_block', 'NonBottleneck1D',
                             '--decoder_channels_mode', 'constant', '--no_imagenet_pretraining', '--dataset', 'synthetic',
                             '--height', '96', '--width', '128', '--batch_size', '4', '--synthetic_samples', '8',
                             '--epochs', '2', '--epoch-hard', '1', '--loss-ratio', '0.1', '--eval-every', '1',
                             '--results_dir', str(tmp_path)])
    assert len(logs) == 2 and all(np.isfinite(r['loss_train_total']) for r in logs)
    assert 'mIoU_test' in logs[0]


@pytest.mark.gpu
def test_train_driver_smoke_per_stage_gates(tmp_path):
    """--dynamic without --global-gate: SkipESANet (per-stage Gumbel gates) through the same driver."""
    from dynmm_amd import train
    logs = train.train_main(['--dynamic', '--block-rule', '2222', '--encoder', 'resnet34', '--encoder_block',
                             'NonBottleneck1D', '--decoder_channels_mode', 'constant', '--no_imagenet_pretraining',
                             '--dataset', 'synthetic', '--height', '96', '--width', '128', '--batch_size', '4',
                             '--synthetic_samples', '8', '--epochs', '2', '--epoch-hard', '1', '--eval-every', '1',
                             '--results_dir', str(tmp_path)])
    assert len(logs) == 2 and all(np.isfinite(r['loss_train_total']) for r in logs)
    assert 'mIoU_test' in logs[0]


# ---------------------------------------------------------------------------------------------------
# fused flat optimizers, --freeze, hipGraph keying (GPU)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['SGD', 'Adam'])
def test_fused_optimizers_match_torch_optim(kind):
    """dynmm_sgd_nesterov / dynmm_adam vs torch.optim.SGD(nesterov) / torch.optim.Adam on the CPU over 5 steps
    with OneCycle-style lr AND momentum/beta1 changes, odd-sized parameters (unaligned range edges) and a
    parameter group that only starts receiving gradients at step 2 (the gate after ini_stage)."""
    from dynmm_amd import engine
    torch.manual_seed(3)
    sizes = [(7,), (33, 5), (4, 3, 3, 3), (1,), (130,)]
    ref_p = [torch.nn.Parameter(torch.randn(s)) for s in sizes]
    hip_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
    fp = engine.FlatParameters(hip_p)
    flat_g = torch.zeros_like(fp.flat)
    for q in hip_p:
        lo, hi = fp.span[id(q)]
        q.grad = flat_g[lo:hi].view_as(q)
    groups = {'gate': [hip_p[4]], 'rest': hip_p[:4]}
    if kind == 'SGD':
        ref = torch.optim.SGD(ref_p, lr=0.1, momentum=0.9, weight_decay=1e-2, nesterov=True)
        opt = engine.SGDNesterov(fp, flat_g, 0.1, 0.9, 1e-2, groups)
    else:
        ref = torch.optim.Adam(ref_p, lr=0.1, betas=(0.9, 0.999), weight_decay=1e-2)
        opt = engine.Adam(fp, flat_g, 0.1, weight_decay=1e-2, groups=groups)
    loss = torch.ones(1, device='cuda')
    for step in range(5):
        lr, mom = 0.1 / (1 + step), 0.95 - 0.02 * step
        for gr in ref.param_groups:
            gr['lr'] = lr
            if kind == 'SGD':
                gr['momentum'] = mom
            else:
                gr['betas'] = (mom, 0.999)
        opt.set_lr(lr)
        opt.set_momentum(mom)
        touched = set()
        for i, (rp, hp) in enumerate(zip(ref_p, hip_p)):
            # no gradient: torch.optim skips the parameter.  (A parameter skipped while its group-mates are not would
            # also desynchronise torch.Adam's per-parameter step counters from the per-group counter; the training
            # protocol never does that — the gate group is all-or-nothing — so it is exercised for SGD only.)
            if i == 4 and step < 2 or (kind == 'SGD' and i == 1 and step == 3):
                rp.grad = None
                continue
            g = torch.randn(sizes[i])
            rp.grad = g.clone()
            hp.grad.copy_(g)
            touched.add(id(hp))
        ref.step()
        opt.step(touched, loss)
        for rp, hp in zip(ref_p, hip_p):
            assert torch.allclose(hp.detach().cpu(), rp.detach(), rtol=1e-5, atol=1e-6), \
                (kind, step, (hp.detach().cpu() - rp.detach()).abs().max().item())
    opt.check_finite()
    # NaN guard: a non-finite loss skips the update and latches the step
    before = fp.flat.clone()
    opt.step(None, torch.full((1,), float('nan'), device='cuda'))
    assert torch.equal(fp.flat, before)
    with pytest.raises(ValueError, match='Loss is None'):
        opt.check_finite()
    opt.check_finite()           # latch cleared


def _small_model(seed=0, h=96, w=128):
    from dynmm_amd.nn.net import SkipGateESANet
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), seed)
    return m.cuda().train()


def _batch(n=3, h=96, w=128):
    rgb, depth = synth.synth_inputs(n, h, w, seed=1234, device='cuda')
    labels = [synth.synth_labels(n, h // s, w // s, seed=300 + s, device='cuda') for s in (1, 8, 16, 32)]
    return rgb, depth, labels


@pytest.mark.gpu
def test_freeze_trains_only_the_gate():
    """--freeze (train.py:139-141 + …globalgate.py:225-228): everything but the gate keeps its values bit for
    bit (no weight decay, no momentum), the gate parameters move by lr * (their oracle gradient)."""
    from dynmm_amd import engine
    from oracle import dynmm_oracle as O
    m = _small_model()
    m.temp, m.hard_gate = 1.0, False
    m.freeze()
    sd0 = {k: v.detach().clone().cpu() for k, v in m.state_dict().items()}
    cw = np.linspace(0.5, 2.0, 40)
    rgb, depth, labels = _batch()
    step = engine.TrainStep(m, cw, lr=0.05, momentum=0.0, weight_decay=0.0, loss_ratio=0.5, flop_budget=0.0)
    assert step.flatp.flat.numel() == sum(p.numel() for n, p in m.named_parameters() if 'gate' in n)
    step(rgb, depth, labels)
    torch.cuda.synchronize()
    # oracle gradient of the same loss w.r.t. the gate parameters
    sdo = {k: v.clone() for k, v in sd0.items()}
    params = {k: v.requires_grad_(True) for k, v in sdo.items() if 'gate' in k and v.dtype.is_floating_point
              and 'running_' not in k}
    outs, lf = O.forward(sdo, rgb.cpu(), depth.cpu(), Hh.CFGS['P_se'], training=True, temp=1.0)
    losses = O.cross_entropy_2d(outs, [l.cpu() for l in labels], torch.as_tensor(cw, dtype=torch.float32))
    (sum(losses) + 0.5 * torch.clamp(lf, min=0.0)).backward()
    new = m.state_dict()
    for k, v in sd0.items():
        if 'gate' in k and k in params:
            # the applied update (old - new) / lr is the HIP gate gradient; the gate sits behind the whole net, so its
            # fp32 gradient carries the conditioning noise of DESIGN.md §1 (a few 1e-2 at this tiny batch)
            upd = ((v - new[k].cpu()) / 0.05).double().flatten()
            ref_g = params[k].grad.double().flatten()
            if ref_g.norm() > 1e-6:
                assert not torch.equal(new[k].cpu(), v), k
                cos = torch.nn.functional.cosine_similarity(upd, ref_g, dim=0).item()
                print(f'gate gradient {k}: cosine {cos:.5f}, norm ratio {(upd.norm() / ref_g.norm()).item():.4f}')
                assert cos > 0.995 and abs((upd.norm() / ref_g.norm()).item() - 1) < 0.05, (k, cos, upd.norm(), ref_g.norm())
        elif 'gate' not in k and v.dtype.is_floating_point and 'running_' not in k:
            assert torch.equal(new[k].cpu(), v), k                  # frozen: bit-identical


@pytest.mark.gpu
def test_hip_graph_follows_epoch_schedule():
    """temp / hard_gate / ini_stage are frozen into a captured graph.  TrainStep keys its
    captures on them (and runs ini_stage steps eagerly), so a graph run must track an eager run through the
    epoch protocol of train.py:193-197; the returned losses must be fresh tensors each step."""
    from dynmm_amd import engine
    cw = np.linspace(0.5, 2.0, 40)
    rgb, depth, labels = _batch(4)
    runs = {}
    for use_graph in (False, True):
        m = _small_model()
        step = engine.TrainStep(m, cw, lr=0.01, loss_ratio=0.1, use_graph=use_graph)
        outs = []
        for temp, hard, ini in ((1.0, False, True), (1.0, False, False), (0.5, False, False), (0.5, False, False),
                                (0.25, True, False)):
            m.temp, m.hard_gate, m.ini_stage = temp, hard, ini
            m.ini_branches = [1, 4, 0, 2]
            outs.append(step(rgb, depth, labels))
        torch.cuda.synchronize()
        runs[use_graph] = outs
        if use_graph:
            # one capture per (shapes, hard_gate, baseline, …) key: soft (taken at temp 1.0, RE-captured at 0.5 — the
            # temperature is not part of the key, ADVICE r2: one capture per epoch would exhaust memory) and hard;
            # ini ran eagerly
            assert len(step._graphs) == 2
            for t_ in (0.2, 0.1, 0.05, 0.02):                      # ExpDecayTemp over further "epochs": the cache stays put
                m.temp = t_
                step(rgb, depth, labels)
            assert len(step._graphs) == 2 and len(step._graphs) <= step.MAX_GRAPHS
            m.temp = 0.25
        step.opt.check_finite()
    assert len({o['total'].data_ptr() for o in runs[True]}) == len(runs[True])       # no aliasing of `last`
    for a, b in zip(runs[False], runs[True]):
        assert torch.allclose(a['losses'], b['losses'], rtol=2e-3), (a['losses'], b['losses'])
        assert torch.allclose(a['loss_flop'], b['loss_flop'], rtol=1e-3, atol=1e-5)
    # the temperature change must be visible: step 2 (temp 0.5) differs from what temp 1.0 would have given
    assert not torch.allclose(runs[True][1]['loss_flop'], runs[True][2]['loss_flop'], rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_train_driver_adam_and_freeze(tmp_path):
    from dynmm_amd import train
    common = ['--dynamic', '--global-gate', '--encoder', 'resnet34', '--encoder_block', 'NonBottleneck1D',
              '--decoder_channels_mode', 'constant', '--no_imagenet_pretraining', '--dataset', 'synthetic',
              '--height', '96', '--width', '128', '--batch_size', '4', '--synthetic_samples', '8', '--epochs', '2',
              '--eval-every', '1', '--results_dir', str(tmp_path)]
    logs = train.train_main(common + ['--optimizer', 'Adam', '--lr', '0.001'])
    assert len(logs) == 2 and all(np.isfinite(r['loss_train_total']) for r in logs)
    logs = train.train_main(common + ['--freeze', '--hip_graph', '--loss-ratio', '0.1'])
    assert len(logs) == 2 and all(np.isfinite(r['loss_train_total']) for r in logs)


@pytest.mark.gpu
def test_train_driver_cli_default_encoder(tmp_path):
    """The reference CLI's defaults (src/args.py:105,110-111,151): --encoder resnet50 (Bottleneck), decreasing decoder
    channels [512, 256, 128], SE-add — the README commands as written (README.md:78-98 pass no --encoder)."""
    from dynmm_amd import train
    logs = train.train_main(['--dynamic', '--global-gate', '--no_imagenet_pretraining', '--dataset', 'synthetic',
                             '--height', '96', '--width', '128', '--batch_size', '2', '--synthetic_samples', '4',
                             '--epochs', '1', '--eval-every', '1', '--results_dir', str(tmp_path)])
    assert len(logs) == 1 and np.isfinite(logs[0]['loss_train_total']) and 'mIoU_test' in logs[0]


@pytest.mark.gpu
def test_fused_tail_step_equals_unfused_step():
    """TrainStep with the last up-sampling fused into the loss (csrc/tail.hip) against the same step with the logits
    materialised: same losses, same gradients (summation order only)."""
    from dynmm_amd import engine
    from tests.test_hip_model import hip_model
    torch.manual_seed(0)
    h, w, n = 96, 128, 3
    rgb, depth = torch.randn(n, 3, h, w).cuda(), torch.randn(n, 1, h, w).cuda()
    labels = [torch.randint(0, 41, (n, h // s, w // s), dtype=torch.uint8).cuda() for s in (1, 8, 16, 32)]
    cw = np.linspace(0.5, 1.5, 40).astype(np.float32)
    out = {}
    for fused in (True, False):
        m = hip_model('P_se', h, w, seed=3)
        m.train()
        m.temp, m.hard_gate = 1.0, False
        step = engine.TrainStep(m, cw, lr=0.0, loss_ratio=1e-3, fuse_tail=fused)
        step._body(rgb, depth, labels)
        torch.cuda.synchronize()
        assert m.decoder.defer_tail is False
        out[fused] = (step.last['losses'].cpu(), {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    assert torch.allclose(out[True][0], out[False][0], rtol=2e-6, atol=2e-6)
    # per tensor, relative to the tensor's own scale — but not below 1e-2 of the largest gradient of the model:
    # conv biases in front of a batch norm have an analytically zero gradient (~1e-6 of rounding noise on both sides)
    gmax = max(gb.abs().max().item() for gb in out[False][1].values())
    worst = 0.0
    for k, ga in out[True][1].items():
        gb = out[False][1][k]
        worst = max(worst, ((ga - gb).abs().max() / max(gb.abs().max().item(), 1e-2 * gmax)).item())
    assert worst < 5e-4, worst
    va = torch.cat([g.flatten() for g in out[True][1].values()]).double()
    vb = torch.cat([g.flatten() for g in out[False][1].values()]).double()
    assert abs(torch.nn.functional.cosine_similarity(va, vb, dim=0).item() - 1) < 1e-9
    assert abs((va.norm() / vb.norm()).item() - 1) < 1e-5


@pytest.mark.gpu
def test_prepacked_weights_step_is_bit_identical(monkeypatch):
    """ops.PackedWeights: from the second step on every conv weight is packed by ONE multi-tensor launch at the start
    of the step; losses, gradients and updated parameters must be bit-identical to per-convolution packing, the arena
    must follow the optimizer's updates, and a step that meets an unregistered weight falls back for it."""
    from dynmm_amd import engine, ops
    from tests.test_hip_model import hip_model
    torch.manual_seed(1)
    h, w, n = 96, 128, 2
    rgb, depth = torch.randn(n, 3, h, w).cuda(), torch.randn(n, 1, h, w).cuda()
    labels = [torch.randint(0, 41, (n, h // s, w // s), dtype=torch.uint8).cuda() for s in (1, 8, 16, 32)]
    cw = np.linspace(0.5, 1.5, 40).astype(np.float32)
    res = {}
    for pre in (True, False):
        m = hip_model('P_se', h, w, seed=4)
        m.train()
        m.temp, m.hard_gate = 1.0, False
        step = engine.TrainStep(m, cw, lr=0.05, loss_ratio=1e-3, prepack=pre)
        assert (step.prepack is not None) == pre
        losses = [step(rgb, depth, labels)['total'].item() for _ in range(3)]
        torch.cuda.synchronize()
        if pre:
            assert len(step.prepack.reg) > 150 and step.prepack.arena is not None and not step.prepack.valid
            assert ops.PREPACK is None
        res[pre] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()},
                    {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
    assert res[True][0] == res[False][0]
    assert res[True][0][0] != res[True][0][2]          # the weights (and so the packed operands) did change
    for k, v in res[True][1].items():
        assert torch.equal(v, res[False][1][k]), k
    for k, v in res[True][2].items():
        assert torch.equal(v, res[False][2][k]), k


# ---------------------------------------------------------------------------------------------------
# validate(): validation losses + per-camera mIoU (train.py:368-551)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_validation_losses_match_reference_fixture(golden_dir):
    """dynmm_ce2d_valid (one pass: weighted sum, weight sum, unweighted sum, pixel count) accumulated over two batches
    vs the reference's CrossEntropyLoss2dForValidData / ...Unweighted (tests/golden/valid_loss.npz)."""
    from dynmm_amd import ops
    g = np.load(os.path.join(golden_dir, 'valid_loss.npz'))
    cw = torch.from_numpy(g['weight']).cuda()
    acc = torch.zeros(4, dtype=torch.float64, device='cuda')
    for i in range(2):
        ops.validation_loss_accumulate(torch.from_numpy(g[f'x{i}']).cuda(), torch.from_numpy(g[f't{i}']).cuda(), cw, acc)
    sw, ws, su, npx = acc.tolist()
    assert abs(ws - float(g['weighted_pixel_sum'])) < 1e-6 * ws       # = sum_c pixels_c * w_c of the labels (train.py:104-108)
    assert abs(sw / float(g['weighted_pixel_sum']) - float(g['loss_weighted'])) < 2e-6 * float(g['loss_weighted'])
    assert abs(su / npx - float(g['loss_unweighted'])) < 2e-6 * float(g['loss_unweighted'])
    assert npx == sum(int((g[f't{i}'] > 0).sum()) for i in range(2))


@pytest.mark.gpu
def test_validate_mirrors_the_reference_protocol():
    """engine.validate = train.py:368-551: hard gates unless soft_eval, one confusion matrix per camera, mIoU per camera,
    weighted + unweighted validation loss; every number against the oracle evaluated on the same (HIP) logits, and the
    model's mode / gate flag restored afterwards."""
    from dynmm_amd import engine
    from dynmm_amd.data import SyntheticRGBD
    from dynmm_amd.nn.net import SkipGateESANet
    from oracle import dynmm_oracle as O
    h, w = 96, 128
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), 0)
    m = m.cuda().train()
    m.hard_gate = False
    cw = np.linspace(0.5, 2.0, 40)
    cams = {'kv1': SyntheticRGBD(6, 4, h, w, seed=5, device='cuda'), 'xtion': SyntheticRGBD(3, 3, h, w, seed=9, device='cuda')}
    miou, logs = engine.validate(m, cams, cw, split='test')
    assert m.training and m.hard_gate is False
    assert set(miou) == {'kv1', 'xtion'}
    for k in ('loss_test', 'loss_test_unweighted', 'mIoU_test_kv1', 'mIoU_test_xtion', 'time_validation'):
        assert k in logs, k
    m.eval()
    m.hard_gate = True
    xs, ts = [], []
    for cam, loader in cams.items():
        cm = torch.zeros(40, 40, dtype=torch.int64)
        for s in loader:
            with torch.no_grad():
                logits = m(s['image'], s['depth'], True).cpu()
            xs.append(logits)
            ts.append(s['label'].cpu())
            lab, pred = O.eval_postprocess(logits, s['label_orig'].cpu().long())
            cm += O.confusion_matrix(lab, pred, 40)
        assert torch.equal(cm, logs['confusion_matrices'][cam])
        assert abs(100 * float(O.iou_from_cm(cm)[1]) - miou[cam]) < 1e-9
    lw, lu = O.validation_losses(xs, ts, cw)
    assert abs(lw - logs['loss_test']) < 1e-5 * lw and abs(lu - logs['loss_test_unweighted']) < 1e-5 * lu


def test_legacy_flat_optimizer_checkpoints_load_into_the_right_buffers():
    """the rounds-1-2 checkpoint layout ({'exp_avg': flat, 'exp_avg_sq': flat, 'steps': ...}) must restore Adam's
    two moment buffers (it used to copy exp_avg into v and exp_avg_sq into the int32 step counters)."""
    from dynmm_amd import engine
    ps = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(3, 2))]
    fp = engine.FlatParameters(ps)
    grads = torch.zeros_like(fp.flat)
    adam = engine.Adam(fp, grads, lr=1e-3)
    sd = {'exp_avg': torch.arange(11.0), 'exp_avg_sq': torch.arange(11.0) * 2, 'steps': torch.tensor([7], dtype=torch.int32)}
    adam.load_state_dict(sd)
    assert torch.equal(adam.m, sd['exp_avg']) and torch.equal(adam.v, sd['exp_avg_sq']) and int(adam.steps[0]) == 7
    sgd = engine.SGDNesterov(fp, grads, lr=1e-3)
    sgd.load_state_dict({'momentum_buffer': torch.arange(11.0) * 3, 'steps': torch.tensor([2], dtype=torch.int32)})
    assert torch.equal(sgd.buf, torch.arange(11.0) * 3) and int(sgd.steps[0]) == 2


@pytest.mark.gpu
def test_stream_plan_is_process_wide_three_models_in_sequence():
    """the step's side streams are ONE plan per process and device (ops.stream_plan) — the second and third model
    built in a process run on the same four streams as the first and take the same time (round 5: a per-instance torch pool
    stream for the depth encoder made every later model 17 % slower at BASELINE configs[2]'s size), and a caller that brings a
    fifth stream is refused."""
    import time
    from dynmm_amd import engine, ops
    from dynmm_amd.nn.net import SkipGateESANet
    dev = torch.device('cuda:0')
    n, h, w = 32, 480, 640
    g = torch.Generator().manual_seed(7)
    rgb, depth = torch.randn(n, 3, h, w, generator=g).to(dev), torch.randn(n, 1, h, w, generator=g).to(dev)
    labels = [torch.randint(0, 41, (n, h // s, w // s), generator=g, dtype=torch.uint8).to(dev) for s in (1, 8, 16, 32)]
    plan = ops.stream_plan()
    expect = {torch.cuda.current_stream().cuda_stream, plan.side.cuda_stream} | {s.cuda_stream for s in plan.wgrad[:ops.WGRAD_STREAMS]}
    times = []
    for i in range(3):
        m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
        synth.fill_state_dict(m.state_dict(), seed=i)
        m = m.to(dev).train()
        m.temp, m.hard_gate = 1.0, bool(i == 1)             # the second one runs configs[3]'s dense hard-gate step
        ts = engine.TrainStep(m, np.linspace(0.5, 2.0, 40), lr=1e-4, loss_ratio=1.0)
        for _ in range(3):
            ts(rgb, depth, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            ts(rgb, depth, labels)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / 6 * 1e3)
        assert ts.census['streams'] == ops.MAX_BUSY_STREAMS == 4, ts.census
        assert ops.busy_streams() == expect, (ops.busy_streams(), expect)
        assert ops.stream_plan() is plan
        if i == 2:                                          # a stream of the caller's own beside the plan: refused, loudly
            extra = torch.cuda.Stream()

            def fifth(module, args):
                extra.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(extra):
                    ops.adaptive_avg_pool(args[1], 1)       # any launch of the library on a stream outside the plan
            hook = m.register_forward_pre_hook(fifth)
            try:
                with pytest.raises(Exception, match='enqueued work on 5 streams'):
                    ts._body(rgb, depth, labels)
            finally:
                hook.remove()
                ops.flush_wgrad_groups()
                torch.cuda.synchronize()
        ts.reducer.remove_hooks()
        del m, ts
        torch.cuda.empty_cache()
    assert max(times) / min(times) < 1.03, times


@pytest.mark.gpu
def test_infer_step_graph_replay_equals_eager():
    """engine.InferStep: the inference forward replayed as hipGraphs is bit-identical to the eager forward —
    baseline (configs[1]: static fuse), soft gates, hard gates with the branches injected, and hard DATA-DEPENDENT gates with
    compaction (front graph -> 16-byte host read -> back graph per stage-count tuple, captured on its second sighting) — follows
    new inputs, returns the gate weights on request, and drops its captures when the weights move."""
    from dynmm_amd import engine
    from dynmm_amd.nn.net import SkipGateESANet
    from dynmm_amd.nn.esanet import ESANet
    dev = torch.device('cuda:0')
    h, w, n = 96, 128, 6
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.to(dev).eval()
    m.dual_stream = True
    batches = [synth.synth_inputs(n, h, w, seed=900 + i, device=dev) for i in range(3)]
    step = engine.InferStep(m, capture_after=2)

    def eager(rgb, depth):
        with torch.no_grad():
            out, wgt = m(rgb, depth, True, True)
        return out.clone(), wgt.clone()

    def check(tag, expect_launch=None):
        for i, (rgb, depth) in enumerate(batches + batches[:1]):
            ref, ref_w = eager(rgb, depth)
            out, wgt = step(rgb, depth, return_weight=True)
            assert torch.equal(out, ref) and torch.equal(wgt, ref_w), (tag, i, step.launch)
            out2 = step(rgb, depth)
            assert torch.equal(out2, ref), (tag, i, 'no weight')
        if expect_launch is not None:
            assert step.launch == expect_launch, (tag, step.launch, step.replays)

    m.baseline, m.hard_gate, m.temp = True, False, 1.0
    check('baseline', 'hipGraph replay')
    m.compact = False
    check('baseline dense', 'hipGraph replay')
    m.compact = True
    m.baseline = False
    check('soft', 'hipGraph replay')
    m.hard_gate = True
    m.branch_override = [4, 0, 2, 3, 1, 4]
    check('hard injected', 'hipGraph replay')
    m.branch_override = None
    m.temp = 0.1
    before = dict(step.replays)
    check('hard data-dependent')
    assert step.replays['back'] > before['back']                  # count tuples seen twice were replayed as graphs
    assert step.launch in ('hipGraph replay', 'hipGraph front + eager back')
    captures = step.replays['captures']
    check('hard data-dependent again')
    # weights move (an optimizer step / load_state_dict): every capture is dropped and retaken from the new weights
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    synth.fill_state_dict(sd, seed=5)
    m.load_state_dict(sd)
    check('after load_state_dict')
    assert step.replays['captures'] > captures
    # policy='auto': the faster of replay and eager launches is measured once per key and kept; results stay identical
    m.baseline, m.hard_gate = True, False
    auto = engine.InferStep(m, policy='auto')
    for rgb, depth in batches + batches:
        ref, _ = eager(rgb, depth)
        assert torch.equal(auto(rgb, depth), ref)
    assert list(auto._choice.values()) and set(auto._choice.values()) <= {'replay', 'eager'} and auto.auto_timing['replay_ms'] > 0
    assert auto.launch == ('eager' if list(auto._choice.values())[0] == 'eager' else 'hipGraph replay')
    m.baseline, m.hard_gate = False, True
    m.ini_stage = True                                             # host RNG per call: always eager
    with torch.no_grad():
        step(*batches[0])
    assert step.launch == 'eager'
    m.ini_stage = False
    # the static ESANet (no gate): same machinery with baseline forced
    e = ESANet(height=h, width=w, num_classes=40, encoder_rgb='resnet34', encoder_depth='resnet34', encoder_block='NonBottleneck1D',
               pretrained_on_imagenet=False, fuse_depth_in_rgb_encoder='SE-add', upsampling='learned-3x3-zeropad')
    synth.fill_state_dict(e.state_dict(), seed=1)
    e = e.to(dev).eval()
    es = engine.InferStep(e)
    for rgb, depth in batches:
        with torch.no_grad():
            ref = e(rgb, depth).clone()
        assert torch.equal(es(rgb, depth), ref) and es.launch == 'hipGraph replay'


@pytest.mark.gpu
def test_eval_run_takes_the_callers_loader():
    """dynmm_amd.eval.run(args, model, loader) (VERDICT r5 missing #3): the evaluation protocol of eval.py:63-151 on a loader the
    CALLER brings (host tensors, dict samples) — what a user with NYUv2 on disk calls; graph replay and eager launches agree."""
    import argparse
    from dynmm_amd import eval as ev
    from dynmm_amd.nn.net import SkipGateESANet
    dev = torch.device('cuda:0')
    h, w, n = 96, 128, 4
    m = SkipGateESANet(height=h, width=w, encoder_block='NonBottleneck1D', fuse_depth_in_rgb_encoder='SE-add')
    synth.fill_state_dict(m.state_dict(), seed=0)
    m = m.to(dev)
    loader = []
    for i in range(3):
        rgb, depth = synth.synth_inputs(n, h, w, seed=40 + i)                       # HOST tensors: run() moves them
        loader.append({'image': rgb, 'depth': depth, 'label_orig': synth.synth_labels(n, 2 * h, 2 * w, seed=50 + i).to(torch.uint8)})
    args = argparse.Namespace(hard=True, ini=False, baseline=False, num_runs=2, mode=2, noise=0.1)
    a = ev.run(args, m, loader, use_graph=True)
    b = ev.run(args, m, loader, use_graph=False)
    assert len(a) == 2 and a == b and all(0.0 <= v <= 100.0 for v in a)
    args.baseline, args.mode = True, -1
    assert ev.run(args, m, loader, use_graph=True) == ev.run(args, m, loader, use_graph=False)
