"""Modality-level DynMM (ModalityDynMM/affect/affect_dyn.py, BASELINE configs[4]) — PARITY UNPINNED: the experts are
MultiBench modules that /root/reference neither contains nor pins, so the checker is oracle/affect_oracle.py, a
restatement built from the torch.nn layers MultiBench wraps (see its header).  CPU: the two sides agree on their
state_dict layout; GPU: kernels and whole models, forward and backward, against that oracle."""
import numpy as np
import pytest
import torch


def test_state_dict_layout_matches_oracle():
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    for mine, ref in ((A.DynMMNetV2(), O.DynMMNetV2()), (A.DynMMNet(freeze=False), O.DynMMNet())):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(tuple(a[k].shape) == tuple(b[k].shape) for k in a)
    v2 = A.DynMMNetV2(freeze=True)
    assert all(p.requires_grad == n.startswith('gate') for n, p in v2.named_parameters())
    with pytest.raises(NotImplementedError):
        A.DynMMNetV2(model_name_list=['b1.pt', 'b2.pt'])


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.gpu
@pytest.mark.parametrize('B,D,T', [(3, 60, 50), (2, 120, 50), (4, 10, 7)])
def test_layernorm_and_attention_kernels(B, D, T):
    from dynmm_amd import ops_seq as S
    torch.manual_seed(0)
    x = torch.randn(B, D, T, requires_grad=True)
    r = torch.randn(B, D, T, requires_grad=True)
    gamma = (torch.rand(D) + 0.5).requires_grad_(True)
    beta = torch.randn(D).requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm((x + r).permute(0, 2, 1), (D,), gamma, beta, 1e-5).permute(0, 2, 1)
    g = torch.randn(B, D, T)
    y_ref.backward(g)
    xc, rc, gc, bc = (t.detach().cuda().requires_grad_(True) for t in (x, r, gamma, beta))
    y = S.layernorm_bdt(xc, gc, bc, 1e-5, residual=rc)
    y.backward(g.cuda())
    assert _rel(y, y_ref) < 1e-5
    for a, b in ((xc, x), (rc, r), (gc, gamma), (bc, beta)):
        assert _rel(a.grad, b.grad) < 2e-5
    # attention core vs torch's scaled_dot_product_attention on the same q | k | v split
    heads = 5
    qkv = torch.randn(B, 3 * D, T, requires_grad=True)
    q, k, v = (t.reshape(B, heads, D // heads, T).permute(0, 1, 3, 2) for t in qkv.split(D, dim=1))
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).permute(0, 1, 3, 2).reshape(B, D, T)
    ref.backward(g)
    qc = qkv.detach().cuda().requires_grad_(True)
    out = S.mha_core(qc, heads)
    out.backward(g.cuda())
    assert _rel(out, ref) < 1e-5 and _rel(qc.grad, qkv.grad) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('kind,hard', [('v2', False), ('v2', True), ('v1', False)])
def test_dynmm_affect_model_matches_oracle(kind, hard):
    """Whole model, eval-mode arithmetic (dropout = 0), forward and every trainable gradient, HIP vs oracle."""
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    ref = O.fill_(O.DynMMNetV2(0.7, hard) if kind == 'v2' else O.DynMMNet(0.7, hard), seed=1)
    mine = (A.DynMMNetV2(0.7, hard) if kind == 'v2' else A.DynMMNet(0.7, hard, freeze=False))
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda()
    inputs, y = O.synth_batch(6, seed=3)
    out_r, aux_r, w_r = ref(inputs)
    tot_r, _ = O.train_objective(out_r, aux_r, y, 0.3)
    tot_r.backward()
    inputs_c = [[x.cuda() for x in inputs[0]], inputs[1]]
    out, aux = mine(inputs_c)
    tot = (out - y.cuda()).abs().mean() + 0.3 * aux
    tot.backward()
    torch.cuda.synchronize()
    assert _rel(out, out_r) < 2e-4 and abs(aux.item() - aux_r.item()) < 1e-5 and abs(tot.item() - tot_r.item()) < 1e-5
    gr = dict(ref.named_parameters())
    errs = {}
    for n, p in mine.named_parameters():
        if gr[n].grad is None or gr[n].grad.abs().max() < 1e-9:
            continue
        errs[n] = ((p.grad.cpu().double() - gr[n].grad.double()).norm() / gr[n].grad.double().norm()).item()
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-3, (worst, errs[worst])
    assert np.median(list(errs.values())) < 2e-4


@pytest.mark.gpu
def test_affect_train_step_matches_torch_adamw():
    """AffectTrainStep (fused loss + seeds, clip_grad_norm_ 8, AdamW on flat buffers) vs the oracle model driven by
    torch.optim.AdamW + torch.nn.utils.clip_grad_norm_, two steps, frozen experts (the `--freeze` usage) and not."""
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    for freeze, use_graph in ((True, False), (False, False), (True, True)):
        ref = O.fill_(O.DynMMNetV2(1.0, False), seed=2)
        mine = A.DynMMNetV2(1.0, False, freeze=freeze)
        mine.load_state_dict(ref.state_dict())
        mine = mine.cuda()
        if freeze:
            for n, p in ref.named_parameters():
                p.requires_grad = n.startswith('gate')
        params = [p for p in ref.parameters() if p.requires_grad]
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-2)
        step = A.AffectTrainStep(mine, lr=1e-3, weight_decay=1e-2, lossw=0.2, clip_val=0.05 if freeze else 8.0,
                                 use_graph=use_graph)
        clip = 0.05 if freeze else 8.0
        for it in range(2):
            inputs, y = O.synth_batch(5, seed=10 + it)
            opt.zero_grad()
            out_r, aux_r, _ = ref(inputs)
            tot_r, l1_r = O.train_objective(out_r, aux_r, y, 0.2)
            tot_r.backward()
            norm_r = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
            opt.step()
            res = step([[x.cuda() for x in inputs[0]], inputs[1]], y.cuda())
            tol = 2e-5 if it == 0 else 2e-4      # step 2 sits behind one Adam update (sign-like: amplifies rounding)
            assert abs(res['total'].item() - tot_r.item()) < tol and abs(res['loss1'].item() - l1_r.item()) < tol
            assert abs(res['grad_norm'].item() - norm_r.item()) < 2e-3 * max(norm_r.item(), 1e-3)
        sd_r, sd = ref.state_dict(), mine.state_dict()
        for k in sd_r:
            # Adam's first updates are lr * sign(g) whatever |g|: an element whose gradient is rounding noise can move the
            # other way (2 update sizes apart).  Almost every element must agree to a fraction of an update, none may be
            # further apart than the two updates allow.
            d = (sd[k].cpu() - sd_r[k]).abs()
            if k.endswith('in_proj_bias'):
                # the key bias of softmax attention has an analytically ZERO gradient (shift invariance): what Adam sees
                # there is pure rounding noise on both sides
                third = d.numel() // 3
                d = torch.cat([d[:third], d[2 * third:]])
            # (small tensors: a single flipped element is allowed whatever the tensor's size)
            n_far = int((d > 0.2 * 2 * 1e-3).sum().item())
            assert n_far <= max(1, int(2e-3 * d.numel())) and d.max().item() < 2.2 * 2 * 1e-3, (freeze, k, n_far)
        step.opt.check_finite()


@pytest.mark.gpu
def test_infer_modes_record_the_gate_weights():
    """affect_dyn.py:152-165: the gate's DiffSoftmax weight is stored BEFORE infer_mode is looked at — cal_flop / weight_stat
    read weight_list after single-branch (infer_mode > 0) and uniform (infer_mode = -1) evaluations too."""
    from dynmm_amd.nn import affect as A
    torch.manual_seed(1)
    m = A.DynMMNetV2(1.0, True).cuda().eval()
    B, T = 6, 50
    xs = [torch.randn(B, T, f).cuda() for f in (35, 74, 300)]
    inputs = [xs, [torch.full((B,), T, dtype=torch.long)] * 3]
    with torch.no_grad():
        m.infer_mode = 0
        m.reset_weight()
        out0, _ = m(inputs)
        w0 = m.weight_list.clone()
        preds = m.experts(inputs)
        for mode in (1, 2, -1):
            m.infer_mode = mode
            m.reset_weight()
            out, _ = m(inputs)
            assert torch.equal(m.weight_list, w0), mode                   # the gate's real (hard) decisions, every mode
            want = preds[mode - 1] if mode > 0 else 0.5 * (preds[0] + preds[1])
            assert _rel(out, want) < 1e-5, mode
    assert w0.shape == (B, 2) and bool(((w0 == 0) | (w0 == 1)).all())
    assert np.isfinite(m.cal_flop())
