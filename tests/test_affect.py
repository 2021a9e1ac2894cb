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


class _Masks:
    """The n-th dropout site of a forward pass keeps element e iff rand_n(e) >= p: one deterministic stream both sides
    walk in the same order (gate, experts; per layer: attn, dropout1, dropout, dropout2)."""

    def __init__(self, p, seed, device='cpu'):
        self.p, self.seed, self.n, self.device = p, seed, 0, device
        self.names = []

    def __call__(self, name, shape):
        g = torch.Generator().manual_seed(self.seed * 100003 + self.n)
        self.n += 1
        self.names.append(name)
        m = (torch.rand(shape, generator=g) >= self.p).to(torch.uint8)
        return m.to(self.device)


def test_oracle_dropout_layer_is_torchs_layer():
    """oracle.encoder_layer_dropout against torch.nn.TransformerEncoderLayer itself: identical without dropout, and — with
    nn.Dropout.forward replaced by the same injected keep flags — identical at the three nn.Dropout sites (dropout1,
    dropout, dropout2).  (The fourth site sits inside scaled_dot_product_attention and cannot be injected into torch; its
    position — on the softmax output, before the product with V — is torch/nn/functional.py's.)"""
    from oracle import affect_oracle as O
    torch.manual_seed(0)
    T, B, D, p = 7, 3, 20, 0.25
    layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=5, dim_feedforward=48, dropout=p)
    x = torch.randn(T, B, D)
    ones = lambda name, shape: torch.full(shape, 1.0)
    layer.eval()
    assert _rel(O.encoder_layer_dropout(layer, x, 0.0, ones), layer(x)) < 1e-6
    layer.train()
    layer.self_attn.dropout = 0.0
    mk = _Masks(p, 5)
    served = []

    def patched(self, inp):                                    # inp [T, B, C]; the masks are kept in the HIP layout [B, C, T]
        m = mk('nn.Dropout', (inp.shape[1], inp.shape[2], inp.shape[0]))
        served.append(m)
        return inp * m.permute(2, 0, 1).to(inp.dtype) / (1.0 - p)
    orig = torch.nn.Dropout.forward
    torch.nn.Dropout.forward = patched
    try:
        want = layer(x)
    finally:
        torch.nn.Dropout.forward = orig
    it = iter(served)
    got = O.encoder_layer_dropout(layer, x, p, lambda name, shape: torch.full(shape, 1.0 - p) if name == 'attn' else next(it))
    assert len(served) == 3 and _rel(got, want) < 1e-6


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
def test_dropout_kernels_with_injected_masks():
    """LayerNorm(dropout(x) + r), attention with dropout on the probabilities and the plain dropout op, forward and
    backward, against the same arithmetic in torch with the same keep flags."""
    from dynmm_amd import ops_seq as S
    torch.manual_seed(0)
    B, D, T, H, p = 3, 60, 50, 5, 0.2
    x = torch.randn(B, D, T, requires_grad=True)
    r = torch.randn(B, D, T, requires_grad=True)
    gamma = (torch.rand(D) + 0.5).requires_grad_(True)
    beta = torch.randn(D).requires_grad_(True)
    g = torch.randn(B, D, T)
    mk = _Masks(p, 11, 'cuda')
    S.MASKS = mk
    try:
        xc, rc, gc, bc = (t.detach().cuda().requires_grad_(True) for t in (x, r, gamma, beta))
        y = S.layernorm_bdt(xc, gc, bc, 1e-5, residual=rc, drop=(p, 7, 'dropout1'))
        y.backward(g.cuda())
        m = _Masks(p, 11)('dropout1', (B, D, T)).float() / (1 - p)
        y_ref = torch.nn.functional.layer_norm((x * m + r).permute(0, 2, 1), (D,), gamma, beta, 1e-5).permute(0, 2, 1)
        y_ref.backward(g)
        assert _rel(y, y_ref) < 1e-5
        for a, b in ((xc, x), (rc, r), (gc, gamma), (bc, beta)):
            assert _rel(a.grad, b.grad) < 2e-5
        assert float((xc.grad == 0).float().mean()) > 0.5 * p           # dropped elements receive no gradient
        # attention
        qkv = torch.randn(B, 3 * D, T, requires_grad=True)
        qc = qkv.detach().cuda().requires_grad_(True)
        mk.n = 1
        out = S.mha_core(qc, H, drop=(p, 8, 'attn'))
        out.backward(g.cuda())
        mg = _Masks(p, 11); mg.n = 1
        ma = mg('attn', (B * H, T, T)).float() / (1 - p)
        q, k, v = (t.reshape(B * H, D // H, T).transpose(1, 2) for t in qkv.split(D, dim=1))       # [B*H, T, dh]
        att = torch.softmax(q @ k.transpose(1, 2) / (D // H) ** 0.5, dim=-1) * ma
        ref = (att @ v).transpose(1, 2).reshape(B, D, T)
        ref.backward(g)
        assert _rel(out, ref) < 1e-5 and _rel(qc.grad, qkv.grad) < 2e-5
        # plain op
        x2 = x.detach().cuda().requires_grad_(True)
        mk.n = 2
        y2 = S.dropout_bdt(x2, p, 9, 'dropout')
        y2.backward(g.cuda())
        mg.n = 2
        m2 = mg('dropout', (B, D, T)).float() / (1 - p)
        assert _rel(y2, x.detach() * m2) < 1e-6 and _rel(x2.grad, g * m2) < 1e-6
    finally:
        S.MASKS = None


@pytest.mark.gpu
def test_dropout_generator_statistics_and_replay():
    """The Philox path: keep rate, forward/backward consistency (the backward regenerates the forward's decisions), new
    decisions after advance_dropout_step, identical decisions for identical (seed, site, step), different sites differ."""
    from dynmm_amd import ops, ops_seq as S
    ops.manual_seed(1234)
    p = 0.1
    x = torch.randn(8, 120, 50, device='cuda').abs() + 0.1

    def keep(site):
        xi = x.clone().requires_grad_(True)
        y = S.dropout_bdt(xi, p, site, 'dropout')
        y.backward(torch.ones_like(y))
        k = (y.detach() != 0)
        assert torch.equal(xi.grad != 0, k)                                     # same decisions in the backward
        assert _rel(y.detach()[k], (x / (1 - p))[k]) < 1e-6
        return k
    k0 = keep(3)
    assert abs(k0.float().mean().item() - (1 - p)) < 0.005                      # 48 000 draws: sigma = 0.0014
    assert torch.equal(keep(3), k0)
    k1 = keep(4)
    assert 0.7 < (k1 == k0).float().mean().item() < 0.9                         # independent: agree on 0.82 of the elements
    S.advance_dropout_step(x.device)
    k2 = keep(3)
    assert 0.7 < (k2 == k0).float().mean().item() < 0.9
    # along each axis the decisions are not constant (a wrong index would repeat rows or columns)
    assert k0.float().mean((0, 1)).std().item() > 0 and k0.float().mean((1, 2)).std().item() > 0
    # attention and LayerNorm sites draw from the same generator
    qkv = torch.randn(4, 180, 50, device='cuda')
    o_eval = S.mha_core(qkv, 5)
    o_drop = S.mha_core(qkv, 5, drop=(p, 5, 'attn'))
    assert 0.01 < _rel(o_drop, o_eval) < 2.0
    g = torch.ones(120, device='cuda')
    y_a = S.layernorm_bdt(x, g, g, 1e-5, residual=x, drop=(p, 6, 'dropout1'))
    y_b = S.layernorm_bdt(x, g, g, 1e-5, residual=x, drop=(p, 6, 'dropout1'))
    assert torch.equal(y_a, y_b) and not torch.equal(y_a, S.layernorm_bdt(x, g, g, 1e-5, residual=x))


@pytest.mark.gpu
def test_shaped_dropout_generators_of_layernorm_and_attention():
    """The Philox path of the LayerNorm and attention sites (one call per EIGHT elements a lane owns: eight channels of a token,
    eight keys of a query row — csrc/seq.hip DropState::keep8 / row8): the keep pattern read back through the kernels' own
    outputs has the right rate, takes two values, does not repeat along any axis, its eight elements are independent, the
    backward regenerates exactly the forward's decisions, a new step draws new ones, and the LayerNorm backward with a
    workspace (parameter gradients out of the input-gradient pass) equals the two-pass entry."""
    from dynmm_amd import lib as L, ops, ops_seq as S
    ops.manual_seed(4321)
    p = 0.1
    B, D, T, H = 8, 120, 50, 5
    dev = 'cuda'
    one = torch.ones(D, device=dev)
    zero = torch.zeros(D, device=dev)

    def ln_keep(site, d=D, b=B):
        # x = 1, res = 0, gamma = 1, beta = 0: the gradient of x under a random g is dres * keep, and dres != 0 almost surely
        x = torch.ones(b, d, T, device=dev, requires_grad=True)
        r = torch.randn(b, d, T, device=dev, requires_grad=True)
        g = torch.randn(b, d, T, device=dev)
        y = S.layernorm_bdt(x, torch.ones(d, device=dev), torch.zeros(d, device=dev), 1e-5, residual=r, drop=(p, site, 'dropout1'))
        y.backward(g)
        k_bwd = x.grad != 0
        # forward: y = LN(keep/(1-p) + r); the same keep flags reproduce it in torch
        want = torch.nn.functional.layer_norm((k_bwd.float() / (1 - p) + r.detach()).permute(0, 2, 1), (d,)).permute(0, 2, 1)
        assert _rel(y, want) < 1e-5
        assert _rel(x.grad, r.grad * k_bwd.float() / (1 - p)) < 1e-6
        return k_bwd
    k0 = ln_keep(3)
    assert abs(k0.float().mean().item() - (1 - p)) < 0.005
    assert torch.equal(ln_keep(3), k0)
    assert 0.7 < (ln_keep(4) == k0).float().mean().item() < 0.9
    for dims in ((0, 1), (0, 2), (1, 2)):
        assert k0.float().mean(dims).std().item() > 0
    q8 = k0.view(B, D // 8, 8, T).float()
    for e in range(1, 8):
        assert abs((q8[:, :, 0] * q8[:, :, e]).mean().item() - (1 - p) ** 2) < 0.01
    assert abs(ln_keep(5, d=60, b=3).float().mean().item() - (1 - p)) < 0.01          # D not a multiple of 8
    assert abs(ln_keep(5, d=10, b=16).float().mean().item() - (1 - p)) < 0.015
    S.advance_dropout_step(torch.device(dev, 0))
    assert 0.7 < (ln_keep(3) == k0).float().mean().item() < 0.9

    # attention: out is linear in v — one-hot values read the dropped probabilities P' = P * keep / (1 - p) column by column
    dh = D // H
    qkv = torch.randn(B, 3 * D, T, device=dev)
    eye_cols = []
    lib = S._lib()
    probs = torch.empty(B * H, T, T, device=dev)
    out = torch.empty(B, D, T, device=dev)
    pk = torch.zeros(B * H, T, T, device=dev)
    d = S.Drop(p, 9, 'attn', (B * H, T, T), qkv.device)
    for j0 in range(0, T, dh):
        qv = qkv.clone()
        v = qv[:, 2 * D:].view(B, H, dh, T)
        v.zero_()
        for c in range(min(dh, T - j0)):
            v[:, :, c, j0 + c] = 1.0
        L.check(lib.dynmm_mha_drop_fwd(qv.data_ptr(), out.data_ptr(), probs.data_ptr(), B, D, T, H, S._drop_arg(d),
                                       torch.cuda.current_stream().cuda_stream), 'mha_fwd')
        o = out.view(B * H, dh, T)
        for c in range(min(dh, T - j0)):
            pk[:, :, j0 + c] = o[:, c, :]
    ratio = pk / probs                                                                # keep / (1 - p)
    keep = ratio > 0.5
    assert _rel(ratio[keep], torch.full_like(ratio[keep], 1 / (1 - p))) < 1e-4 and float(ratio[~keep].abs().max()) < 1e-6
    assert abs(keep.float().mean().item() - (1 - p)) < 0.005                          # 100 000 draws
    for dims in ((0, 1), (0, 2), (1, 2)):
        assert keep.float().mean(dims).std().item() > 0
    k8 = keep[:, :, :48].reshape(B * H, T, 6, 8).float()
    for e in range(1, 8):
        assert abs((k8[..., 0] * k8[..., e]).mean().item() - (1 - p) ** 2) < 0.01
    # the backward regenerates them: same arithmetic in torch with `keep` injected
    qr = qkv.clone().requires_grad_(True)
    q, k, v = (t.reshape(B * H, dh, T).transpose(1, 2) for t in qr.split(D, dim=1))
    ref = ((torch.softmax(q @ k.transpose(1, 2) / dh ** 0.5, dim=-1) * keep.float() / (1 - p)) @ v).transpose(1, 2).reshape(B, D, T)
    g = torch.randn(B, D, T, device=dev)
    ref.backward(g)
    qc = qkv.clone().requires_grad_(True)
    o2 = S._MHACore.apply(qc, H, d)
    o2.backward(g)
    assert _rel(o2, ref) < 1e-5 and _rel(qc.grad, qr.grad) < 2e-5

    # LayerNorm backward: workspace entry == two-pass entry (generator path, D = 60: a partial channel block)
    Bq, Dq = 5, 60
    x, r, g = (torch.randn(Bq, Dq, T, device=dev) for _ in range(3))
    gam = torch.rand(Dq, device=dev) + 0.5
    y, mean, rstd = torch.empty_like(x), torch.empty(Bq * T, device=dev), torch.empty(Bq * T, device=dev)
    dd = S.Drop(p, 11, 'dropout1', x.shape, x.device)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.dynmm_layernorm_drop_fwd(x.data_ptr(), r.data_ptr(), gam.data_ptr(), gam.data_ptr(), y.data_ptr(),
                                         mean.data_ptr(), rstd.data_ptr(), Bq, Dq, T, 1e-5, S._drop_arg(dd), st), 'ln_fwd')
    res = []
    for use_ws in (False, True):
        dx, dres, dg, db = torch.empty_like(x), torch.empty_like(x), torch.empty(Dq, device=dev), torch.empty(Dq, device=dev)
        nb = lib.dynmm_layernorm_bwd_workspace_bytes(Bq, Dq, T) if use_ws else 0
        ws = torch.empty(max(nb // 4, 1), device=dev)
        L.check(lib.dynmm_layernorm_drop_bwd_ws(g.data_ptr(), x.data_ptr(), r.data_ptr(), gam.data_ptr(), mean.data_ptr(),
                                                rstd.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                                Bq, Dq, T, S._drop_arg(dd), ws.data_ptr() if use_ws else None, nb, st), 'ln_bwd')
        res.append((dx, dres, dg, db))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert _rel(res[1][2], res[0][2]) < 1e-5 and _rel(res[1][3], res[0][3]) < 1e-5
    assert lib.dynmm_layernorm_drop_bwd_ws(g.data_ptr(), x.data_ptr(), r.data_ptr(), gam.data_ptr(), mean.data_ptr(),
                                           rstd.data_ptr(), dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                           Bq, Dq, T, None, ws.data_ptr(), 16, st) == L.DYNMM_EINVAL       # workspace too small


@pytest.mark.gpu
@pytest.mark.parametrize('B,D,T,p', [(4, 120, 50, 0.1), (3, 60, 50, 0.0), (5, 10, 13, 0.1)])
def test_residual_gradient_link_equals_autograd_accumulation(B, D, T, p):
    """An encoder layer's input feeds in_proj and norm1's residual input; with LINK_RESIDUAL norm1's backward hands the residual
    branch's gradient to in_proj's input-gradient epilogue (ops.GradLink) instead of autograd's add pass: same output, input
    gradient and parameter gradients, under plain autograd and under the in-place gradient protocol."""
    from dynmm_amd import engine, ops, ops_seq as S
    from dynmm_amd.nn import affect as A
    torch.manual_seed(7 * B + D)
    layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=5, dim_feedforward=64, dropout=p).cuda().train()
    h0 = torch.randn(B, D, T, device='cuda')
    gy = torch.randn(B, D, T, device='cuda')

    def run(link, direct):
        A.LINK_RESIDUAL = link
        S.MASKS = _Masks(p, 3, 'cuda') if p > 0 else None
        layer._dynmm_sites = None
        for q in layer.parameters():
            q.grad = torch.zeros_like(q) if direct else None
        pre = torch.nn.Parameter(torch.ones(1, D, 1, device='cuda'))     # h is a non-leaf, as inside a Transformer
        try:
            h = h0 * pre
            if direct:
                with engine.direct_gradients(False):
                    ops.touched_reset()
                    y = A.encoder_layer(h, layer, 5)
                    y.backward(gy)
                    ops.flush_wgrad_groups()
            else:
                y = A.encoder_layer(h, layer, 5)
                y.backward(gy)
            torch.cuda.synchronize()
        finally:
            A.LINK_RESIDUAL, S.MASKS = True, None
        return y.detach(), pre.grad.clone(), {n: q.grad.clone() for n, q in layer.named_parameters()}

    y0, dh0, g0 = run(False, False)
    for direct in (False, True):
        y1, dh1, g1 = run(True, direct)
        assert torch.equal(y1, y0) and _rel(dh1, dh0) < 2e-5, (direct, _rel(dh1, dh0))
        for n in g0:
            assert _rel(g1[n], g0[n]) < 2e-4, (direct, n, _rel(g1[n], g0[n]))


@pytest.mark.gpu
@pytest.mark.parametrize('B,D,T,F,heads,p', [(6, 120, 50, 2048, 5, 0.1), (3, 60, 50, 2048, 5, 0.1), (5, 120, 50, 2048, 5, 0.0),
                                              (2, 20, 7, 64, 5, 0.2), (3, 124, 11, 96, 4, 0.1), (9, 10, 50, 2048, 5, 0.1), (4, 35, 13, 64, 5, 0.1), (70, 120, 50, 256, 5, 0.1)])
def test_fused_feed_forward_block_equals_the_layer_by_layer_path(B, D, T, F, heads, p):
    """csrc/seq_ffn.hip (linear1 -> ReLU -> dropout -> linear2 in one launch, hidden-unit split summed inside norm2; backward:
    one launch for both data gradients) against the same encoder layer run as separate linear / dropout / LayerNorm launches,
    with the same injected keep flags: output, input gradient, every parameter gradient — under plain autograd and under the
    in-place gradient protocol of the training step (queued weight gradients)."""
    from dynmm_amd import engine, ops, ops_seq as S
    from dynmm_amd.nn import affect as A
    torch.manual_seed(B * 1000 + D)
    layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=heads, dim_feedforward=F, dropout=p).cuda().train()
    h0 = torch.randn(B, D, T, device='cuda')
    gy = torch.randn(B, D, T, device='cuda')
    assert S.ffn_fused_ok(h0, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias)

    def run(fused, direct):
        S.FFN_FUSED = fused
        S.MASKS = _Masks(p, 5, 'cuda') if p > 0 else None
        layer._dynmm_sites = None
        for q in layer.parameters():
            q.grad = torch.zeros_like(q) if direct else None
        h = h0.clone().requires_grad_(True)
        try:
            if direct:
                with engine.direct_gradients(False):
                    ops.touched_reset()
                    y = A.encoder_layer(h, layer, heads)
                    y.backward(gy)
                    ops.flush_wgrad_groups()
            else:
                y = A.encoder_layer(h, layer, heads)
                y.backward(gy)
            torch.cuda.synchronize()
        finally:
            S.FFN_FUSED, S.MASKS = True, None
        return y.detach(), h.grad, {n: q.grad.clone() for n, q in layer.named_parameters()}

    y0, dh0, g0 = run(False, False)
    for direct in (False, True):
        y1, dh1, g1 = run(True, direct)
        assert _rel(y1, y0) < 2e-5 and _rel(dh1, dh0) < 1e-4, (direct, _rel(y1, y0), _rel(dh1, dh0))
        for n in g0:
            assert _rel(g1[n], g0[n]) < 2e-4, (direct, n, _rel(g1[n], g0[n]))


@pytest.mark.gpu
def test_fused_feed_forward_dropout_generator():
    """The Philox path of the fused block (one call per four hidden units): with W1 = 0, b1 = 1 the stored hidden activation
    IS the keep pattern.  Keep rate, values in {0, 1/(1-p)}, new decisions per step, the same decisions for the same
    (seed, site, step), no repetition along any axis, and a backward that masks exactly the dropped units."""
    from dynmm_amd import ops, ops_seq as S
    ops.manual_seed(99)
    B, D, T, F, p = 8, 120, 50, 2048, 0.1
    layer = torch.nn.TransformerEncoderLayer(d_model=D, nhead=5, dim_feedforward=F, dropout=p).cuda().train()
    with torch.no_grad():
        layer.linear1.weight.zero_()
        layer.linear1.bias.fill_(1.0)
    h = torch.randn(B, D, T, device='cuda')
    lib = S._lib()
    ns = lib.dynmm_ffn_nsplit(B, D, T, F)

    def hidden(site):
        hid = torch.empty(B, F, T, device='cuda')
        parts = torch.empty(ns, B, D, T, device='cuda')
        d = S.Drop(p, site, 'dropout', (B, F, T), h.device)
        S.L.check(lib.dynmm_ffn_fwd(h.data_ptr(), layer.linear1.weight.data_ptr(), layer.linear1.bias.data_ptr(),
                                    layer.linear2.weight.data_ptr(), hid.data_ptr(), parts.data_ptr(), B, D, T, F, ns,
                                    S._drop_arg(d), torch.cuda.current_stream().cuda_stream), 'ffn_fwd')
        torch.cuda.synchronize()
        ref = torch.einsum('df,bft->bdt', layer.linear2.weight.detach().double(), hid.double())
        assert _rel(parts.double().sum(0), ref) < 1e-5
        return hid
    k0 = hidden(3)
    vals = torch.unique(k0)
    assert vals.numel() == 2 and vals[0].item() == 0 and abs(vals[1].item() - 1 / (1 - p)) < 1e-6
    keep = k0 > 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.002                    # 819 200 draws: sigma = 0.0003
    assert torch.equal(hidden(3) > 0, keep)
    assert 0.7 < ((hidden(4) > 0) == keep).float().mean().item() < 0.9          # independent sites agree on 0.82
    S.advance_dropout_step(h.device)
    assert 0.7 < ((hidden(3) > 0) == keep).float().mean().item() < 0.9
    for dims in ((0, 1), (0, 2), (1, 2)):
        assert keep.float().mean(dims).std().item() > 0
    # the four units of a Philox call are independent of each other
    q = keep.view(B, F // 4, 4, T).float()
    assert abs((q[:, :, 0] * q[:, :, 1]).mean().item() - (1 - p) ** 2) < 0.003


@pytest.mark.gpu
def test_dynmm_affect_training_mode_matches_oracle_with_injected_dropout():
    """DynMMNetV2 in training mode (p = 0.1 at 4 sites x 5 layers x 5 transformers = 100 dropout sites), keep flags injected on
    both sides: forward, objective and every gradient."""
    from dynmm_amd import ops_seq as S
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    ref = O.fill_(O.DynMMNetV2(0.7, False), seed=1)
    mine = A.DynMMNetV2(0.7, False)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().train()
    inputs, y = O.synth_batch(4, seed=3)
    mr, mh = _Masks(0.1, 21), _Masks(0.1, 21, 'cuda')
    O.Transformer.dropout_masks = (0.1, mr)
    S.MASKS = mh
    try:
        out_r, aux_r, _ = ref(inputs)
        tot_r, _ = O.train_objective(out_r, aux_r, y, 0.3)
        tot_r.backward()
        out, aux = mine([[x.cuda() for x in inputs[0]], inputs[1]])
        tot = (out - y.cuda()).abs().mean() + 0.3 * aux
        tot.backward()
        torch.cuda.synchronize()
    finally:
        O.Transformer.dropout_masks = None
        S.MASKS = None
    assert mr.n == mh.n == 100 and mr.names == mh.names
    assert _rel(out, out_r) < 2e-4 and abs(tot.item() - tot_r.item()) < 1e-5
    gr = dict(ref.named_parameters())
    errs = {}
    for n, p_ in mine.named_parameters():
        if gr[n].grad is None or gr[n].grad.abs().max() < 1e-9:
            continue
        errs[n] = ((p_.grad.cpu().double() - gr[n].grad.double()).norm() / gr[n].grad.double().norm()).item()
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-3, (worst, errs[worst])
    assert np.median(list(errs.values())) < 2e-4
    # and it differs from the eval-mode arithmetic (dropout really acted)
    mine.eval()
    with torch.no_grad():
        out_e, _ = mine([[x.cuda() for x in inputs[0]], inputs[1]])
    assert _rel(out_e, out_r) > 1e-3


@pytest.mark.gpu
def test_affect_train_step_draws_new_masks_every_replay():
    """AffectTrainStep in training mode under hipGraph replay: the captured step must not reuse its dropout decisions —
    with lr = 0 (weights frozen) two replays on the same batch give different losses, and re-seeding reproduces the first."""
    from dynmm_amd import ops, ops_seq as S
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    mine = A.DynMMNetV2(1.0, False)
    mine.load_state_dict(O.fill_(O.DynMMNetV2(1.0, False), seed=2).state_dict())
    mine = mine.cuda().train()
    inputs, y = O.synth_batch(5, seed=10)
    inputs = [[x.cuda() for x in inputs[0]], inputs[1]]
    step = A.AffectTrainStep(mine, lr=0.0, weight_decay=0.0, lossw=0.2, use_graph=True)
    ops.manual_seed(77)
    S.dropout_step(inputs[0][0].device).zero_()
    losses = [step(inputs, y.cuda())['total'].item() for _ in range(3)]
    assert len({round(v, 7) for v in losses}) == 3, losses
    step.opt.check_finite()


@pytest.mark.gpu
@pytest.mark.parametrize('kind,hard', [('v2', False), ('v2', True), ('v1', False)])
def test_dynmm_affect_model_matches_oracle(kind, hard):
    """Whole model, eval-mode arithmetic (dropout = 0), forward and every trainable gradient, HIP vs oracle."""
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    ref = O.fill_(O.DynMMNetV2(0.7, hard) if kind == 'v2' else O.DynMMNet(0.7, hard), seed=1)
    mine = (A.DynMMNetV2(0.7, hard) if kind == 'v2' else A.DynMMNet(0.7, hard, freeze=False))
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().eval()                # eval(): no dropout, like the oracle's dropout=0 layers
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
        mine = mine.cuda().eval()            # the optimiser arithmetic is what is compared here: dropout off on both sides
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
            # step 2 sits behind one Adam update (lr * sign(g) whatever |g|: an element whose gradient is rounding noise moves by
            # 2 lr when a summation order changes): relative to the objective's size there (round 6's LayerNorm re-ordering:
            # 7e-5 of 2.95)
            tol = 2e-5 if it == 0 else 2e-4 * max(1.0, abs(tot_r.item()))
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


@pytest.mark.gpu
def test_configs4_batch128_properties():
    """BASELINE configs[4] per GPU at its OWN size (DynMMNetV2, batch 128, T = 50; checker = the self-written oracle:
    PARITY UNPINNED).  Eval-mode arithmetic is per sample, so
      * the batch-128 forward equals two batch-64 forwards of its halves, sample by sample;
      * a batch permutation permutes outputs and gate weights and leaves the mean gate regulariser unchanged;
      * the first 4 samples agree with the oracle (2e-4, the small-batch bar);
      * one full training step (dropout on, clip + AdamW, hipGraph off) is finite and touches every trainable parameter."""
    from dynmm_amd.nn import affect as A
    from oracle import affect_oracle as O
    ref = O.fill_(O.DynMMNetV2(0.7, False), seed=1)
    mine = A.DynMMNetV2(0.7, False)
    mine.load_state_dict(ref.state_dict())
    mine = mine.cuda().eval()
    inputs, y = O.synth_batch(128, seed=5)
    xs, lens = [x.cuda() for x in inputs[0]], inputs[1]
    assert xs[0].shape[:2] == (128, 50)

    def sub(idx):
        li = idx if isinstance(idx, slice) else idx.cpu()
        return [[x[idx].contiguous() for x in xs], [l[li] for l in lens]]
    with torch.no_grad():
        out, aux = mine([xs, lens])
        oa, aux_a = mine(sub(slice(0, 64)))
        ob, aux_b = mine(sub(slice(64, 128)))
        perm = torch.randperm(128, generator=torch.Generator().manual_seed(3)).cuda()
        op, aux_p = mine(sub(perm))
    scale = out.abs().max().item()
    assert bool(torch.isfinite(out).all())
    assert (torch.cat([oa, ob]) - out).abs().max().item() <= 2e-6 * scale
    assert abs(0.5 * (aux_a.item() + aux_b.item()) - aux.item()) < 1e-6
    assert (op - out[perm]).abs().max().item() <= 2e-6 * scale and abs(aux_p.item() - aux.item()) < 1e-6
    with torch.no_grad():
        out_r, aux_r, _ = ref([[x[:4] for x in inputs[0]], [l[:4] for l in lens]])
        o4, _ = mine(sub(slice(0, 4)))
    assert _rel(o4, out_r) < 2e-4
    assert (o4 - out[:4]).abs().max().item() <= 2e-6 * scale
    mine.train()
    step = A.AffectTrainStep(mine, lr=1e-4, weight_decay=0.01, lossw=0.2, use_graph=False)
    res = step([xs, lens], y.cuda())
    torch.cuda.synchronize()
    assert np.isfinite(res['total'].item())
    step.opt.check_finite()
    for n_, p in mine.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n_
