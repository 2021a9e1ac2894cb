"""GPU: each HIP op (through the C ABI, via dynmm_amd.ops) against a plain PyTorch fp32 CPU reference
of the same op, forward and backward, on the shape classes of the hot path."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5      # fp32 MFMA is an exact-f32 fma chain; differences are summation-order only
GTOL = 2e-4     # gradients: long reductions (N*H*W terms) in a different order


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.fixture(scope='module')
def ops():
    from dynmm_amd import ops as o
    from dynmm_amd import lib
    lib.load()
    return o


CONV_CASES = [
    # N, Ci, H, W, Co, k, stride, pad, bias, act
    (2, 64, 24, 32, 64, (3, 1), (1, 1), (1, 0), True, 'relu'),
    (2, 64, 24, 32, 64, (1, 3), (1, 1), (0, 1), True, None),
    (2, 64, 24, 32, 128, (3, 1), (2, 1), (1, 0), True, 'relu'),
    (2, 128, 12, 32, 128, (1, 3), (1, 2), (0, 1), True, None),
    (2, 64, 24, 32, 128, (1, 1), (2, 2), (0, 0), False, None),
    (3, 64, 17, 23, 64, (3, 3), (1, 1), (1, 1), False, None),
    (2, 64, 24, 32, 128, (3, 3), (2, 2), (1, 1), False, None),
    (2, 256, 6, 8, 512, (3, 1), (2, 1), (1, 0), True, 'relu'),
    (2, 512, 3, 4, 512, (1, 3), (1, 1), (0, 1), True, None),
    (2, 3, 48, 64, 64, (7, 7), (2, 2), (3, 3), False, None),
    (2, 1, 48, 64, 64, (7, 7), (2, 2), (3, 3), False, None),
    (2, 128, 24, 32, 40, (3, 3), (1, 1), (1, 1), True, None),
    (2, 128, 12, 16, 40, (1, 1), (1, 1), (0, 0), True, None),
    (2, 8, 28, 38, 8, (5, 5), (2, 2), (0, 0), True, None),
    (2, 1024, 3, 4, 128, (1, 1), (1, 1), (0, 0), False, None),
    (1, 64, 120, 160, 64, (3, 1), (1, 1), (1, 0), True, 'relu'),
    (2, 192, 10, 12, 64, (3, 3), (1, 1), (1, 1), True, None),       # Ci = 3 x 64: wgrad tap groups, 64x192 tile
    (1, 64, 7, 9, 256, (1, 3), (1, 1), (0, 1), True, 'relu'),        # ragged M (63 pixels), bias via wgrad
    (3, 128, 9, 11, 72, (3, 1), (2, 1), (1, 0), True, None),         # Co tail 72 = 64 + 8 in a 128-wide tile
    (2, 64, 6, 40, 512, (1, 3), (1, 2), (0, 1), True, None),         # 128x32 tile (small grid), stride-2 dgrad
    (2, 64, 15, 21, 128, (3, 1), (2, 1), (1, 0), True, None),        # odd H: strided dgrad without sub-pixel classes
    (2, 64, 6, 10, 128, (1, 3), (1, 2), (0, 1), False, None),        # parity classes of 30 pixels: tiles straddle classes
    (3, 128, 8, 8, 64, (3, 3), (2, 2), (1, 1), False, None),         # 4 parity classes, 3x3: 1/2/2/4 live taps
    (2, 32, 21, 37, 8, (5, 5), (2, 2), (0, 0), True, None),          # gate-conv class: direct (no-MFMA) kernel
    (1, 128, 30, 171, 6, (5, 5), (2, 2), (0, 0), True, 'relu'),      # ... 84 output columns = 2 column tiles, Co = 6
    # operand-ring kernels (conv_igemm_v5.hip): stride 1, same padding, Ci % 32 == 0 >= 64, Co % 64 == 0, W % 4 == 0
    (3, 128, 10, 12, 128, (3, 3), (1, 1), (1, 1), False, 'relu'),    # 120-pixel images: every 64-pixel tile straddles two
    (2, 256, 30, 40, 256, (1, 3), (1, 1), (0, 1), True, None),       # halo taps across row ends, 2 channel tiles
    (5, 64, 6, 8, 64, (3, 1), (1, 1), (1, 0), True, 'relu'),         # 48-pixel images under a 128-pixel tile, ragged M = 240
    (2, 64, 12, 16, 128, (1, 1), (1, 1), (0, 0), False, None),       # 1x1: no padding at all, 4 K-steps (ring depth)
    (2, 512, 15, 20, 512, (3, 1), (1, 1), (1, 0), True, None),       # K = 1536: 96 K-steps
    (1, 96, 8, 16, 192, (3, 3), (1, 1), (1, 1), True, None),         # Ci = 6 chunks, Co = 3 x 64
    (4, 512, 15, 20, 512, (1, 3), (1, 1), (0, 1), True, 'relu'),     # 76 tiles, 96 K-steps
    (2, 256, 30, 40, 256, (3, 3), (1, 1), (1, 1), False, None),      # 76 tiles, K = 2304
    # three-tap weight-gradient kernel (conv_wgrad_v6.hip): stride 1, same padding, W % 4 == 0 >= 16, Ci, Co % 64 == 0
    (3, 128, 15, 20, 128, (1, 3), (1, 1), (0, 1), True, None),       # M = 900: last step has one live quad; rows of 20 pixels
    (5, 64, 17, 20, 64, (3, 1), (1, 1), (1, 0), True, 'relu'),       # vertical taps, 64-row tile, M = 1700, odd H
    (2, 128, 9, 16, 256, (3, 1), (1, 1), (1, 0), True, None),        # narrowest rows (one step = one row), 2 co tiles x 2 ci tiles
    (1, 192, 8, 24, 64, (1, 3), (1, 1), (0, 1), False, None),        # 3 ci tiles, no bias, a step straddles rows
    # stride-2 three-tap weight-gradient kernel (conv_wgrad_s2.hip): the first block of a stage, Wo % 4 == 0 >= 16, Ci, Co % 64 == 0
    (3, 256, 30, 40, 512, (3, 1), (2, 1), (1, 0), True, None),       # rows of 40: steps straddle rows; 4 x 4 tiles, several splits
    (3, 512, 15, 40, 512, (1, 3), (1, 2), (0, 1), True, None),       # Wo = 20, 300-pixel images: ragged M = 900, steps straddle images
    (2, 64, 24, 32, 64, (3, 1), (2, 1), (1, 0), False, 'relu'),      # 64-row tile, no bias
    (2, 64, 24, 64, 64, (1, 3), (1, 2), (0, 1), True, None),         # 64-row tile (3 workgroups per CU)
    (1, 64, 120, 160, 128, (3, 1), (2, 1), (1, 0), True, None),      # the stage-2 shape at batch 1
    (1, 128, 60, 160, 128, (1, 3), (1, 2), (0, 1), True, None),
    # ... 3x3 filters on the same kernel, one vertical tap per workgroup (round 5: BasicBlock / decoder conv3x3 weight gradients)
    (3, 64, 17, 20, 64, (3, 3), (1, 1), (1, 1), True, None),         # 64-row tile, odd H, M = 1020: steps straddle rows AND images
    (2, 128, 9, 16, 256, (3, 3), (1, 1), (1, 1), True, 'relu'),      # narrowest rows: every step is one image row, 2 x 2 x 3 k-tiles
    (1, 64, 2, 32, 128, (3, 3), (1, 1), (1, 1), False, None),        # H = 2: the outer taps see one live row each
    (2, 128, 24, 32, 40, (3, 3), (1, 1), (1, 1), True, None),        # conv_out: 40 rows in a 64-row tile (rows past Co read zeros)
    (3, 64, 9, 20, 72, (1, 3), (1, 1), (0, 1), True, None),          # 72 = 64 + 8 rows: a full tile and a tail tile
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d_fwd_bwd(ops, case):
    N, Ci, H, W, Co, k, s, p, bias, act = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, *k, seed=2, scale=(Ci * k[0] * k[1]) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1) if bias else None
    x_requires = Ci > 3
    xr = x.clone().requires_grad_(x_requires)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    y_ref = F.conv2d(xr, wr, br, s, p)
    if act == 'relu':
        y_ref = F.relu(y_ref)
    gy = rnd(*y_ref.shape, seed=4)
    y_ref.backward(gy)

    xg = x.cuda().requires_grad_(x_requires)
    wg = w.cuda().requires_grad_(True)
    bg = b.cuda().requires_grad_(True) if bias else None
    y = ops.conv2d(xg, wg, bg, s, p, act)
    assert rel(y, y_ref) < TOL
    y.backward(gy.cuda())
    assert rel(wg.grad, wr.grad) < GTOL
    if bias:
        assert rel(bg.grad, br.grad) < GTOL
    if x_requires:
        assert rel(xg.grad, xr.grad) < GTOL


@pytest.mark.parametrize('k,p', [((1, 3), (0, 1)), ((3, 1), (1, 0))])
def test_conv2d_tensors_off_the_16_byte_grid(ops, k, p):
    """The operand-ring kernels and the three-tap weight-gradient kernel move 16 bytes per lane; a contiguous tensor that starts
    4 bytes into its storage (a view of a larger buffer) must take the kernels that do not, with the same results — forward,
    input gradient (gy misaligned), weight / bias gradient (x and gy misaligned)."""
    N, Ci, H, W, Co = 2, 128, 12, 32, 128
    x, w, b = rnd(N, Ci, H, W, seed=1), rnd(Co, Ci, *k, seed=2, scale=(Ci * 3) ** -0.5), rnd(Co, seed=3, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y_ref = F.conv2d(xr, wr, br, 1, p)
    gy = rnd(*y_ref.shape, seed=4)
    y_ref.backward(gy)

    def off_grid(t):
        buf = torch.empty(t.numel() + 1, device='cuda')
        v = buf[1:].view(t.shape)
        v.copy_(t)
        assert v.is_contiguous() and v.data_ptr() % 16 == 4
        return v
    xg = off_grid(x).requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv2d(xg, wg, bg, 1, p)
    assert rel(y, y_ref) < TOL
    y.backward(off_grid(gy))
    assert rel(wg.grad, wr.grad) < GTOL and rel(bg.grad, br.grad) < GTOL and rel(xg.grad, xr.grad) < GTOL


def test_conv2d_random_shapes_across_kernel_families(ops):
    """40 seeded random geometries around the eligibility edges of the kernel families (operand-ring forward / input gradient:
    stride 1, same padding, Ci % 32 == 0 >= 64, Co % 64 == 0, W % 4 == 0; three-tap weight gradient: W >= 16, Ci, Co % 64 == 0;
    everything else: the round-2 tiles), each against torch in fp64: forward, input, weight and bias gradients."""
    rng = np.random.default_rng(20260928)
    done = set()
    for it in range(40):
        k = [(1, 3), (3, 1), (3, 3), (1, 1)][int(rng.integers(4))]
        ci = int(rng.choice([32, 64, 96, 128, 192]))
        co = int(rng.choice([40, 64, 72, 128, 192]))
        n = int(rng.integers(1, 5))
        h = int(rng.integers(3, 14))
        w_ = int(rng.choice([8, 12, 16, 20, 22, 24, 36]))
        stride = (1, 1) if rng.random() < 0.8 else ((2, 1) if k[0] == 3 else (1, 2) if k[1] == 3 else (2, 2))
        pad = (k[0] // 2, k[1] // 2)
        bias = bool(rng.integers(2))
        x = torch.from_numpy(rng.standard_normal((n, ci, h, w_)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((co, ci, *k)) * (ci * k[0] * k[1]) ** -0.5).astype(np.float32))
        b = torch.from_numpy((0.1 * rng.standard_normal(co)).astype(np.float32)) if bias else None
        xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
        br = b.double().requires_grad_(True) if bias else None
        y_ref = F.conv2d(xr, wr, br, stride, pad)
        gy = torch.from_numpy(rng.standard_normal(tuple(y_ref.shape)).astype(np.float32))
        y_ref.backward(gy.double())
        xg, wg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
        bg = b.cuda().requires_grad_(True) if bias else None
        y = ops.conv2d(xg, wg, bg, stride, pad)
        y.backward(gy.cuda())
        tag = (n, ci, h, w_, co, k, stride, bias)
        assert rel(y, y_ref) < TOL, tag
        assert rel(xg.grad, xr.grad) < GTOL and rel(wg.grad, wr.grad) < GTOL, tag
        if bias:
            assert rel(bg.grad, br.grad) < GTOL, tag
        g = ops._geom(xg, None, wg, stride, pad)
        import ctypes as C
        done.add((int(ops._lib().dynmm_conv2d_uses_operand_ring(C.byref(g), 0)), int(ops._lib().dynmm_conv2d_wgrad_variant(C.byref(g)))))
    assert {v for _, v in done} >= {0, 4, 6} and {r for r, _ in done} == {0, 1}, done      # every family was exercised


def test_conv2d_dual_input(ops):
    """GlobalGate's first conv: cat(rgb, depth) is never materialised."""
    a, b2 = rnd(2, 64, 24, 32, seed=5), rnd(2, 64, 24, 32, seed=6)
    w = rnd(8, 128, 5, 5, seed=7, scale=0.02)
    bias = rnd(8, seed=8, scale=0.1)
    ar, br_, wr, biasr = [t.clone().requires_grad_(True) for t in (a, b2, w, bias)]
    y_ref = F.conv2d(torch.cat([ar, br_], 1), wr, biasr, 2)
    gy = rnd(*y_ref.shape, seed=9)
    y_ref.backward(gy)
    ag, bg, wg, biasg = [t.cuda().requires_grad_(True) for t in (a, b2, w, bias)]
    y = ops.conv2d(ag, wg, biasg, 2, 0, None, x2=bg)
    assert rel(y, y_ref) < TOL
    y.backward(gy.cuda())
    for got, ref in ((ag.grad, ar.grad), (bg.grad, br_.grad), (wg.grad, wr.grad), (biasg.grad, biasr.grad)):
        assert rel(got, ref) < GTOL


@pytest.mark.parametrize('case', [(2, 128, 64, 24, 32, 8, True),        # the gate conv's own geometry class: rgb + depth, bias
                                  (3, 32, 32, 21, 40, 6, True),         # single input, Co = 6, odd H, Wo = 18
                                  (17, 128, 64, 13, 16, 8, False),      # 17 images: image groups of 2 (ragged last group), no bias
                                  (2, 64, 32, 75, 164, 8, True)])       # Wo = 80: the last column group is full; 3 passes of rows
def test_gate_conv_weight_gradient_on_the_vector_alus(ops, case):
    """dynmm_conv2d_wgrad for the gate head's first convolution (…globalgate.py:378-386: 5x5, stride 2, <= 8 output channels, the
    rgb | depth pair as two inputs): csrc/conv_small.hip conv_co8_wgrad_kernel — weight AND bias gradient in one launch, vs
    float64, bit-identical between two calls, and the library reports the variant."""
    import ctypes as C
    from dynmm_amd import lib as L
    lib = L.load()
    N, Ci, split, H, W, Co, bias = case
    st = torch.cuda.current_stream().cuda_stream
    x = rnd(N, Ci, H, W, seed=1)
    w0 = torch.empty(Co, Ci, 5, 5)
    xa, xb = (x[:, :split].contiguous().cuda(), x[:, split:].contiguous().cuda()) if split < Ci else (x.cuda(), None)
    g = ops._geom(xa, xb, w0, (2, 2), (0, 0))
    assert lib.dynmm_conv2d_wgrad_variant(C.byref(g)) == 8
    dy = rnd(N, Co, g.Ho, g.Wo, seed=2)
    wd = torch.zeros(Co, Ci, 5, 5, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, None, 2).backward(dy.double())
    dyc = dy.cuda()

    def run():
        dw = torch.full((Co, Ci, 5, 5), float('nan'), device='cuda')
        db = torch.full((Co,), float('nan'), device='cuda') if bias else None
        nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
        ws = torch.empty(max(nbytes // 4, 1), device='cuda')
        L.check(lib.dynmm_conv2d_wgrad(xa.data_ptr(), None if xb is None else xb.data_ptr(), dyc.data_ptr(), dw.data_ptr(),
                                       None if db is None else db.data_ptr(), ws.data_ptr(), nbytes, C.byref(g), st), 'wgrad')
        torch.cuda.synchronize()
        return dw, db
    dw, db = run()
    assert rel(dw, wd.grad) < GTOL
    if bias:
        assert rel(db, dy.double().sum((0, 2, 3))) < GTOL
    dw2, db2 = run()
    assert torch.equal(dw, dw2) and (not bias or torch.equal(db, db2))


@pytest.mark.parametrize('act', [None, 'relu'])
@pytest.mark.parametrize('with_res', [False, True])
def test_conv_fused_eval(ops, act, with_res):
    bn = torch.nn.BatchNorm2d(128, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(rnd(128, seed=1).abs() + 0.5)
        bn.bias.copy_(rnd(128, seed=2) * 0.1)
        bn.running_mean.copy_(rnd(128, seed=3) * 0.1)
        bn.running_var.copy_(rnd(128, seed=4).abs() + 0.5)
    bn.eval()
    x, w, cb = rnd(2, 128, 12, 16, seed=5), rnd(128, 128, 1, 3, seed=6, scale=0.05), rnd(128, seed=7, scale=0.1)
    res = rnd(2, 128, 12, 16, seed=8) if with_res else None
    with torch.no_grad():
        ref = bn(F.conv2d(x, w, cb, 1, (0, 1)))
        if with_res:
            ref = ref + res
        if act:
            ref = F.relu(ref)
        bng = torch.nn.BatchNorm2d(128, eps=1e-3).cuda()
        bng.load_state_dict(bn.state_dict())
        bng.eval()
        y = ops.conv2d_fused_eval(x.cuda(), w.cuda(), cb.cuda(), bng, act, res.cuda() if with_res else None, 1, (0, 1))
    assert rel(y, ref) < TOL


@pytest.mark.parametrize('shape', [(4, 64, 24, 32),        # HW = 768: three full 256-element groups per plane
                                   (3, 128, 15, 20),       # HW = 300: one full group + a ragged one
                                   (2, 64, 120, 160),      # HW = 19200: three 8192-element chunks per plane (the apply passes' grid)
                                   (5, 32, 2, 2)])         # HW = 4: a plane is a single lane
@pytest.mark.parametrize('training', [True, False])
def test_batchnorm_residual_relu_decisions_as_bits(ops, shape, training):
    """relu(BN(x) + residual) (resnet.py:136-147): the forward leaves the ReLU decisions as one bit per element and the two
    backward passes read them instead of the output tensor (dynmm_bn_apply / _bwd_reduce / _bwd_apply `relu_bits`).  Same output
    and — bit for bit — the same gradients as the path that reads y; against float64 at GTOL; the words themselves against the
    output's sign pattern."""
    N, C, H, W = shape
    x, res, gy = rnd(*shape, seed=1) * 2 + 0.3, rnd(*shape, seed=2), rnd(*shape, seed=3)
    bn = torch.nn.BatchNorm2d(C, eps=1e-3).cuda().train(training)
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=4).abs() + 0.5)
        bn.bias.copy_(rnd(C, seed=5) * 0.1)
        bn.running_mean.copy_(rnd(C, seed=6) * 0.1)
        bn.running_var.copy_(rnd(C, seed=7).abs() + 0.5)
    state = {k: v.clone() for k, v in bn.state_dict().items()}
    lib = ops._lib()
    old = ops.BN_RELU_BITS

    def run(flag):
        ops.BN_RELU_BITS = flag
        bn.load_state_dict(state)
        for q in bn.parameters():
            q.grad = None
        xi, ri = x.clone().cuda().requires_grad_(True), res.clone().cuda().requires_grad_(True)
        y = ops.batch_norm_act(xi, bn, 'relu', ri)
        saved = y.grad_fn.saved_tensors
        assert (saved[6] is not None) == flag and (saved[1] is None) == flag       # bits instead of y
        if flag:
            words = saved[6].cpu().numpy().view(np.uint64).reshape(N * C, -1, 4)
            pos = (y.detach() > 0).reshape(N * C, H * W).cpu().numpy()
            for plane in (0, N * C - 1):
                for e in (0, H * W - 1, (H * W) // 2):
                    grp, lane, j = e // 256, (e % 256) // 4, e % 4
                    assert bool((int(words[plane, grp, j]) >> lane) & 1) == bool(pos[plane, e]), (plane, e)
        y.backward(gy.cuda())
        torch.cuda.synchronize()
        return y.detach(), xi.grad, ri.grad, bn.weight.grad.clone(), bn.bias.grad.clone()
    try:
        a, b = run(False), run(True)
    finally:
        ops.BN_RELU_BITS = old
    for t, u in zip(a, b):
        assert torch.equal(t, u)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = state['weight'].cpu().double().requires_grad_(True), state['bias'].cpu().double().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, state['running_mean'].cpu().double(), state['running_var'].cpu().double(), gr, br, training,
                             0.1, 1e-3) + rr)
    yr.backward(gy.double())
    assert rel(b[0], yr) < TOL
    for t, ref in zip(b[1:], (xr.grad, rr.grad, gr.grad, br.grad)):
        assert rel(t, ref) < GTOL


@pytest.mark.parametrize('shape', [(4, 64, 24, 32), (3, 8, 27, 37), (4, 256, 1, 1), (2, 128, 5, 5)])
@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
@pytest.mark.parametrize('training', [True, False])
def test_batch_norm_act(ops, shape, act, training):
    C = shape[1]
    x = rnd(*shape, seed=1) * 2 + 0.3
    res = rnd(*shape, seed=2) if act == 'relu' else None
    bn = torch.nn.BatchNorm2d(C, eps=1e-3)
    with torch.no_grad():
        bn.weight.copy_(rnd(C, seed=3).abs() + 0.5)
        bn.bias.copy_(rnd(C, seed=4) * 0.1)
        bn.running_mean.copy_(rnd(C, seed=5) * 0.1)
        bn.running_var.copy_(rnd(C, seed=6).abs() + 0.5)
    bng = torch.nn.BatchNorm2d(C, eps=1e-3).cuda()
    bng.load_state_dict(bn.state_dict())
    bn.train(training)
    bng.train(training)
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if res is not None else None
    y_ref = bn(xr)
    if rr is not None:
        y_ref = y_ref + rr
    y_ref = {'relu': F.relu, 'tanh': torch.tanh, None: lambda t: t}[act](y_ref)
    gy = rnd(*shape, seed=7)
    y_ref.backward(gy)
    xg = x.cuda().requires_grad_(True)
    rg = res.cuda().requires_grad_(True) if res is not None else None
    y = ops.batch_norm_act(xg, bng, act, rg)
    assert rel(y, y_ref) < TOL
    y.backward(gy.cuda())
    assert rel(xg.grad, xr.grad) < GTOL
    assert rel(bng.weight.grad, bn.weight.grad) < GTOL
    assert rel(bng.bias.grad, bn.bias.grad) < GTOL
    if rg is not None:
        assert rel(rg.grad, rr.grad) < GTOL
    assert rel(bng.running_mean, bn.running_mean) < TOL
    assert rel(bng.running_var, bn.running_var) < TOL
    assert int(bng.num_batches_tracked) == int(bn.num_batches_tracked)


@pytest.mark.parametrize('hw', [(48, 64), (30, 44), (17, 23), (6, 8)])       # wide (even H, W % 8 == 0) / W % 4 / scalar paths
def test_maxpool_with_ties(ops, hw):
    x = F.relu(rnd(2, 64, *hw, seed=1))             # post-ReLU: many exact-zero ties
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    gy = rnd(*y_ref.shape, seed=2)
    y_ref.backward(gy)
    xg = x.cuda().requires_grad_(True)
    y = ops.max_pool_3x3_s2(xg)
    assert torch.equal(y.cpu(), y_ref.detach())
    y.backward(gy.cuda())
    assert rel(xg.grad, xr.grad) < 1e-6


@pytest.mark.parametrize('hw,out', [((15, 20), 1), ((15, 20), 5), ((3, 4), 5), ((5, 6), (1, 5)), ((27, 37), 1)])
def test_adaptive_avg_pool(ops, hw, out):
    x = rnd(2, 16, *hw, seed=1)
    xr = x.clone().requires_grad_(True)
    y_ref = F.adaptive_avg_pool2d(xr, out)
    gy = rnd(*y_ref.shape, seed=2)
    y_ref.backward(gy)
    xg = x.cuda().requires_grad_(True)
    y = ops.adaptive_avg_pool(xg, out)
    assert rel(y, y_ref) < TOL
    y.backward(gy.cuda())
    assert rel(xg.grad, xr.grad) < 1e-5


@pytest.mark.parametrize('hw', [(15, 20), (3, 4), (5, 6)])
def test_nearest_concat(ops, hw):
    x, y1, y5 = rnd(2, 32, *hw, seed=1), rnd(2, 16, 1, 1, seed=2), rnd(2, 16, 5, 5, seed=3)
    ts = [t.clone().requires_grad_(True) for t in (x, y1, y5)]
    ref = torch.cat([ts[0]] + [F.interpolate(t, hw, mode='nearest') for t in ts[1:]], 1)
    gy = rnd(*ref.shape, seed=4)
    ref.backward(gy)
    tg = [t.cuda().requires_grad_(True) for t in (x, y1, y5)]
    out = ops.nearest_concat(*tg)
    assert torch.equal(out.cpu(), ref.detach())
    out.backward(gy.cuda())
    for a, b in zip(tg, ts):
        assert rel(a.grad, b.grad) < 1e-5


@pytest.mark.parametrize('shape', [(2, 128, 15, 20), (2, 40, 24, 32), (1, 3, 5, 7)])
@pytest.mark.parametrize('with_skip', [False, True])
def test_upsample2x_dw3x3(ops, shape, with_skip):
    N, C, H, W = shape
    x, w, b = rnd(*shape, seed=1), rnd(C, 1, 3, 3, seed=2, scale=0.3), rnd(C, seed=3, scale=0.1)
    skip = rnd(N, C, 2 * H, 2 * W, seed=4) if with_skip else None
    xr, wr, br = [t.clone().requires_grad_(True) for t in (x, w, b)]
    sr = skip.clone().requires_grad_(True) if with_skip else None
    ref = F.conv2d(F.interpolate(xr, (2 * H, 2 * W), mode='nearest'), wr, br, 1, 1, groups=C)
    if with_skip:
        ref = ref + sr
    gy = rnd(*ref.shape, seed=5)
    ref.backward(gy)
    xg, wg, bg = [t.cuda().requires_grad_(True) for t in (x, w, b)]
    sg = skip.cuda().requires_grad_(True) if with_skip else None
    y = ops.upsample2x_dw3x3(xg, wg, bg, sg)
    assert rel(y, ref) < TOL
    y.backward(gy.cuda())
    assert rel(xg.grad, xr.grad) < GTOL
    assert rel(wg.grad, wr.grad) < GTOL
    assert rel(bg.grad, br.grad) < GTOL
    if with_skip:
        assert rel(sg.grad, sr.grad) < 1e-6


def _se_ref(x, p):
    s = F.adaptive_avg_pool2d(x, 1)
    s = torch.sigmoid(F.conv2d(F.relu(F.conv2d(s, p[0], p[1])), p[2], p[3]))
    return x * s


@pytest.mark.parametrize('use_se', [True, False])
@pytest.mark.parametrize('col', [None, 0, 3])
@pytest.mark.parametrize('shape', [(3, 64, 24, 32), (2, 512, 3, 4), (2, 128, 9, 11)])
def test_se_fuse_blend(ops, use_se, col, shape):
    N, C, H, W = shape
    rgb, depth = rnd(*shape, seed=1), rnd(*shape, seed=2)
    params = [rnd(C // 16, C, 1, 1, seed=3, scale=0.2), rnd(C // 16, seed=4, scale=0.1),
              rnd(C, C // 16, 1, 1, seed=5, scale=0.3), rnd(C, seed=6, scale=0.1),
              rnd(C // 16, C, 1, 1, seed=7, scale=0.2), rnd(C // 16, seed=8, scale=0.1),
              rnd(C, C // 16, 1, 1, seed=9, scale=0.3), rnd(C, seed=10, scale=0.1)]
    wcum = torch.rand(N, 4, generator=torch.Generator().manual_seed(11))

    def run(dev, fn):
        r, d = [t.detach().clone().to(dev).requires_grad_(True) for t in (rgb, depth)]
        ps = [p.detach().clone().to(dev).requires_grad_(True) for p in params]
        wc = wcum.detach().clone().to(dev).requires_grad_(True)
        out = fn(r, d, ps, wc)
        out.backward(rnd(*shape, seed=12).to(dev))
        return out, r, d, ps, wc

    def ref_fn(r, d, ps, wc):
        fused = (_se_ref(r, ps[:4]) + _se_ref(d, ps[4:])) if use_se else r + d
        if col is None:
            return fused
        w = wc[:, col].view(-1, 1, 1, 1)
        return w * r + (1 - w) * fused

    def hip_fn(r, d, ps, wc):
        return ops.se_fuse_blend(r, d, ps if use_se else None, None if col is None else wc, col or 0)

    o_ref, r_ref, d_ref, p_ref, w_ref = run('cpu', ref_fn)
    o, r, d, p, w = run('cuda', hip_fn)
    assert rel(o, o_ref) < TOL
    assert rel(r.grad, r_ref.grad) < GTOL
    assert rel(d.grad, d_ref.grad) < GTOL
    if use_se:
        for a, b in zip(p, p_ref):
            assert rel(a.grad, b.grad) < GTOL
    if col is not None:
        assert rel(w.grad, w_ref.grad) < GTOL


@pytest.mark.parametrize('temp', [1.0, 0.1, 0.001])
@pytest.mark.parametrize('hard', [False, True])
def test_gate_head(ops, temp, hard):
    from oracle import dynmm_oracle as O
    N, J = 6, 8
    pooled, fc = rnd(N, J, 1, 1, seed=1), rnd(5, J, 1, 1, seed=2)
    tab = torch.tensor(O.DEPTH_ENC_FLOP_R34)
    gw, gc, gl = rnd(N, 5, seed=3), rnd(N, 4, seed=4), torch.tensor(0.7)

    pr, fr = pooled.clone().requires_grad_(True), fc.clone().requires_grad_(True)
    w_ref = O.diff_softmax(F.conv2d(pr, fr), tau=temp, hard=hard, dim=1).squeeze(-1).squeeze(-1)
    wc_ref = torch.stack([w_ref[:, 0], w_ref[:, 0] + w_ref[:, 1], w_ref[:, 0] + w_ref[:, 1] + w_ref[:, 2],
                          1 - w_ref[:, 4]], 1)
    l_ref = (w_ref.mean(0) * tab).mean()
    ((w_ref * gw).sum() + (wc_ref * gc).sum() + l_ref * gl).backward()

    pg, fg = pooled.cuda().requires_grad_(True), fc.cuda().requires_grad_(True)
    w, wc, l = ops.gate_head(pg, fg, tab.cuda(), temp, hard)
    assert rel(w, w_ref) < 1e-5 and rel(wc, wc_ref) < 1e-5 and abs(l.item() - l_ref.item()) < 1e-5
    if hard:
        assert torch.equal(w.argmax(1).cpu(), w_ref.argmax(1))
        # the hard weights are the reference's ARITHMETIC, not exact one-hots: `y_hard - y.detach() + y`
        # (...globalgate.py:27-28) is (1 - y) + y at the arg-max and (0 - y) + y elsewhere, which `end_weight` counts with `== 1`
        # (...globalgate.py:241): the same bit pattern from the kernel's own soft weights y
        y, _, _ = ops.gate_head(pooled.cuda(), fc.cuda(), tab.cuda(), temp, False)
        onehot = torch.zeros_like(y).scatter_(1, y.argmax(1, keepdim=True), 1.0)
        assert torch.equal(w.detach(), (onehot - y) + y)
        assert int((w.detach() == 1).sum()) == int(((onehot - y) + y == 1).sum())
    ((w * gw.cuda()).sum() + (wc * gc.cuda()).sum() + l * gl.cuda()).backward()
    assert rel(pg.grad, pr.grad) < 2e-4
    assert rel(fg.grad, fr.grad) < 2e-4


def test_gate_from_weight(ops):
    from oracle import dynmm_oracle as O
    w = torch.zeros(4, 5)
    w[range(4), [1, 4, 0, 3]] = 1
    tab = torch.tensor(O.DEPTH_ENC_FLOP_R34)
    ww, wc, l = ops.gate_from_weight(w.cuda(), tab.cuda())
    assert torch.equal(ww.cpu(), w)
    assert torch.equal(wc.cpu(), torch.tensor([[0., 1, 1, 1], [0, 0, 0, 0], [1, 1, 1, 1], [0, 0, 0, 1]]))
    assert abs(l.item() - (w.mean(0) * tab).mean().item()) < 1e-6


def test_cross_entropy_2d(ops, golden_dir):
    import os
    from oracle import dynmm_oracle as O
    g = np.load(os.path.join(golden_dir, 'ops.npz'))
    cw = torch.from_numpy(g['ce/weight'])
    for i in range(2):
        x, t = torch.from_numpy(g[f'ce/x{i}']), torch.from_numpy(g[f'ce/t{i}'])
        xr = x.clone().requires_grad_(True)
        l_ref = O.cross_entropy_2d([xr], [t], cw)[0]
        l_ref.backward()
        xg = x.cuda().requires_grad_(True)
        l = ops.cross_entropy_2d(xg, t.cuda(), cw.cuda())
        assert abs(l.item() - float(g[f'ce/loss{i}'])) < 1e-5      # the reference's own value
        l.backward()
        assert rel(xg.grad, xr.grad) < 1e-4


def test_rejects_cpu_tensors(ops):
    from dynmm_amd.lib import DynmmHipError
    with pytest.raises(DynmmHipError):
        ops.conv2d(torch.randn(1, 16, 8, 8), torch.randn(16, 16, 3, 3), None, 1, 1)


@pytest.mark.parametrize('label_hw', [(24, 32), (37, 50), (12, 16)])
def test_eval_confusion(ops, label_hw):
    """eval.py:117-141 fused: bilinear resize to the label size, argmax, void mask, confusion matrix."""
    x = rnd(3, 40, 24, 32, seed=1)
    g = torch.Generator().manual_seed(5)
    label = torch.randint(0, 41, (3, *label_hw), generator=g)
    pred = F.interpolate(x, label_hw, mode='bilinear', align_corners=False).argmax(1)
    mask = label > 0
    ref = torch.bincount(40 * (label[mask] - 1) + pred[mask], minlength=1600).reshape(40, 40)
    cm = torch.zeros(40, 40, dtype=torch.int64, device='cuda')
    ops.eval_confusion(x.cuda(), label.cuda(), cm)
    ops.eval_confusion(x.cuda(), label.cuda(), cm)       # accumulates
    diff = (cm.cpu() - 2 * ref).abs().sum().item()
    assert diff <= 4, diff      # a handful of fp32 arg-max ties at most
    assert cm.sum().item() == 2 * mask.sum().item()


WINO_CASES = [
    # (N, Ci, H, W, Co, kernel): stride 1, same padding, Ci, Co % 64 == 0, W % 4 == 0
    (3, 128, 15, 20, 128, (1, 3)),       # odd H, 900 pixels: ragged last pair tile
    (5, 64, 17, 20, 64, (3, 1)),         # vertical pairs with an odd number of rows (the last pair has one live row)
    (2, 128, 9, 16, 256, (3, 1)),        # two output-channel tiles
    (2, 192, 8, 24, 64, (1, 3)),         # 64-channel output tile (128 pairs per workgroup), 24 chunks
    (2, 64, 12, 16, 128, (3, 3)),        # 3x3: vertical taps looped, zero slot for rows outside the image
    (4, 128, 30, 40, 128, (3, 3)),
    (7, 64, 6, 12, 64, (3, 1)),
    (2, 256, 15, 20, 64, (1, 3)),
    (3, 128, 24, 32, 40, (3, 3)),        # conv_out's shape class: 40 output rows in a 64-row tile (zero-padded filter operand)
    (3, 64, 10, 12, 72, (3, 1)),         # forward with a row tail in the second tile (72 = 64 + 8); its input gradient reduces over 72
    (3, 128, 7, 16, 24, (1, 3)),         # the smallest row count the forward takes
]


@pytest.mark.parametrize('mode', ['dgrad', 'all', 'dgrad43'])
@pytest.mark.parametrize('case', WINO_CASES)
def test_conv2d_winograd(ops, mode, case):
    """csrc/conv_wino.hip (1-D Winograd F(2,3), fp32 MFMA; 1x3 / 3x1), conv_wino43.hip (F(4,3), 1x3 input gradients) and
    conv_wino2d.hip (2-D F(2x2,3x3); 3x3) through ops.conv2d: the forward (mode 'all') and the input gradient with both epilogue
    operands (ReLU mask of the producer, residual-branch gradient) against a float64 PyTorch reference at
    the SAME bars as the direct kernels (TOL / GTOL), weight and bias gradients unchanged; and the kernel really ran."""
    import ctypes as C
    N, Ci, H, W, Co, k = case
    p = (k[0] // 2, k[1] // 2)
    x, w = rnd(N, Ci, H, W, seed=1).relu_(), rnd(Co, Ci, *k, seed=2, scale=(Ci * k[0] * k[1]) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    xr, wr, br = [t.double().requires_grad_(True) for t in (x, w, b)]
    y_ref = F.conv2d(xr, wr, br, 1, p)
    gy = rnd(*y_ref.shape, seed=4)
    dres = rnd(N, Ci, H, W, seed=5)
    y_ref.backward(gy.double())
    dx_ref = xr.grad * (x > 0) + dres.double()
    old, old_d = ops.WINO, ops.WINO_DGRAD
    # 'dgrad43' = the product default WINO_DGRAD = '43h': the input gradient of the 1x3 filters by F(4,3) (csrc/conv_wino43.hip:
    # half of the direct matrix work, 1e-6 .. 4e-6 from fp64 — held to the same GTOL), the 3x1 ones by F(2,3); the other two modes
    # pin F(2,3) for both; 3x3 filters take the 2-D kernel in every mode
    f43 = mode == 'dgrad43'
    ops.WINO, ops.WINO_DGRAD = ('dgrad' if f43 else mode), ('43h' if f43 else '23')
    mode = 'dgrad' if f43 else mode
    calls = []
    lib = ops._lib()
    try:
        xg, wg, bg = [t.cuda().requires_grad_(True) for t in (x, w, b)]
        g = ops._geom(xg, None, wg, (1, 1), p)
        k33 = k == (3, 3)
        assert lib.dynmm_conv2d_wino_supported(C.byref(g), 1) == (0 if k33 else 1)       # each filter shape has ONE Winograd home
        assert bool(lib.dynmm_conv2d_wino2d_supported(C.byref(g), 1)) == k33
        for sel in (ops._wino2d, ops._wino):
            assert sel(g, True) == (k33 == (sel is ops._wino2d)) and sel(g, False) == (sel(g, True) and mode == 'all')
        link = ops.GradLink()
        ops.PROFILE = calls
        y = ops.conv2d(xg, wg, bg, 1, p, None, mask_input=True, link=link)
        link.dres = dres.cuda()
        y.backward(gy.cuda())
    finally:
        ops.WINO, ops.WINO_DGRAD = old, old_d
        ops.PROFILE = None
    torch.cuda.synchronize()
    names = [c[0] for c in calls]
    if k == (3, 3):        # 3x3 filters: the 2-D F(2x2,3x3) kernel (csrc/conv_wino2d.hip) in every mode
        assert any(n.startswith('conv_wino2d_dgrad') for n in names), names
        assert any(n.startswith('conv_wino2d_fwd') for n in names) == (mode == 'all'), names
    else:
        assert any(n.startswith('conv_wino43_dgrad' if (f43 and k[1] == 3) else 'conv_wino_dgrad') for n in names), names
        assert any(n.startswith('conv_wino_fwd') for n in names) == (mode == 'all'), names
    assert rel(y, y_ref) < TOL
    assert rel(xg.grad, dx_ref) < GTOL
    assert rel(wg.grad, wr.grad) < GTOL and rel(bg.grad, br.grad) < GTOL


@pytest.mark.parametrize('case', [(3, 128, 15, 20, 128, (1, 3)), (2, 64, 12, 16, 128, (3, 3)), (5, 64, 9, 12, 64, (1, 3)),
                                  (2, 192, 8, 24, 256, (1, 3)), (3, 128, 7, 20, 128, (1, 3)), (8, 64, 120, 160, 64, (1, 3))])
def test_conv_epilogue_batchnorm_statistics(ops, case):
    """conv2d(bn_stats=True) -> batch_norm_act (training): the BatchNorm's batch statistics come out of the convolution's epilogue
    (csrc/conv_wino.hip STATS) instead of a bn_stats pass.  Same output, running statistics, step counter and gradients as the
    two-pass path; against float64 the statistics themselves to 1e-6."""
    import torch.nn as nn
    N, Ci, H, W, Co, k = case
    p = (k[0] // 2, k[1] // 2)
    torch.manual_seed(N * 100 + Co)
    conv = nn.Conv2d(Ci, Co, k, padding=p).cuda()
    bn = nn.BatchNorm2d(Co).cuda().train()
    with torch.no_grad():
        conv.bias.mul_(5.0)
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    x = rnd(N, Ci, H, W, seed=1).cuda()
    gz = rnd(N, Co, H, W, seed=2).cuda()
    y_ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), 1, p)
    old = ops.CONV_BN_STATS

    def run(flag):
        ops.CONV_BN_STATS = flag
        bn.reset_running_stats()
        for q in list(conv.parameters()) + list(bn.parameters()):
            q.grad = None
        xi = x.clone().requires_grad_(True)
        y = ops.conv2d(xi, conv.weight, conv.bias, 1, p, None, bn_stats=True)
        assert hasattr(y, '_bn_sums') == flag
        if flag:
            M = N * H * W
            tot = y._bn_sums.view(-1, 2, Co).sum(0)
            assert rel(tot[0] / M, y_ref.mean((0, 2, 3))) < 1e-6
            assert rel(tot[1] / M, (y_ref * y_ref).mean((0, 2, 3))) < 1e-6
        z = ops.batch_norm_act(y, bn, 'relu')
        z.backward(gz)
        torch.cuda.synchronize()
        return (z.detach(), xi.grad, bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked),
                conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
    try:
        a, b = run(False), run(True)
    finally:
        ops.CONV_BN_STATS = old
    assert a[4] == b[4] == 1
    assert rel(b[0], a[0]) < 1e-5 and rel(b[2], a[2]) < 1e-6 and rel(b[3], a[3]) < 1e-6
    for i in (1, 5, 6, 7):
        assert rel(b[i], a[i]) < 2e-4, i


@pytest.mark.parametrize('case', [(3, 3, 40, 56), (2, 1, 40, 56), (2, 3, 37, 131), (3, 1, 30, 70), (4, 3, 96, 128)])
def test_stem_conv_epilogue_batchnorm_statistics(ops, case):
    """The 7x7 / stride-2 stem convolution with bn_stats=True (csrc/conv_small.hip conv_stem_fwd_kernel<CI, STATS>): the
    statistics of bn1 come out of the persistent kernel's epilogue.  Through the C ABI against float64 (sums to 1e-6, the output
    bit-identical to the plain launch), then the two consumers — batch_norm_act and stem_bn_fuse_pool — against their two-pass
    forms, with a spy that no dynmm_bn_stats launch is left.  Shapes: whole tiles, ragged rows / columns (Ho % 4, Wo % 64,
    Wo % 4 != 0), more tiles than resident workgroups."""
    import ctypes as C
    import torch.nn as nn
    from dynmm_amd import lib as L
    lib = L.load()
    N, Ci, H, W = case
    torch.manual_seed(7 * N + Ci)
    conv = nn.Conv2d(Ci, 64, 7, stride=2, padding=3, bias=(Ci == 1)).cuda()
    x = (rnd(N, Ci, H, W, seed=1) + 0.25).cuda()
    y_ref = F.conv2d(x.double(), conv.weight.double(), None if conv.bias is None else conv.bias.double(), 2, 3)
    g = ops._geom(x, None, conv.weight, (2, 2), (3, 3))
    assert lib.dynmm_conv2d_stem_fwd_stats_supported(C.byref(g))
    st = torch.cuda.current_stream().cuda_stream
    wp = torch.empty(lib.dynmm_packed_weight_floats(64, Ci, 7, 7, 0), device='cuda')
    L.check(lib.dynmm_pack_weight(ops._p(conv.weight.detach()), ops._p(wp), None, 64, Ci, 7, 7, st), 'pack')
    y0, y1 = torch.empty(N, 64, g.Ho, g.Wo, device='cuda'), torch.empty(N, 64, g.Ho, g.Wo, device='cuda')
    sums = torch.zeros(2, 64, device='cuda', dtype=torch.float64)
    bias = None if conv.bias is None else conv.bias.detach()
    L.check(lib.dynmm_conv2d_fwd(ops._p(x), None, ops._p(wp), None, ops._p(bias), None, ops._p(y0), C.byref(g), L.ACT_NONE, st), 'fwd')
    L.check(lib.dynmm_conv2d_stem_fwd_stats(ops._p(x), ops._p(wp), ops._p(bias), ops._p(y1), ops._p(sums), C.byref(g), st), 'stats')
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert rel(y1, y_ref) < TOL
    M = N * g.Ho * g.Wo
    assert rel(sums[0] / M, y_ref.mean((0, 2, 3))) < 1e-6
    assert rel(sums[1] / M, (y_ref * y_ref).mean((0, 2, 3))) < 1e-6

    # consumer 1: batch_norm_act (the unfused stem of ESANet / SkipESANet, resnet.py:229-231)
    bn = nn.BatchNorm2d(64).cuda().train()
    old = ops.CONV_BN_STATS

    def run(flag):
        ops.CONV_BN_STATS = flag
        bn.reset_running_stats()
        with _CallSpy(lib, 'dynmm_bn_stats', 'dynmm_conv2d_stem_fwd_stats') as spy:
            y = ops.conv2d(x, conv.weight, conv.bias, 2, 3, None, bn_stats=True)
            assert hasattr(y, '_bn_sums') == flag
            z = ops.batch_norm_act(y, bn, 'relu')
        assert spy.count == ({'dynmm_bn_stats': 0, 'dynmm_conv2d_stem_fwd_stats': 1} if flag else
                             {'dynmm_bn_stats': 1, 'dynmm_conv2d_stem_fwd_stats': 0}), spy.count
        return z.detach(), bn.running_mean.clone(), bn.running_var.clone()
    try:
        a, b = run(False), run(True)
    finally:
        ops.CONV_BN_STATS = old
    assert rel(b[0], a[0]) < 1e-5 and rel(b[1], a[1]) < 1e-6 and rel(b[2], a[2]) < 1e-6

    # consumer 2: the fused stem tail of SkipGateESANet (ops.stem_bn_fuse_pool), both stems from their epilogues
    if g.Ho % 2 == 0 and ops.stem_bn_fuse_supported(g.Ho, g.Wo, bn, bn):
        convd = nn.Conv2d(1, 64, 7, stride=2, padding=3, bias=False).cuda()
        xd = (rnd(N, 1, H, W, seed=3) - 0.1).cuda()
        bns = [nn.BatchNorm2d(64).cuda().train() for _ in range(4)]

        def tail(flag, br, bd):
            with _CallSpy(lib, 'dynmm_bn_stats') as spy:
                yr = ops.conv2d(x, conv.weight, conv.bias, 2, 3, None, bn_stats=flag)
                yd = ops.conv2d(xd, convd.weight, None, 2, 3, None, bn_stats=flag)
                o, dp = ops.stem_bn_fuse_pool(yr, br, yd, bd, None)
            assert spy.count['dynmm_bn_stats'] == (0 if flag else 2)
            return o.detach(), dp.detach(), br.running_var.clone(), bd.running_var.clone()
        a, b = tail(False, bns[0], bns[1]), tail(True, bns[2], bns[3])
        for u, v in zip(a, b):
            assert rel(v, u) < 1e-5


class _CallSpy:
    """Count the calls of C-ABI entry points (the ctypes function objects of the loaded library are replaced by counting
    wrappers for the duration of the block)."""

    def __init__(self, lib, *names):
        self.lib, self.names, self.count = lib, names, {n: 0 for n in names}

    def __enter__(self):
        self.orig = {n: getattr(self.lib, n) for n in self.names}
        for n in self.names:
            def wrapper(*a, _n=n):
                self.count[_n] += 1
                return self.orig[_n](*a)
            setattr(self.lib, n, wrapper)
        return self

    def __exit__(self, *a):
        for n in self.names:
            setattr(self.lib, n, self.orig[n])


@pytest.mark.parametrize('case,fits', [((3, 128, 15, 20, 128), True),        # odd H: the last row pair has one live row
                                       ((2, 64, 24, 32, 128), True),
                                       ((7, 64, 6, 12, 64), True),           # 252 pairs: a ragged last pixel tile
                                       ((8, 64, 120, 160, 64), True),        # 1200 pixel tiles: the C = 64 stage at batch 8, 2 slabs
                                       ((17, 64, 120, 160, 64), True),       # 2550 tiles: 4 slabs (<= 600 tiles add to one address)
                                       ((3, 96, 15, 20, 128), False)])       # 96 rows: not a multiple of the 64-row tile -> unfused path
def test_batchnorm_backward_reductions_from_the_consumer_dgrad(ops, case, fits):
    """relu(BN(c)) -> conv3x1 (resnet.py:127-135: bn1 -> conv3x1_2): with ops.BNLink the convolution's input-gradient launch
    (dynmm_conv2d_wino_dgrad_bnred, csrc/conv_wino.hip BNRED) masks by [BN(c) > 0] from c itself and leaves the BatchNorm
    backward's two reductions — fp64 atomics, one per channel, statistic and tile — so bn_bwd_reduce is not launched.  Checked
    through the C ABI against float64 autograd: every gradient to GTOL, the two reductions (= dbeta, dgamma) to 1e-6 of their
    absolute sums, fused == unfused; a spy asserts which path ran (the link consumed on the fused one), with one and with several
    slabs of sums (dynmm_conv2d_wino_dgrad_bnred_slots), and for a geometry the kernel does not take."""
    import torch.nn as nn
    N, Ci, H, W, Co = case
    torch.manual_seed(N * 1000 + Ci + Co)
    conv = nn.Conv2d(Ci, Co, (3, 1), padding=(1, 0)).cuda()
    bn = nn.BatchNorm2d(Ci, eps=1e-3).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    c = rnd(N, Ci, H, W, seed=1)
    gam, bet = bn.weight.detach().cpu().double(), bn.bias.detach().cpu().double()
    for _ in range(4):                           # no BatchNorm output within 1e-4 of the ReLU's corner (decisions are not under test)
        z0 = F.batch_norm(c.double(), None, None, gam, bet, True, 0.0, 1e-3)
        near = z0.abs() < 1e-3
        if not near.any():
            break
        c = c + near.float() * 0.02 * torch.where(z0 >= 0, 1.0, -1.0).float() / gam.float().view(1, -1, 1, 1)
    assert F.batch_norm(c.double(), None, None, gam, bet, True, 0.0, 1e-3).abs().min() > 1e-4
    gy = rnd(N, Co, H, W, seed=2)
    # float64 truth
    cr = c.double().requires_grad_(True)
    gr, br, wr, bcr = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True), \
        conv.weight.detach().cpu().double().requires_grad_(True), conv.bias.detach().cpu().double().requires_grad_(True)
    zr = F.relu(F.batch_norm(cr, None, None, gr, br, True, 0.0, 1e-3))
    zr.retain_grad()
    F.conv2d(zr, wr, bcr, 1, (1, 0)).backward(gy.double())
    g_masked = zr.grad * (zr.detach() > 0)
    mu, var = c.double().mean((0, 2, 3), keepdim=True), c.double().var((0, 2, 3), unbiased=False, keepdim=True)
    xhat = (c.double() - mu) / (var + 1e-3).sqrt()
    abs1, abs2 = g_masked.abs().sum((0, 2, 3)), (g_masked * xhat).abs().sum((0, 2, 3))
    lib = ops._lib()
    old = ops.BN_BWD_FUSE

    def run(fuse):
        ops.BN_BWD_FUSE = fuse
        bn.reset_running_stats()
        for q in list(conv.parameters()) + list(bn.parameters()):
            q.grad = None
        ci = c.clone().cuda().requires_grad_(True)
        bnl = ops.BNLink()
        with _CallSpy(lib, 'dynmm_conv2d_wino_dgrad_bnred', 'dynmm_bn_bwd_reduce') as spy:
            z = ops.batch_norm_act(ci, bn, 'relu', bwd_link=bnl)
            y = ops.conv2d(z, conv.weight, conv.bias, 1, (1, 0), None, bn_link=bnl)
            y.backward(gy.cuda())
            torch.cuda.synchronize()
        assert bnl.sums is None and bnl.x is None            # consumed (or never filled) — nothing dangling on the link
        return (ci.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()), dict(spy.count)
    try:
        (a, ca), (b, cb) = run(False), run(True)
    finally:
        ops.BN_BWD_FUSE = old
    assert ca == {'dynmm_conv2d_wino_dgrad_bnred': 0, 'dynmm_bn_bwd_reduce': 1}, ca
    assert cb == ({'dynmm_conv2d_wino_dgrad_bnred': 1, 'dynmm_bn_bwd_reduce': 0} if fits else ca), (cb, fits)
    for got in (a, b):
        for t, ref in zip(got, (cr.grad, gr.grad, br.grad, wr.grad, bcr.grad)):
            assert rel(t, ref) < GTOL
        assert ((got[2].cpu().double() - br.grad).abs() / abs1).max() < 1e-6         # sum g.[z > 0]
        assert ((got[1].cpu().double() - gr.grad).abs() / abs2).max() < 1e-6         # sum g.[z > 0].xhat
    for t, u in zip(a, b):
        assert rel(u, t) < 2e-5


@pytest.mark.parametrize('case', [(3, 64, 16, 24, 128, (3, 1), (2, 1), (1, 0)), (3, 128, 16, 24, 128, (1, 3), (1, 2), (0, 1)),
                                  (2, 128, 30, 40, 256, (3, 1), (2, 1), (1, 0)), (4, 64, 9, 16, 40, (1, 3), (1, 2), (0, 1))])
def test_conv2d_stride2_input_gradient_on_the_pair_kernel(ops, case):
    """The stride-2 three-tap convolutions that open stages 2-4 (resnet.py:104-107): their input gradient in polyphase form on
    csrc/conv_wino.hip's pair kernel (dx[2j] = W1^T dy[j], dx[2j+1] = W2^T dy[j] + W0^T dy[j+1]), with ReLU mask and residual
    gradient, vs float64; forward and weight gradient (unchanged kernels) ride along."""
    N, Ci, H, W, Co, k, st_, p = case
    x, w = rnd(N, Ci, H, W, seed=1).relu_(), rnd(Co, Ci, *k, seed=2, scale=(Ci * 3) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    xr, wr, br = [t.double().requires_grad_(True) for t in (x, w, b)]
    y_ref = F.conv2d(xr, wr, br, st_, p)
    gy = rnd(*y_ref.shape, seed=4)
    dres = rnd(N, Ci, H, W, seed=5)
    y_ref.backward(gy.double())
    dx_ref = xr.grad * (x > 0) + dres.double()
    calls = []
    try:
        xg, wg, bg = [t.cuda().requires_grad_(True) for t in (x, w, b)]
        link = ops.GradLink()
        ops.PROFILE = calls
        y = ops.conv2d(xg, wg, bg, st_, p, None, mask_input=True, link=link)
        link.dres = dres.cuda()
        y.backward(gy.cuda())
    finally:
        ops.PROFILE = None
    torch.cuda.synchronize()
    assert any(n.startswith('conv_wino_dgrad') and n.endswith('s2>') for n in (c[0] for c in calls)), [c[0] for c in calls]
    assert rel(y, y_ref) < TOL and rel(xg.grad, dx_ref) < GTOL
    assert rel(wg.grad, wr.grad) < GTOL and rel(bg.grad, br.grad) < GTOL


def test_winograd_operands_from_the_step_pack(ops):
    """ops.PackedWeights: the filter transforms written by the ONE dynmm_wino_pack_multi launch of a step equal the per-conv
    packs bit for bit, for both directions and all three filter shapes (1x3: F(2,3) + F(4,3); 3x1: F(2,3); 3x3: 2-D); geometries
    outside the kernels' rules are refused."""
    import ctypes as C
    from dynmm_amd import lib as L
    lib = ops._lib()
    st = torch.cuda.current_stream().cuda_stream
    ws = [torch.nn.Parameter(rnd(*s, seed=i).cuda()) for i, s in enumerate([(128, 64, 1, 3), (64, 64, 3, 1), (128, 128, 3, 3)])]
    pw = ops.PackedWeights()
    for w in ws:
        Co, Ci, KH, KW = w.shape
        g = L.ConvGeom(2, Ci, 16, 16, Co, 16, 16, KH, KW, 1, 1, KH // 2, KW // 2, Ci)
        k33 = KH * KW == 9
        pw.register(w, g, True, False, not k33, not k33, KW == 3 and not k33, k33, k33)
    pw.pack()
    torch.cuda.synchronize()
    for w in ws:
        Co, Ci, KH, KW = w.shape
        k33 = KH * KW == 9
        h13 = KW == 3 and not k33
        wp, wpd, utf, utd, utd43, ut2f, ut2d = pw.lookup(w, True, False, not k33, not k33, h13, k33, k33)
        assert wpd is None and pw.lookup(w, True, True) is None          # the direct input-gradient layout was not requested
        assert (ut2f is not None) == k33 and (k33 or pw.lookup(w, True, False, True, True, h13, True) is None)
        assert (utf is None) == k33 and (not k33 or pw.lookup(w, True, False, True) is None)
        assert lib.dynmm_wino_packed_floats(Co, Ci, KH, KW) == (0 if k33 else 4 * max(Ci * ((Co + 63) // 64 * 64), Co * ((Ci + 63) // 64 * 64)))
        assert lib.dynmm_wino43_packed_floats(Co, Ci, KH, KW) == (6 * Co * Ci if h13 else 0)
        for dgrad, got in ((0, ut2f), (1, ut2d)) if k33 else ():         # the 2-D F(2x2,3x3) operands (csrc/conv_wino2d.hip)
            ref = torch.zeros(lib.dynmm_wino2d_packed_floats(Co, Ci), device='cuda')
            L.check(lib.dynmm_wino2d_pack(w.data_ptr(), ref.data_ptr(), None, Co, Ci, dgrad, st), 'wino2d_pack')
            n_live = (Co if dgrad else Ci) * 4 * (((Ci if dgrad else Co) + 63) // 64 * 64) * 4
            assert torch.equal(got[:n_live], ref[:n_live]), (tuple(w.shape), dgrad)
        if h13:
            ref43 = torch.empty(lib.dynmm_wino43_packed_floats(Co, Ci, KH, KW), device='cuda')
            L.check(lib.dynmm_wino43_pack(w.data_ptr(), ref43.data_ptr(), Co, Ci, KH, KW, st), 'wino43_pack')
            assert torch.equal(utd43, ref43), tuple(w.shape)
        else:
            assert utd43 is None
        for dgrad, got in ((0, utf), (1, utd)) if not k33 else ():
            ref = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, KH, KW), device='cuda')
            L.check(lib.dynmm_wino_pack(w.data_ptr(), ref.data_ptr(), None, Co, Ci, KH, KW, dgrad, st), 'wino_pack')
            assert torch.equal(got, ref), (tuple(w.shape), dgrad)
    strided = L.ConvGeom(2, 64, 16, 16, 64, 8, 16, 3, 1, 2, 1, 1, 0, 64)          # stride 2: input gradient only (polyphase form)
    assert lib.dynmm_conv2d_wino_supported(C.byref(strided), 0) == 0 and lib.dynmm_conv2d_wino_supported(C.byref(strided), 1) == 2
    for bad in (L.ConvGeom(2, 64, 16, 16, 64, 8, 8, 3, 3, 2, 2, 1, 1, 64),        # 3x3 stride 2
                L.ConvGeom(2, 64, 16, 18, 64, 16, 18, 1, 3, 1, 1, 0, 1, 64),     # W % 4 != 0
                L.ConvGeom(2, 20, 16, 16, 60, 16, 16, 1, 3, 1, 1, 0, 1, 20),     # neither channel count fits a 64-row tile
                L.ConvGeom(2, 64, 16, 16, 64, 16, 16, 1, 1, 1, 1, 0, 0, 64)):    # 1x1
        assert lib.dynmm_conv2d_wino_supported(C.byref(bad), 0) == 0 and lib.dynmm_conv2d_wino_supported(C.byref(bad), 1) == 0


def test_fused_eval_cache_follows_parameter_changes(ops):
    """conv2d_fused_eval caches the packed weights / folded BN per layer; every way the tensors can change
    (in-place torch ops, a training forward that rewrites the running statistics inside our kernel, a new
    storage after .to()/load_state_dict) must invalidate it."""
    conv = torch.nn.Conv2d(64, 64, (3, 1), padding=(1, 0)).cuda()
    bn = torch.nn.BatchNorm2d(64, eps=1e-3).cuda()
    x = rnd(2, 64, 8, 8, seed=1).cuda()

    def hip():
        bn.eval()
        with torch.no_grad():
            return ops.conv2d_fused_eval(x, conv.weight, conv.bias, bn, 'relu', None, 1, (1, 0))

    def ref():
        bn.eval()
        with torch.no_grad():
            return F.relu(bn(F.conv2d(x, conv.weight, conv.bias, 1, (1, 0))))
    assert rel(hip(), ref().cpu()) < TOL
    assert rel(hip(), ref().cpu()) < TOL                      # cached path
    with torch.no_grad():
        conv.weight.mul_(1.5)                                  # in-place: version counter
    assert rel(hip(), ref().cpu()) < TOL
    bn.train()
    ops.batch_norm_act(rnd(4, 64, 8, 8, seed=2).cuda() * 3 + 1, bn, 'relu')   # running stats rewritten in-kernel
    assert float(bn.running_mean.abs().max()) > 0.05
    assert rel(hip(), ref().cpu()) < TOL
    sd = {k: v.clone() + 0.25 for k, v in bn.state_dict().items() if v.dtype.is_floating_point}
    bn.load_state_dict(sd, strict=False)
    assert rel(hip(), ref().cpu()) < TOL


def test_conv_gradients_are_run_to_run_reproducible(ops):
    """Weight / bias gradients come from slab partials reduced in a fixed order (no float atomics) and the
    dgrad has no split-K at all: repeated launches must be bit-identical."""
    x = rnd(4, 128, 30, 40, seed=1).cuda()
    w = (rnd(128, 128, 3, 1, seed=2) * 0.05).cuda()
    b = (rnd(128, seed=3) * 0.1).cuda()
    gy = rnd(4, 128, 30, 40, seed=4).cuda()
    res = []
    for _ in range(3):
        xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv2d(xg, wg, bg, 1, (1, 0), 'relu')
        y.backward(gy)
        res.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone()))
    for r in res[1:]:
        for a, bb in zip(res[0], r):
            assert torch.equal(a, bb)


@pytest.mark.parametrize('shape', [(2, 40, 12, 16), (3, 40, 7, 9), (1, 5, 30, 70)])
def test_fused_upsample_cross_entropy_tail(ops, shape):
    """csrc/tail.hip: the decoder's last learned 2x up-sampling (model.py:404-410) fused with the weighted CE
    (src/utils.py:34-50), logits never materialised — loss and the gradients of the up-sampling's input, weight and
    bias against an fp64 torch restatement (nearest x2 -> depthwise 3x3 zero-pad -> CE), with void pixels, a second
    (ordinary) scale so the per-scale seeds differ, and twice for bit-reproducibility."""
    import torch.nn as nn
    N, C, H, W = shape
    g = torch.Generator().manual_seed(7 + H)
    x = rnd(N, C, H, W, seed=1, scale=2.0)
    conv = nn.Conv2d(C, C, 3, padding=1, groups=C)
    with torch.no_grad():
        conv.weight.copy_(rnd(C, 1, 3, 3, seed=2, scale=0.4))
        conv.bias.copy_(rnd(C, seed=3, scale=0.3))
    t0 = torch.randint(0, C + 1, (N, 2 * H, 2 * W), generator=g).to(torch.uint8)
    t0[0, :3] = 0                                                  # a run of void pixels
    x1 = rnd(N, C, 4, 5, seed=4)
    t1 = torch.randint(0, C + 1, (N, 4, 5), generator=g).to(torch.uint8)
    cw = torch.rand(C, generator=g) + 0.5

    # fp64 restatement
    xd = x.double().requires_grad_(True)
    wd = conv.weight.detach().double().requires_grad_(True)
    bd = conv.bias.detach().double().requires_grad_(True)
    x1d = x1.double().requires_grad_(True)
    logits = F.conv2d(F.interpolate(xd, scale_factor=2, mode='nearest'), wd, bd, padding=1, groups=C)

    def ce(lg, t):
        tt = t.long() - 1
        m = tt >= 0
        l = F.cross_entropy(lg, tt.clamp_min(0), weight=cw.double(), reduction='none')
        return (l * m).sum() / cw.double()[tt.clamp_min(0)][m].sum()
    l0, l1 = ce(logits, t0), ce(x1d, t1)
    (l0 + l1).backward()

    conv = conv.cuda()
    runs = []
    for _ in range(2):
        conv.zero_grad()
        xg = x.cuda().requires_grad_(True)
        x1g = x1.cuda().requires_grad_(True)
        dl = ops.DeferredLogits(xg, conv)
        assert tuple(dl.shape) == (N, C, 2 * H, 2 * W)
        assert rel(dl.detach(), logits) < TOL
        res = ops.multi_scale_loss_backward([dl, x1g], [t0.cuda(), t1.cuda()], cw.cuda())
        runs.append((res['losses'].clone(), xg.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone(), x1g.grad.clone()))
    losses, dx, dw, db, dx1 = runs[0]
    assert abs(losses[0].item() - l0.item()) < 2e-5 * max(1.0, abs(l0.item()))
    assert abs(losses[1].item() - l1.item()) < 2e-5 * max(1.0, abs(l1.item()))
    assert abs(res['total'].item() - (l0 + l1).item()) < 4e-5 * max(1.0, (l0 + l1).item())
    assert rel(dx, xd.grad) < GTOL
    assert rel(dw, wd.grad) < GTOL
    assert rel(db, bd.grad) < GTOL
    assert rel(dx1, x1d.grad) < GTOL
    # only the fp64 loss accumulators are atomic (their double sums round identically in any order up to 1 ulp of
    # fp64, invisible in the fp32 seed): every gradient is bit-reproducible
    for a, b in zip(runs[0][1:], runs[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('use_se', [True, False])
@pytest.mark.parametrize('shape', [(2, 64, 24, 32), (3, 32, 6, 8)])
def test_se_fuse_pool_equals_unfused_ops(ops, use_se, shape):
    """axpby_pool_*: stem fusion + both max-pools fused (the full-resolution fused map never written) against the
    composition se_fuse_blend -> max_pool, max_pool it replaces: outputs bit-identical (same expressions, same tie
    rule on post-ReLU zeros), gradients to summation order."""
    N, C, H, W = shape
    assert ops.se_fuse_pool_supported(torch.empty(shape))
    rgb, depth = F.relu(rnd(*shape, seed=1)), F.relu(rnd(*shape, seed=2))
    prm = None
    if use_se:
        prm = []
        for k in range(2):
            prm += [rnd(C // 16, C, 1, 1, seed=3 + k, scale=0.2), rnd(C // 16, seed=5 + k, scale=0.1),
                    rnd(C, C // 16, 1, 1, seed=7 + k, scale=0.2), rnd(C, seed=9 + k, scale=0.1)]
    g1, g2 = rnd(N, C, H // 2, W // 2, seed=11), rnd(N, C, H // 2, W // 2, seed=12)

    def run(fused):
        r, d = rgb.cuda().requires_grad_(True), depth.cuda().requires_grad_(True)
        p = [t.cuda().requires_grad_(True) for t in prm] if prm else None
        if fused:
            o, dp = ops.se_fuse_pool(r, d, p)
        else:
            o = ops.max_pool_3x3_s2(ops.se_fuse_blend(r, d, p))
            dp = ops.max_pool_3x3_s2(d)
        torch.autograd.backward([o, dp], [g1.cuda(), g2.cuda()])
        return o.detach(), dp.detach(), r.grad, d.grad, [t.grad for t in p] if p else []
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert rel(a[2], b[2]) < 1e-5 and rel(a[3], b[3]) < 1e-5
    for x, y in zip(a[4], b[4]):
        assert rel(x, y) < 1e-4
    a2 = run(True)                                               # bit-reproducible
    for x, y in zip(a[2:4], a2[2:4]):
        assert torch.equal(x, y)


@pytest.mark.parametrize('use_se', [True, False])
def test_stem_bn_fuse_pool_equals_unfused_ops(ops, use_se):
    """_StemBNFusePool: stem BatchNorm + ReLU applied on load by the squeeze / blend + pooling kernels (the normalised
    tensors never written) against batch_norm_act x 2 -> se_fuse_blend -> max_pool x 2: outputs, running statistics
    and step counters bit-identical, gradients to summation order."""
    import torch.nn as nn
    N, C, H, W = 3, 32, 12, 16
    xr0, xd0 = rnd(N, C, H, W, seed=1, scale=2.0), rnd(N, C, H, W, seed=2, scale=1.5) + 0.3
    prm = None
    if use_se:
        prm = []
        for k in range(2):
            prm += [rnd(C // 16, C, 1, 1, seed=3 + k, scale=0.2), rnd(C // 16, seed=5 + k, scale=0.1),
                    rnd(C, C // 16, 1, 1, seed=7 + k, scale=0.2), rnd(C, seed=9 + k, scale=0.1)]
    g1, g2 = rnd(N, C, H // 2, W // 2, seed=11), rnd(N, C, H // 2, W // 2, seed=12)

    def run(fused):
        bns = []
        for k in range(2):
            bn = nn.BatchNorm2d(C, eps=1e-5 if k == 0 else 1e-3).cuda().train()
            with torch.no_grad():
                bn.weight.copy_(rnd(C, seed=20 + k).abs() + 0.5)
                bn.bias.copy_(rnd(C, seed=22 + k, scale=0.3))
                bn.running_mean.copy_(rnd(C, seed=24 + k, scale=0.1))
            bns.append(bn)
        xr, xd = xr0.cuda().requires_grad_(True), xd0.cuda().requires_grad_(True)
        p = [t.cuda().requires_grad_(True) for t in prm] if prm else None
        if fused:
            assert ops.stem_bn_fuse_supported(H, W, *bns)
            if fused == 'deferred':      # each stem's BatchNorm backward runs in its own autograd node (ops.stem_bn_defer)
                xd_, xr_ = ops.stem_bn_defer(xd, bns[1]), ops.stem_bn_defer(xr, bns[0])
                assert hasattr(xr_, '_stem_slot') and hasattr(xd_, '_stem_slot')
                o, dp = ops.stem_bn_fuse_pool(xr_, bns[0], xd_, bns[1], p)
            elif fused == 'one deferred':    # only one stem wrapped: the fused node does both BatchNorm backwards itself
                o, dp = ops.stem_bn_fuse_pool(ops.stem_bn_defer(xr, bns[0]), bns[0], xd, bns[1], p)
            else:
                o, dp = ops.stem_bn_fuse_pool(xr, bns[0], xd, bns[1], p)
        else:
            yr, yd = ops.batch_norm_act(xr, bns[0], 'relu'), ops.batch_norm_act(xd, bns[1], 'relu')
            yd1, yd2 = ops.fan_out(yd, 2)
            o = ops.max_pool_3x3_s2(ops.se_fuse_blend(yr, yd1, p))
            dp = ops.max_pool_3x3_s2(yd2)
        torch.autograd.backward([o, dp], [g1.cuda(), g2.cuda()])
        bufs = [t.detach().clone() for bn in bns for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked)]
        grads = [xr.grad, xd.grad] + [t.grad for bn in bns for t in (bn.weight, bn.bias)] + ([t.grad for t in p] if p else [])
        return o.detach(), dp.detach(), bufs, grads
    a, b = run(True), run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        assert torch.equal(x, y)
    for i, (x, y) in enumerate(zip(a[3], b[3])):
        assert rel(x, y) < 2e-4, i
    c = run('deferred')                  # the same kernels in another order (fp64-atomic order in the reductions aside)
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    for x, y in zip(a[2], c[2]):
        assert torch.equal(x, y)
    for i, (x, y) in enumerate(zip(a[3], c[3])):
        assert rel(x, y) < 1e-6, i
    d = run('one deferred')
    assert torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])
    for i, (x, y) in enumerate(zip(a[3], d[3])):
        assert rel(x, y) < 1e-6, i


def test_weight_gradient_streams_have_least_priority(ops):
    """ops.stream_plan(): weight-gradient streams of the least priority the device offers, a depth-encoder stream of the normal one,
    the SAME objects on every call (process-wide singletons; created through the
    runtime, wrapped as torch streams), usable like any torch stream (event ordering against the current stream)."""
    import ctypes as C
    from dynmm_amd import lib as L
    mapped = sorted(L._mapped_hip_runtimes())
    assert len(mapped) == 1, mapped
    hip = C.CDLL(mapped[0])                     # the runtime of this process (torch's)
    least, greatest = C.c_int(0), C.c_int(0)
    assert hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) == 0
    streams = [ops._wgrad_stream() for _ in range(ops.WGRAD_STREAMS)]
    assert len({s.cuda_stream for s in streams}) == ops.WGRAD_STREAMS
    for s in streams:
        got = C.c_int(-99)
        assert hip.hipStreamGetPriority(C.c_void_p(s.cuda_stream), C.byref(got)) == 0
        assert got.value == max(least.value, 0), (got.value, least.value, greatest.value)
    plan = ops.stream_plan()
    assert ops.stream_plan() is plan and ops.side_stream() is plan.side           # singletons
    assert {s.cuda_stream for s in streams} == {s.cuda_stream for s in plan.wgrad[:ops.WGRAD_STREAMS]}
    assert ops.exchange_stream() is plan.wgrad[ops.WGRAD_STREAMS - 1]
    got = C.c_int(-99)
    assert hip.hipStreamGetPriority(C.c_void_p(plan.side.cuda_stream), C.byref(got)) == 0 and got.value == 0, got.value
    handles = {plan.side.cuda_stream} | {s.cuda_stream for s in plan.wgrad}
    assert len(handles) == 1 + len(plan.wgrad) and 0 not in handles and torch.cuda.current_stream().cuda_stream not in handles
    # the plan was CHECKED at creation: every pair among (caller, depth, wgrad0, wgrad1) runs two spin kernels side by side
    rep = plan.report
    assert rep['checked'] and rep['clean'] and len(rep['pair_ratio']) == 6, rep
    assert max(rep['pair_ratio'].values()) <= plan.CLEAN
    # ... and the probe tells a shared queue from an independent one: a stream against ITSELF serialises (ratio ~2)
    single = rep['spin_ms']
    assert plan._pair_ratio(plan.side, plan.side, single) > 1.5
    assert plan._pair_ratio(torch.cuda.current_stream(), plan.side, single) <= plan.CLEAN
    # a plan whose depth stream collides (here: IS the caller's stream) repairs itself
    bad = ops._StreamPlan.__new__(ops._StreamPlan)
    bad.device, bad.handles, bad.report = plan.device, [], {}
    bad.side, bad.wgrad = torch.cuda.current_stream(), [ops.low_priority_stream() for _ in range(ops.WGRAD_STREAMS)]
    rep2 = bad.verify()
    assert rep2['checked'] and rep2['replaced'] >= 1 and rep2['clean'], rep2
    assert bad.side.cuda_stream != torch.cuda.current_stream().cuda_stream
    a = rnd(1 << 20, seed=1).cuda()
    s = streams[0]
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        b = a * 2.0
    torch.cuda.current_stream().wait_stream(s)
    assert torch.equal(b, a + a)


@pytest.mark.parametrize('case', [(4, 128, 24, 32, 128, (3, 1), (1, 0), True),      # three-tap kernel, vertical taps
                                  (4, 64, 24, 32, 64, (1, 3), (0, 1), True),        # three-tap kernel, 64-row tile
                                  (3, 128, 15, 20, 256, (1, 3), (0, 1), True),      # ... horizontal taps, M = 900 (ragged)
                                  (2, 128, 12, 16, 128, (1, 1), (0, 0), True),      # vectorised 128x128 kernel
                                  (4, 128, 24, 32, 128, (3, 3), (1, 1), False),       # 3x3 on the three-tap kernel (tap rows as k-tiles)
                                  (3, 64, 15, 20, 64, (3, 3), (1, 1), True),           # ... 64-row tile, bias, ragged M = 900
                                  (3, 128, 30, 40, 256, (3, 1), (1, 0), True, (2, 1)),  # stride-2 kernel (conv_wgrad_s2.hip), vertical taps
                                  (3, 128, 15, 40, 128, (1, 3), (0, 1), True, (1, 2)),  # ... horizontal taps, ragged M = 900
                                  (64, 120, 1, 50, 2048, (1, 1), (0, 0), True),      # Linear 120 -> 2048 over [B, D, T]: 1x1 with Ci % 64 != 0 on the
                                  (64, 60, 1, 50, 180, (1, 1), (0, 0), True),        # grouped-row loader (one tap: any Ci), one reduction launch
                                  (96, 10, 1, 50, 30, (1, 1), (0, 0), True),         # ... per group; K = 10 < one row group
                                  (50, 2048, 1, 50, 120, (1, 1), (0, 0), True)])     # Linear 2048 -> 120
def test_grouped_weight_gradients(ops, case):
    """dynmm_conv2d_wgrad_group through the C ABI: 3 same-geometry convolutions in one launch against fp64 torch (and
    the bias gradients that ride along), bit-identical between two calls, and the n = 1 / not-groupable fallbacks."""
    import ctypes as C
    from dynmm_amd import lib as L
    lib = L.load()
    N, Ci, H, W, Co, k, pad, bias, *rest = case
    stride = rest[0] if rest else (1, 1)
    st = torch.cuda.current_stream().cuda_stream
    xs = [rnd(N, Ci, H, W, seed=10 + i).cuda() for i in range(3)]
    w0 = torch.empty(Co, Ci, *k)
    g = ops._geom(xs[0], None, w0, stride, pad)
    dys = [rnd(N, Co, g.Ho, g.Wo, seed=20 + i).cuda() for i in range(3)]
    ref_w, ref_b = [], []
    for x, dy in zip(xs, dys):
        xd = x.double().cpu()
        wd = torch.zeros(Co, Ci, *k, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xd, wd, None, stride, pad)
        y.backward(dy.double().cpu())
        ref_w.append(wd.grad)
        ref_b.append(dy.double().cpu().sum((0, 2, 3)))
    assert lib.dynmm_conv2d_wgrad_groupable(C.byref(g)) in (1, 2)
    if k != (1, 1) and Ci % 64 == 0 and Co % 64 == 0 and W % 4 == 0 and g.Wo >= 16:
        assert lib.dynmm_conv2d_wgrad_variant(C.byref(g)) == 6          # conv_wgrad_v6.hip (Winograd pairs), 3x3 included

    def run(n):
        dws = [torch.empty(Co, Ci, *k, device='cuda') for _ in range(n)]
        dbs = [torch.empty(Co, device='cuda') for _ in range(n)] if bias else None
        nbytes = lib.dynmm_conv2d_wgrad_group_workspace_bytes(C.byref(g), n)
        ws = torch.empty(max(nbytes // 4, 1), device='cuda')
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts[:n]])
        L.check(lib.dynmm_conv2d_wgrad_group(n, arr(xs), arr(dys), arr(dws), arr(dbs) if bias else None, ws.data_ptr(), nbytes,
                                             C.byref(g), st), 'wgrad_group')
        torch.cuda.synchronize()
        return dws, dbs
    a_w, a_b = run(3)
    for i in range(3):
        assert rel(a_w[i], ref_w[i]) < GTOL
        if bias:
            assert rel(a_b[i], ref_b[i]) < GTOL
    b_w, _ = run(3)
    for x, y in zip(a_w, b_w):
        assert torch.equal(x, y)
    s_w, _ = run(1)                                   # a "group" of one = the ordinary launch
    assert rel(s_w[0], ref_w[0]) < GTOL


@pytest.mark.parametrize('case', [(3, 128, 15, 20), (2, 64, 24, 32), (7, 64, 6, 12), (2, 256, 30, 40), (8, 64, 120, 160)])
def test_bn2_backward_reductions_kernel_vs_float64(ops, case):
    """dynmm_conv2d_wino_dgrad_bnred2 through the C ABI (csrc/conv_wino.hip BNRED == 2): dx = (conv_transpose(dy, w) + accum) . [out > 0]
    with the decisions read from the one-bit record dynmm_bn_apply wrote, and the BatchNorm backward's two reductions over dx —
    against float64 with the SAME decisions (they are an input here): dx to GTOL, the reductions to 1e-6 of their absolute sums;
    a NULL accum is a zero one; the last case spreads its 1200 pixel tiles over two slabs of sums."""
    import ctypes as C
    from dynmm_amd import lib as L
    lib = ops._lib()
    st = torch.cuda.current_stream().cuda_stream
    N, Cc, H, W = case
    HW = H * W
    c, idt = rnd(N, Cc, H, W, seed=1), rnd(N, Cc, H, W, seed=2)
    w = rnd(Cc, Cc, 3, 1, seed=3, scale=(3 * Cc) ** -0.5)
    dy, acc = rnd(N, Cc, H, W, seed=4), rnd(N, Cc, H, W, seed=5)
    bn = torch.nn.BatchNorm2d(Cc, eps=1e-3).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    # the forward's own record: out = relu(BN(c) + idt) by dynmm_bn_apply with the bits requested
    cg = c.cuda().requires_grad_(True)
    out = ops.batch_norm_act(cg, bn, 'relu', residual=idt.cuda().requires_grad_(True))
    x_, y_, gamma, mean, invstd, beta, bits = out.grad_fn.saved_tensors
    assert bits is not None
    m = (out.detach() > 0).double().cpu()
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, 3, 1, 1, 1, 1, 0, Cc)
    assert lib.dynmm_conv2d_wino_dgrad_bnred_supported(C.byref(g)) == 1
    ut = torch.empty(lib.dynmm_wino_packed_floats(Cc, Cc, 3, 1), device='cuda')
    w_g = w.cuda()
    L.check(lib.dynmm_wino_pack(w_g.data_ptr(), ut.data_ptr(), None, Cc, Cc, 3, 1, 1, st), 'wino_pack')
    convT = torch.nn.grad.conv2d_input((N, Cc, H, W), w.double(), dy.double(), 1, (1, 0))
    xhat = (c.double() - mean.double().cpu().view(1, -1, 1, 1)) * invstd.double().cpu().view(1, -1, 1, 1)
    dy_g, acc_g = dy.cuda(), acc.cuda()
    for accum in (acc, None):
        ref = (convT + (accum.double() if accum is not None else 0.0)) * m
        ns = lib.dynmm_conv2d_wino_dgrad_bnred_slots(C.byref(g))
        assert ns == (2 if N * ((H + 1) // 2) * W >= 2 * 600 * 64 else 1)
        slabs = torch.zeros(ns, 2, Cc, device='cuda', dtype=torch.float64)
        dx = torch.full((N, Cc, H, W), float('nan'), device='cuda')
        L.check(lib.dynmm_conv2d_wino_dgrad_bnred2(dy_g.data_ptr(), ut.data_ptr(), acc_g.data_ptr() if accum is not None else None,
                                                   cg.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr(), slabs.data_ptr(),
                                                   dx.data_ptr(), C.byref(g), st), 'bnred2')
        torch.cuda.synchronize()
        sums = slabs.sum(0)
        assert ns == 1 or bool((slabs[1] != 0).any())
        assert rel(dx, ref) < GTOL
        assert ((sums[0].cpu() - ref.sum((0, 2, 3))).abs() / ref.abs().sum((0, 2, 3))).max() < 1e-6
        assert ((sums[1].cpu() - (ref * xhat).sum((0, 2, 3))).abs() / (ref * xhat).abs().sum((0, 2, 3))).max() < 1e-6


@pytest.mark.parametrize('case,fits', [((3, 128, 15, 20), True), ((2, 64, 24, 32), True), ((2, 64, 24, 33), False)])
def test_bn2_backward_reductions_from_the_next_block(ops, case, fits):
    """Two chained NonBottleneck1D blocks (resnet.py:87-147) as the encoder stages run them (chain=True): the input-gradient launch of
    block 1's first convolution masks the complete gradient of block 0's output with the forward's decisions and leaves bn2's two
    reductions (ops.BNLink second form) — bn_bwd_reduce is not launched for it, and its apply pass runs without bits and writes no
    second masked copy.  Same gradients as the unfused path (same forward, same decisions) to 1e-5; a spy asserts which path ran;
    a width off the kernel's rules takes the unfused path."""
    from dynmm_amd.nn.blocks import NonBottleneck1D
    N, Cc, H, W = case
    torch.manual_seed(N * 100 + Cc)
    b0, b1 = NonBottleneck1D(Cc, Cc).cuda().train(), NonBottleneck1D(Cc, Cc).cuda().train()
    x = rnd(N, Cc, H, W, seed=1).relu_().cuda()
    gy = rnd(N, Cc, H, W, seed=2).cuda()
    lib = ops._lib()
    old = ops.BN_BWD_FUSE
    named = [(f'b{j}.{n}', q) for j, m in enumerate((b0, b1)) for n, q in m.named_parameters()]
    params = [q for _, q in named]

    def run(fuse):
        ops.BN_BWD_FUSE = fuse
        for m in (b0, b1):
            m.bn1.reset_running_stats()
            m.bn2.reset_running_stats()
        for q in params:
            q.grad = None
        xi = x.clone().requires_grad_(True)
        with _CallSpy(lib, 'dynmm_conv2d_wino_dgrad_bnred2', 'dynmm_conv2d_wino_dgrad_bnred', 'dynmm_bn_bwd_reduce') as spy:
            y = b1(b0(xi), chain=True)
            y.backward(gy)
            torch.cuda.synchronize()
        return [xi.grad.clone()] + [q.grad.clone() for q in params], y.detach().clone(), dict(spy.count)
    try:
        (ga, ya, ca), (gb, yb, cb) = run(False), run(True)
    finally:
        ops.BN_BWD_FUSE = old
    assert torch.equal(ya, yb)
    assert ca == {'dynmm_conv2d_wino_dgrad_bnred2': 0, 'dynmm_conv2d_wino_dgrad_bnred': 0, 'dynmm_bn_bwd_reduce': 4}, ca
    if fits:       # bn1 of both blocks through BNRED, bn2 of block 0 through BNRED == 2, bn2 of block 1 (nobody downstream) on its own
        assert cb == {'dynmm_conv2d_wino_dgrad_bnred2': 1, 'dynmm_conv2d_wino_dgrad_bnred': 2, 'dynmm_bn_bwd_reduce': 1}, cb
    else:
        assert cb['dynmm_conv2d_wino_dgrad_bnred2'] == 0 and cb['dynmm_bn_bwd_reduce'] >= 2, cb
    for name, a, b in zip(['x'] + [n for n, _ in named], ga, gb):
        if name.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')):      # a bias in front of a training-mode BatchNorm: analytically zero,
            assert a.abs().max() < 1e-3 and b.abs().max() < 1e-3     # what is left is rounding noise on either path
            continue
        assert rel(b, a) < 1e-5, name
