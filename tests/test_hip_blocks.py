"""GPU: block-level parity (forward, input gradient and every parameter gradient) of the HIP
modules against the oracle's functional restatement, at the shapes where the blocks run in the
96x128 / 480x640 models."""
import pytest
import torch

from dynmm_amd import synth
from oracle import dynmm_oracle as O

pytestmark = pytest.mark.gpu
TOL, GTOL = 5e-5, 5e-4


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def tie_tolerant(a, b):
    """Second-chance criterion for GRADIENTS of multi-layer modules.  A train-mode pass of these blocks evaluates ~10^6
    ReLUs; with pre-activations of order 1 a handful of them lie within fp32 rounding (1e-6) of zero, and two correct
    fp32 implementations that sum a convolution in different orders take different ReLU decisions there (measured:
    scratch/v5_ab3.py — ONE flipped decision between the two implicit-GEMM generations, identical inputs to 3e-6, moves
    a weight gradient of the tiny test shapes by 5e-2 of its maximum).  Such a flip is sparse at its origin and small
    everywhere else; a wiring or kernel bug (lost mask, lost residual gradient, wrong tap) is dense and of order one.
    Accept: relative L2 error < 2e-2 with fewer than 2 % of the elements off by more than 10 x GTOL of the maximum."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = (a - b).abs()
    l2 = (d.norm() / b.norm().clamp_min(1e-20)).item()
    frac = (d > 10 * GTOL * b.abs().max()).double().mean().item()
    return l2 < 2e-2 and frac < 0.02


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g)


def run_pair(module, ref_fn, inputs, training=True, prefix='m'):
    """module: HIP nn.Module (CPU-constructed); ref_fn(sd, *inputs, training) -> tensor or tuple."""
    synth.fill_state_dict(module.state_dict(), seed=3)
    sd = {f'{prefix}.{k}': v.detach().clone() for k, v in module.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    xs_ref = [x.clone().requires_grad_(True) for x in inputs]
    out_ref = ref_fn(sd, *xs_ref, training)
    outs_ref = [o for o in (out_ref if isinstance(out_ref, tuple) else (out_ref,)) if o is not None]
    gs = [rnd(*o.shape, seed=11 + i) for i, o in enumerate(outs_ref)]
    torch.autograd.backward(outs_ref, gs)

    module = module.cuda().train(training)
    xs = [x.clone().cuda().requires_grad_(True) for x in inputs]
    out = module(*xs)
    outs = [o for o in (out if isinstance(out, tuple) else (out,)) if o is not None]
    torch.autograd.backward(outs, [g.cuda() for g in gs])
    report = []
    for i, (a, b) in enumerate(zip(outs, outs_ref)):
        report.append((rel(a, b), f'out{i}', TOL))
    grads = []
    for i, (a, b) in enumerate(zip(xs, xs_ref)):
        grads.append((a.grad, b.grad, f'dinput{i}'))
    gmax = max(v.grad.abs().max().item() for v in params.values())
    for name, p in module.named_parameters():
        ref = params[f'{prefix}.{name}'].grad
        if ref.abs().max() < 1e-5 * gmax:   # analytically-zero grads (conv bias before a train-mode BN):
            continue                        # pure rounding noise on both sides
        grads.append((p.grad, ref, name))
    bad = [r for r in report if not r[0] < r[2]]          # outputs: always the strict bar
    tolerated = []
    for a, b, name in grads:
        e = rel(a, b)
        if e < GTOL:
            continue
        if tie_tolerant(a, b):
            tolerated.append((e, name))
        else:
            bad.append((e, name, GTOL))
    assert not bad, sorted(bad, reverse=True)[:10]
    if tolerated:
        print(f'[tie-tolerant] {len(tolerated)} gradient tensors beyond {GTOL} in max norm (largest {max(tolerated)}): '
              'ReLU decisions at rounding-level pre-activations differ from the oracle\'s')
    new_sd = module.state_dict()
    for k, v in sd.items():
        if 'running_' in k:
            assert rel(new_sd[k[len(prefix) + 1:]], v) < 1e-4, k


@pytest.mark.parametrize('c,h,w,n', [(128, 24, 32, 3), (64, 24, 32, 2), (512, 3, 4, 3), (128, 12, 16, 2)])
@pytest.mark.parametrize('training', [True, False])
def test_non_bottleneck_1d(c, h, w, n, training):
    from dynmm_amd.nn.blocks import NonBottleneck1D
    run_pair(NonBottleneck1D(c, c), lambda sd, x, tr: O.non_bottleneck_1d(sd, 'm', x, tr), [rnd(n, c, h, w)], training)


@pytest.mark.parametrize('blk', ['NonBottleneck1D', 'BasicBlock'])
def test_strided_block_with_downsample(blk):
    import torch.nn as nn
    from dynmm_amd.nn import blocks
    down = nn.Sequential(nn.Conv2d(64, 128, 1, stride=2, bias=False), nn.BatchNorm2d(128))
    m = blocks.BLOCKS[blk](64, 128, 2, down)
    fn = O.non_bottleneck_1d if blk == 'NonBottleneck1D' else O.basic_block
    run_pair(m, lambda sd, x, tr: fn(sd, 'm', x, tr, 2), [rnd(2, 64, 24, 32)])


def test_decoder_module():
    from dynmm_amd.nn.decoder import DecoderModule
    m = DecoderModule(128, 128, 3, 40)
    run_pair(m, lambda sd, x, skip, tr: O.decoder_module(sd, 'm', x, skip, tr, 3),
             [rnd(3, 128, 12, 16), rnd(3, 128, 24, 32, seed=5)])


def test_pyramid_pooling():
    from dynmm_amd.nn.context import PyramidPoolingModule
    run_pair(PyramidPoolingModule(512, 128), lambda sd, x, tr: O.pyramid_pooling(sd, 'm', x, tr), [rnd(3, 512, 3, 4)])


@pytest.mark.parametrize('hard', [False, True])
def test_global_gate(hard):
    from dynmm_amd.nn.net import GlobalGate

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.g = GlobalGate(5)

        def forward(self, r, d):
            return self.g(r, d, 0.7, hard)

    w = Wrap()
    run_pair(w, lambda sd, r, d, tr: O.global_gate(sd, 'm.g', r, d, tr, 0.7, hard),
             [rnd(3, 64, 24, 32), rnd(3, 64, 24, 32, seed=9)])


def test_encoder_stage_and_stem():
    from dynmm_amd.nn.blocks import ResNetEncoder

    class Wrap(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.e = ResNetEncoder('resnet34', 'NonBottleneck1D', 1)

        def forward(self, x):
            from dynmm_amd import ops
            y = ops.max_pool_3x3_s2(self.e.forward_first_conv(x))
            return self.e.forward_layer2(self.e.forward_layer1(y))

    def ref(sd, x, tr):
        y = torch.nn.functional.max_pool2d(O.encoder_stem(sd, 'm.e', x, tr), 3, 2, 1)
        y = O.encoder_stage(sd, 'm.e', y, tr, O.Config(), 1)
        return O.encoder_stage(sd, 'm.e', y, tr, O.Config(), 2)

    m = Wrap()
    # only stem/layer1/layer2 take part; drop the unused stages so every parameter gets a gradient
    del m.e.layer3, m.e.layer4
    run_pair(m, ref, [rnd(2, 1, 96, 128)])


@pytest.mark.parametrize('shape', [(4, 128, 30, 40, 128, (1, 3)), (3, 64, 15, 20, 64, (1, 3)), (2, 64, 24, 32, 256, (3, 3)),
                                   (5, 128, 9, 12, 128, (1, 1)), (3, 128, 12, 16, 128, (3, 3)), (3, 128, 24, 32, 128, (1, 3))])
def test_conv_bn_statistics_from_the_conv_epilogue(shape):
    """Training-mode conv -> BatchNorm: the operand-ring kernel's per-tile channel sums (+ the fixed-order finalise) against
    the statistics pass over y they replace — same normalised output, running statistics and gradients (the two orders of
    summation differ below 1e-6), and the partials really were used.  M = 540 / 900 pixels: ragged last tile."""
    import ctypes as C
    from dynmm_amd import ops
    from dynmm_amd.nn.blocks import conv_bn_act
    monkey = ops.CONV_BN_STATS
    ops.CONV_BN_STATS = True                     # opt-in path (default: the statistics pass)
    try:
        _conv_bn_statistics_case(shape, C, ops, conv_bn_act)
    finally:
        ops.CONV_BN_STATS = monkey


def _conv_bn_statistics_case(shape, C, ops, conv_bn_act):
    N, Ci, H, W, Co, k = shape
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(Ci, Co, k, padding=(k[0] // 2, k[1] // 2)).cuda()
    x = torch.randn(N, Ci, H, W, device='cuda')
    res = torch.randn(N, Co, H, W, device='cuda')
    gy = torch.randn(N, Co, H, W, device='cuda')
    g = ops._geom(x, None, conv.weight, (1, 1), conv.padding)
    assert ops._lib().dynmm_conv2d_stats_tiles(C.byref(g)) > 0
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm2d(Co, eps=1e-3).cuda().train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
        torch.manual_seed(4)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(Co) + 0.5); bn.bias.copy_(torch.randn(Co))
        xi = x.clone().requires_grad_(True)
        conv.zero_grad()
        if fused:
            y = conv_bn_act(xi, conv, bn, 'relu', residual=res)
        else:
            yc = ops.conv2d(xi, conv.weight, conv.bias, 1, conv.padding)          # no bn_stats: the statistics pass runs
            assert getattr(yc, '_dynmm_stats', None) is None
            y = ops.batch_norm_act(yc, bn, 'relu', res)
        y.backward(gy)
        outs.append((y.detach(), bn.running_mean.clone(), bn.running_var.clone(), xi.grad.clone(), conv.weight.grad.clone(),
                     bn.weight.grad.clone(), int(bn.num_batches_tracked)))
    yc = ops.conv2d(x.requires_grad_(True), conv.weight, conv.bias, 1, conv.padding, bn_stats=True)
    part, tiles = yc._dynmm_stats
    y64 = yc.detach().double()
    assert rel(part.view(tiles, 2, Co)[:, 0].double().sum(0), y64.sum((0, 2, 3))) < 1e-5
    assert rel(part.view(tiles, 2, Co)[:, 1].double().sum(0), (y64 * y64).sum((0, 2, 3))) < 1e-5
    # closer to the fp64 statistics than the pass over y is (var = E[x^2] - mean^2 from 64-pixel fp32 trees + fp64 above them)
    bn = torch.nn.BatchNorm2d(Co).cuda().train()
    bn.momentum = 1.0
    ops.batch_norm_act(yc, bn, None)
    v64 = y64.var((0, 2, 3), unbiased=True)
    assert ((bn.running_var.double() - v64).abs() / v64).max().item() < 2e-5
    a, b = outs
    assert a[6] == b[6] == 1
    for i in range(6):
        assert rel(a[i], b[i]) < (2e-5 if i >= 3 else 5e-6), i


@pytest.mark.parametrize('shape', [(4, 128, 30, 40), (3, 64, 15, 20), (2, 256, 12, 16)])
def test_bn_backward_reductions_from_the_dgrad_epilogue(shape):
    """NonBottleneck1D, training: the input-gradient kernel of conv3x1_2 also produces the two reductions of bn1's backward
    (sum g, sum g*xhat per pixel tile; ops.BNLink) — every gradient against the same block with bn1's own reduction pass
    (the two summation orders differ below 1e-6; nothing downstream of them takes a ReLU decision), and the link was used."""
    from dynmm_amd import ops
    from dynmm_amd.nn.blocks import NonBottleneck1D
    N, Cc, H, W = shape
    torch.manual_seed(5)
    blk = NonBottleneck1D(Cc, Cc)
    synth.fill_state_dict(blk.state_dict(), seed=4)
    blk = blk.cuda().train()
    x = torch.randn(N, Cc, H, W, device='cuda')
    gy = torch.randn(N, Cc, H, W, device='cuda')
    used = []
    orig = ops._lib().dynmm_conv2d_dgrad_bnstats
    outs = []
    default = ops.BN_BWD_FUSE
    for fused in (True, False):
        ops.BN_BWD_FUSE = fused                    # opt-in path (default: bn1's own reduction pass)
        try:
            blk.zero_grad()
            xi = x.clone().requires_grad_(True)
            blk(xi).backward(gy)
            outs.append([xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
        finally:
            ops.BN_BWD_FUSE = default
    names = ['dx'] + [n for n, _ in blk.named_parameters()]
    gmax = max(t.abs().max().item() for t in outs[1][1:])
    for n, a, b in zip(names, *outs):
        if b.abs().max().item() < 1e-5 * gmax:
            continue                              # analytically-zero gradients (conv bias in front of a train-mode BatchNorm)
        assert rel(a, b) < 2e-5, n
    # the fused run really used the link: bn1's output carries it, conv3x1_2's backward fills it, bn1's backward empties it
    seen = []
    orig = ops._BatchNormAct.backward

    def spy(ctx, g_):
        seen.append(ctx.bnlink is not None and ctx.bnlink.partials is not None)
        return orig(ctx, g_)
    ops._BatchNormAct.backward = staticmethod(spy)
    ops.BN_BWD_FUSE = True
    try:
        xi = x.clone().requires_grad_(True)
        blk(xi).backward(gy)
    finally:
        ops._BatchNormAct.backward = staticmethod(orig)
        ops.BN_BWD_FUSE = default
    assert seen == [False, True], seen            # bn2 (residual + ReLU: own reduction), then bn1 (from the epilogue)
