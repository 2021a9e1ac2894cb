"""GPU: block-level parity (forward, input gradient and every parameter gradient) of the HIP
modules against the oracle's functional restatement, at the shapes where the blocks run in the
96x128 / 480x640 models."""
import contextlib

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


@contextlib.contextmanager
def hip_relu_decisions(trace, tau=1e-5):
    """Run the oracle with the ReLU decisions the HIP pass took — but only where the oracle's own pre-activation lies within
    `tau` (relative to the tensor's maximum) of zero.  There two correct fp32 evaluations of the same sums decide differently
    (tests/flip_probe.py: over 12 seeds of the decoder module the direct kernels differ from the fp64 decision in 6, the
    Winograd kernels in 5, and ONE such flip moves a BatchNorm parameter gradient of these small maps by 5e-3 ... 1.2e-1), and a
    gradient is only comparable at equal decisions.  Everywhere else the HIP decision must EQUAL the oracle's
    (`outside_band` is asserted to be 0 by the caller: a wrong mask, tap or residual shows up there), and every traced decision
    must be consumed (the k-th ReLU of a shape on one side is the k-th of that shape on the other)."""
    queues = {}
    for y in trace:
        queues.setdefault(tuple(y.shape), []).append((y > 0).cpu())
    census = {'sites': 0, 'imposed': 0, 'outside_band': 0, 'queues': queues}
    orig = O.F.relu

    def relu(v, inplace=False):
        q = queues.get(tuple(v.shape))
        if not q:
            return orig(v)
        hip = q.pop(0)
        vd = v.detach()
        own = vd > 0
        band = vd.abs() <= tau * vd.abs().max()
        dis = own != hip
        census['sites'] += 1
        census['imposed'] += int((dis & band).sum())
        census['outside_band'] += int((dis & ~band).sum())
        return v * torch.where(band, hip, own).to(v.dtype)
    O.F.relu = relu
    try:
        yield census
    finally:
        O.F.relu = orig


def run_pair(module, ref_fn, inputs, training=True, prefix='m'):
    """module: HIP nn.Module (CPU-constructed); ref_fn(sd, *inputs, training) -> tensor or tuple.  The oracle runs in fp64
    (the truth both fp32 implementations approximate) with the HIP pass's ReLU decisions at its near-ties."""
    from dynmm_amd import ops
    synth.fill_state_dict(module.state_dict(), seed=3)
    sd = {f'{prefix}.{k}': (v.detach().clone().double() if v.dtype.is_floating_point else v.detach().clone())
          for k, v in module.state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}

    module = module.cuda().train(training)
    xs = [x.clone().cuda().requires_grad_(True) for x in inputs]
    ops.ACT_TRACE = []
    try:
        out = module(*xs)
    finally:
        trace, ops.ACT_TRACE = ops.ACT_TRACE, None
    outs = [o for o in (out if isinstance(out, tuple) else (out,)) if o is not None]

    xs_ref = [x.clone().double().requires_grad_(True) for x in inputs]
    with hip_relu_decisions(trace) as census:
        out_ref = ref_fn(sd, *xs_ref, training)
    outs_ref = [o for o in (out_ref if isinstance(out_ref, tuple) else (out_ref,)) if o is not None]
    assert census['outside_band'] == 0, f'{census["outside_band"]} ReLU decisions differ from the fp64 oracle away from zero'
    assert not any(census['queues'].values()), 'traced HIP ReLU outputs the oracle never matched'
    gs = [rnd(*o.shape, seed=11 + i) for i, o in enumerate(outs_ref)]
    torch.autograd.backward(outs_ref, [g.double() for g in gs])
    torch.autograd.backward(outs, [g.cuda() for g in gs])
    if census['imposed']:
        print(f'[ties] {census["imposed"]} ReLU decisions of {census["sites"]} traced activations taken from the HIP pass '
              '(oracle pre-activation within 1e-5 of zero)')
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


@pytest.mark.parametrize('forward', ['all', 'dgrad'])
@pytest.mark.parametrize('seed', [0, 1, 6, 9])
def test_decoder_module(seed, forward):
    """Four input draws x both training-forward kernels (Winograd F(2,3) = the default, direct operand-ring).  Against the
    fp64 oracle taken at face value, each of these draws has a ReLU tie that one or both fp32 forwards resolve the other way
    (tests/flip_probe.py: seed 0 direct, seed 1 Winograd, seeds 6 and 9 both) and a worst parameter-gradient error of
    2e-2 ... 1.2e-1; with the HIP decisions imposed at the oracle's near-ties every gradient meets the strict bar."""
    from dynmm_amd import ops
    from dynmm_amd.nn.decoder import DecoderModule
    m = DecoderModule(128, 128, 3, 40)
    saved, ops.WINO = ops.WINO, forward
    try:
        run_pair(m, lambda sd, x, skip, tr: O.decoder_module(sd, 'm', x, skip, tr, 3),
                 [rnd(3, 128, 12, 16, seed=seed), rnd(3, 128, 24, 32, seed=5 + seed)])
    finally:
        ops.WINO = saved


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


def test_feature_tap_on_an_intermediate_block_keeps_gradients_right():
    """the chain contract (block i is the ONLY consumer of block i - 1's output) is withdrawn when a forward hook can see
    the intermediate tensor.  A hook that keeps block 0's output in the graph (a feature tap entering the loss) must give the same
    gradients as the same computation with the BatchNorm-backward fusion switched off altogether."""
    from dynmm_amd import ops
    from dynmm_amd.nn.blocks import ResNetEncoder
    enc = ResNetEncoder('resnet34', 'NonBottleneck1D', input_channels=3).cuda().train()
    sd = enc.state_dict()
    synth.fill_state_dict(sd, seed=3)
    enc.load_state_dict(sd)
    x0 = rnd(2, 64, 24, 32, seed=5).cuda()
    taps = []
    hook = enc.layer1[0].register_forward_hook(lambda m, i, o: taps.append(o))

    def run(fuse):
        taps.clear()
        saved, ops.BN_BWD_FUSE = ops.BN_BWD_FUSE, fuse
        try:
            for p_ in enc.parameters():
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            y = enc.forward_layer1(x)
            loss = y.square().mean() + 0.5 * taps[0].square().mean()          # the tapped feature map enters the loss
            loss.backward()
            torch.cuda.synchronize()
            return x.grad.clone(), {k: v.grad.clone() for k, v in enc.layer1.named_parameters()}
        finally:
            ops.BN_BWD_FUSE = saved
    try:
        gx1, gp1 = run(True)
        gx0, gp0 = run(False)
    finally:
        hook.remove()
    assert rel(gx1, gx0) < 1e-5, rel(gx1, gx0)
    scale = max(v.abs().max().item() for v in gp0.values())
    for k in gp0:
        # (a bias in front of a BatchNorm has a mathematically zero gradient: 1e-10 of rounding, compared on the common scale)
        err = (gp1[k] - gp0[k]).abs().max().item() / max(gp0[k].abs().max().item(), 1e-2 * scale)
        assert err < 1e-4, (k, err)
