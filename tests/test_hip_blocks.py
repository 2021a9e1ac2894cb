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
