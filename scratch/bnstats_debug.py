"""DecoderModule, fused conv->BN statistics vs the statistics pass: where do the two runs part?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops, synth
from dynmm_amd.nn.decoder import DecoderModule
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape)); return torch.randn(*shape, generator=g)
m = DecoderModule(128, 128, 3, 40)
synth.fill_state_dict(m.state_dict(), seed=3)
m = m.cuda().train()
x0, s0 = rnd(3, 128, 12, 16), rnd(3, 128, 24, 32, seed=5)
runs = []
orig = ops._stats_tiles
for fused in (True, False, True):
    ops._stats_tiles = orig if fused else (lambda g: 0)
    acts = {}
    hooks = []
    for name, mod in m.named_modules():
        if name in ('conv3x3',) or name.startswith('decoder_blocks.') and name.count('.') == 1:
            hooks.append(mod.register_forward_hook(lambda mod_, i, o, name=name: acts.__setitem__(name, o.detach().clone())))
    m.zero_grad()
    x, s = x0.clone().cuda().requires_grad_(True), s0.clone().cuda().requires_grad_(True)
    out, side = m(x, s)
    g1, g2 = rnd(*out.shape, seed=11).cuda(), rnd(*side.shape, seed=12).cuda()
    torch.autograd.backward([out, side], [g1, g2])
    for h in hooks: h.remove()
    runs.append((acts, {n: p.grad.clone() for n, p in m.named_parameters()}, x.grad.clone()))
ops._stats_tiles = orig
def cmp(a, b, what):
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        mx = b[k].abs().max().item()
        print(f'{what:8s} {k:40s} max rel {d.max().item() / max(mx, 1e-30):.2e}  elements > 1e-4 max: {int((d > 1e-4 * mx).sum())} of {d.numel()}')
print('--- fused vs fused (determinism)'); cmp(runs[0][0], runs[2][0], 'act')
print('--- fused vs unfused'); cmp(runs[0][0], runs[1][0], 'act'); cmp(runs[0][1], runs[1][1], 'grad')
