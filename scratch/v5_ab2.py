import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops, synth, lib as L
from dynmm_amd.nn.blocks import ResNetEncoder
lib = L.load()
class Wrap(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.e = ResNetEncoder('resnet34', 'NonBottleneck1D', 1)
    def forward(self, x):
        y = ops.max_pool_3x3_s2(self.e.forward_first_conv(x))
        return self.e.forward_layer2(self.e.forward_layer1(y))
real_fwd, real_dg = lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad
def run(mode, sync=0):
    lib.dynmm_debug_set_igemm_v5(mode)
    def fwd(*a):
        if sync & 1: torch.cuda.synchronize()
        rc = real_fwd(*a)
        if sync & 2: torch.cuda.synchronize()
        return rc
    def dg(*a):
        if sync & 1: torch.cuda.synchronize()
        rc = real_dg(*a)
        if sync & 2: torch.cuda.synchronize()
        return rc
    lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad = fwd, dg
    m = Wrap(); del m.e.layer3, m.e.layer4
    synth.fill_state_dict(m.state_dict(), seed=3)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 96, 128, generator=g).cuda().requires_grad_(True)
    y = m(x)
    gy = torch.randn(y.shape, generator=g).cuda()
    gy0 = gy.clone()
    y.backward(gy)
    torch.cuda.synchronize()
    print('mode', mode, 'gy modified in place:', float((gy - gy0).abs().max()))
    return {'y': y.detach().clone(), 'dx': x.grad.clone(), **{n: p.grad.clone() for n, p in m.named_parameters()}}
a = run(0); b = run(1); c = run(1, 3)
d_ = run(1, 1); e_ = run(1, 2)
for nm, t in (('sync before', d_), ('sync after', e_)):
    print(nm, max(float((a[k] - t[k]).abs().max() / (a[k].abs().max() + 1e-30)) for k in a if 'bias' not in k))
for k in a:
    d = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30)); d2 = float((a[k] - c[k]).abs().max() / (a[k].abs().max() + 1e-30))
    if d > 1e-4 or d2 > 1e-4 or k in ('y', 'dx'):
        print(f'{k:40s} old-vs-v5 {d:.2e}   old-vs-v5sync {d2:.2e}')
