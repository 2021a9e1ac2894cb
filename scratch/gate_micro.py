"""micro-benchmark of the gate head's first conv (128 -> 8, 5x5 s2, two inputs) fwd / dgrad / wgrad."""
import sys, torch
from dynmm_amd import ops
torch.manual_seed(0)
N = 32
r = torch.randn(N, 64, 120, 160, device='cuda', requires_grad=True)
d = torch.randn(N, 64, 120, 160, device='cuda', requires_grad=True)
w = (torch.randn(8, 128, 5, 5, device='cuda') * 0.02).requires_grad_(True)
b = torch.zeros(8, device='cuda', requires_grad=True)
def run():
    y = ops.conv2d(r, w, b, 2, 0, x2=d)
    y.backward(torch.ones_like(y))
ops.PROFILE = []
for _ in range(3): run()
torch.cuda.synchronize()
import collections
agg = collections.defaultdict(list)
for name, flops, e0, e1, *_ in ops.PROFILE:
    agg[name].append(e0.elapsed_time(e1) * 1e3)
for k, v in agg.items(): print(k, [round(x) for x in v])

print('--- stems')
ops.PROFILE = []
for ci in (3, 1):
    x = torch.randn(N, ci, 480, 640, device='cuda')
    ws = (torch.randn(64, ci, 7, 7, device='cuda') * 0.05).requires_grad_(True)
    for _ in range(3):
        y = ops.conv2d(x, ws, None, 2, 3)
        y.backward(torch.ones_like(y))
torch.cuda.synchronize()
agg = collections.defaultdict(list)
for name, flops, e0, e1, shp in ops.PROFILE:
    agg[(name, shp[1])].append(e0.elapsed_time(e1) * 1e3)
for k, v in agg.items(): print(k, [round(x) for x in v])
