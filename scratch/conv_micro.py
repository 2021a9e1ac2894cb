"""Micro-benchmark of one conv shape through the C ABI (fwd, dgrad, wgrad) for PMC profiling."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops
N, C, H, W = 32, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
which = sys.argv[4] if len(sys.argv) > 4 else 'all'
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
w = (torch.randn(C, C, 3, 1, device='cuda') * 0.05).requires_grad_(True)
b = torch.zeros(C, device='cuda', requires_grad=True)
g = torch.randn(N, C, H, W, device='cuda')
for _ in range(iters):
    y = ops.conv2d(x, w, b, 1, (1, 0), None)
    if which != 'fwd':
        y.backward(g)
torch.cuda.synchronize()
print('done')
