import torch
from dynmm_amd import ops
x = torch.relu(torch.randn(32, 64, 240, 320, device='cuda')).requires_grad_(True)
for _ in range(2):
    y = ops.max_pool_3x3_s2(x); y.backward(torch.ones_like(y))
torch.cuda.synchronize()
def t(f, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    print('fwd (no idx) us', round(t(lambda: ops.max_pool_3x3_s2(x))))
def fb():
    y = ops.max_pool_3x3_s2(x); y.backward(g)
g = torch.ones(32, 64, 120, 160, device='cuda')
print('fwd+bwd us', round(t(fb)))
