"""The pyramid pooling module's backward launches in isolation at the benchmark shape (batch 32, 512 channels, 15 x 20, bins 1 / 5):
DYNMM_LIB=<other .so> python scratch/r6/ppm_bwd_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dynmm_amd import lib as L  # noqa: E402

if os.environ.get('DYNMM_LIB'):
    L.LIB_PATH = os.environ['DYNMM_LIB']
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
N, C, H, W, red = 32, 512, 15, 20, 256
Ctot = C + 2 * red
g = torch.randn(N, Ctot, H, W, device='cuda')


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


out = []
for name, c, hh, ww, off in (('x', C, H, W, 0), ('bin1', red, 1, 1, C), ('bin5', red, 5, 5, C + red)):
    dy = torch.empty(N, c, hh, ww, device='cuda')
    us = timed(lambda: L.check(lib.dynmm_nearest_into_bwd(g.data_ptr(), dy.data_ptr(), N, c, hh, ww, Ctot, off, H, W, st), 'nib'))
    out.append(f'nearest_into_bwd[{name}] {us:.1f} us (sum {dy.double().sum().item():.6f})')
for b in (1, 5):
    gp = torch.randn(N, C, b, b, device='cuda')
    dx = torch.empty(N, C, H, W, device='cuda')
    us = timed(lambda: L.check(lib.dynmm_adaptive_avgpool_bwd(gp.data_ptr(), dx.data_ptr(), N * C, H, W, b, b, st), 'apb'))
    out.append(f'adaptive_avgpool_bwd[{b}] {us:.1f} us')
print(' | '.join(out))
