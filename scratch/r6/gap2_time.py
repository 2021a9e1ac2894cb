"""gap2 (SE squeeze of the two encoders' stage outputs) in isolation at the four stage shapes of config P, batch 32:
DYNMM_LIB=<other .so> python scratch/r6/gap2_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dynmm_amd import lib as L  # noqa: E402

if os.environ.get('DYNMM_LIB'):
    L.LIB_PATH = os.environ['DYNMM_LIB']
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
out = []
for C, H, W in ((64, 120, 160), (128, 60, 80), (256, 30, 40), (512, 15, 20)):
    N = 32
    xr, xd = torch.randn(N, C, H, W, device='cuda'), torch.randn(N, C, H, W, device='cuda')
    sr, sd = torch.empty(N * C, device='cuda'), torch.empty(N * C, device='cuda')
    f = lambda: L.check(lib.dynmm_gap2_fwd(xr.data_ptr(), xd.data_ptr(), sr.data_ptr(), sd.data_ptr(), N * C, H * W, st), 'gap2')
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = xr.double().mean((2, 3)).flatten()
    err = ((sr.double() - ref).abs().max() / ref.abs().max()).item()
    out.append(f'C={C} {H}x{W}: {us:.1f} us ({2 * xr.numel() * 4 / us / 1e6:.2f} TB/s, err {err:.1e})')
print(' | '.join(out))
