"""The learned 2x up-sampling (nearest + depthwise 3x3, model.py:404-410) forward and input gradient in isolation at the decoder's
shapes, batch 32: DYNMM_LIB=<other .so> python scratch/r6/upsample_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dynmm_amd import lib as L  # noqa: E402

if os.environ.get('DYNMM_LIB'):
    L.LIB_PATH = os.environ['DYNMM_LIB']
lib = L.load()
st = torch.cuda.current_stream().cuda_stream


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
N = 32
for C, H, W, skip in ((512, 15, 20, True), (256, 30, 40, True), (128, 60, 80, True), (40, 120, 160, False)):
    x = torch.randn(N, C, H, W, device='cuda')
    w, b = torch.randn(C, 9, device='cuda'), torch.randn(C, device='cuda')
    sk = torch.randn(N, C, 2 * H, 2 * W, device='cuda') if skip else None
    y = torch.empty(N, C, 2 * H, 2 * W, device='cuda')
    g = torch.randn_like(y)
    dx = torch.empty_like(x)
    tf = timed(lambda: L.check(lib.dynmm_upsample2x_dw3x3_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                             None if sk is None else sk.data_ptr(), y.data_ptr(), N, C, H, W, st), 'f'))
    tb = timed(lambda: L.check(lib.dynmm_upsample2x_dw3x3_bwd(g.data_ptr(), None, w.data_ptr(), dx.data_ptr(), None, None, None,
                                                             N, C, H, W, st), 'b'))
    out.append(f'{C}x{H}x{W}: fwd {tf:.1f} bwd_dx {tb:.1f} us (chk {y.double().sum().item():.3f} {dx.double().sum().item():.3f})')
print(' | '.join(out))
