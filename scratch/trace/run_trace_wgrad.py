"""Phase timeline of the wgrad kernel (trace build)."""
import os, sys, ctypes as C
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from dynmm_amd import lib as L
lib = C.CDLL(os.path.join(HERE, os.environ.get('TRACE_LIB', 'libdynmm_trace.so')))
for name, (res, args) in L.SIGNATURES.items():
    f = getattr(lib, name); f.restype = res; f.argtypes = args
lib.dynmm_debug_set_trace.argtypes = [C.c_void_p]
N = 32
st = torch.cuda.current_stream().cuda_stream
for (Cc, H, W, KH, KW) in ((128, 60, 80, 3, 1), (256, 30, 40, 3, 1), (64, 120, 160, 3, 1)):
    x = torch.randn(N, Cc, H, W, device='cuda')
    gy = torch.randn(N, Cc, H, W, device='cuda')
    dw = torch.empty(Cc, Cc, KH, KW, device='cuda')
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
    ws = torch.empty(max(nbytes // 4, 1), device='cuda')
    tr = torch.zeros(8192, 6, dtype=torch.int64, device='cuda')

    def run():
        return lib.dynmm_conv2d_wgrad(x.data_ptr(), None, gy.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), nbytes, C.byref(g), st)
    lib.dynmm_debug_set_trace(None)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1000
    lib.dynmm_debug_set_trace(tr.data_ptr())
    torch.cuda.synchronize()
    assert run() == 0
    torch.cuda.synchronize()
    lib.dynmm_debug_set_trace(None)
    t = tr.cpu().numpy()
    t = t[t[:, 0] != 0]
    t0 = t[:, 0].min()
    rel = (t[:, :4] - t0) / 100.0
    fl = 2.0 * N * H * W * KH * KW * Cc * Cc
    print(f'wgrad C={Cc} {H}x{W} k{KH}x{KW}: untraced {us:.1f} us (incl. slab reduce) {fl/us/1e6:.1f} TF; traced span {rel[:, 3].max():.1f} us; blocks {len(t)}; ws {nbytes/1e6:.1f} MB')
    for nm, a, bb in (('loop', 0, 2), ('epilogue', 2, 3)):
        d = rel[:, bb] - rel[:, a]
        print('   %-9s us  mean %.2f p50 %.2f p95 %.2f max %.2f' % (nm, d.mean(), np.median(d), np.percentile(d, 95), d.max()))
    dw_ = (t[:, 3] - t[:, 0]).astype(np.float64) / 100.0
    dc_ = (t[:, 5] - t[:, 4]).astype(np.float64)
    print('   clock64 ticks per us (MHz): mean %.1f min %.1f max %.1f' % ((dc_ / dw_).mean(), (dc_ / dw_).min(), (dc_ / dw_).max()))
    print('   start times: p50 %.1f p95 %.1f max %.1f' % (np.median(rel[:, 0]), np.percentile(rel[:, 0], 95), rel[:, 0].max()))
    # MFMA throughput over time (uniform progress inside each workgroup's loop)
    T = rel[:, 3].max(); bins = np.arange(0, T + 5, 5.0); prog = np.zeros(len(bins) - 1)
    for a_, b_ in zip(rel[:, 0], rel[:, 2]):
        lo = np.clip(bins[:-1], a_, b_); hi = np.clip(bins[1:], a_, b_); prog += (hi - lo) / max(b_ - a_, 1e-9)
    print('   TF/s per 5 us:', ' '.join(f'{v:.0f}' for v in prog * (fl / len(t)) / 5e-6 / 1e12))
    d = rel[:, 2] - rel[:, 0]
    nb = len(t)
    print('   loop us by blockIdx class: [0,256) mean %.1f | [256,512) mean %.1f | even %.1f odd %.1f | by xcd %s' % (
        d[:256].mean(), d[256:].mean(), d[0::2].mean(), d[1::2].mean(), ' '.join('%.0f' % d[i::8].mean() for i in range(8))))
    print('   quartiles of loop us over blockIdx ranges of 64:', ' '.join('%.0f' % d[i:i + 64].mean() for i in range(0, nb, 64)))
