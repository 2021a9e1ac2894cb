"""Per-workgroup phase timeline of the implicit-GEMM forward kernel (trace build, -DDYNMM_TRACE)."""
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
out = {}
shapes = ((128, 60, 80, 3, 1), (128, 60, 80, 1, 3), (256, 30, 40, 3, 1), (64, 120, 160, 3, 1))
for (Cc, H, W, KH, KW) in shapes:
    x = torch.randn(N, Cc, H, W, device='cuda')
    b = torch.zeros(Cc, device='cuda')
    y = torch.empty_like(x)
    w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    wp = torch.empty(KH * KW * Cc * Cc, device='cuda')
    lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), None, Cc, Cc, KH, KW, st)
    nblk = 8192
    tr = torch.zeros(nblk, 6, dtype=torch.int64, device='cuda')

    def run():
        return lib.dynmm_conv2d_fwd(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), None, y.data_ptr(),
                                    C.byref(g), 0, st)
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
    out[f'{Cc}_{H}x{W}_{KH}x{KW}'] = t
    t0 = t[:, 0].min()
    rel = (t[:, :4] - t0) / 100.0     # 100 MHz -> us
    print(f'C={Cc} {H}x{W} k{KH}x{KW}: untraced {us:.1f} us; traced span {rel[:, 3].max():.1f} us; blocks {len(t)}')
    for nm, a, bb in (('prologue', 0, 1), ('loop', 1, 2), ('epilogue', 2, 3)):
        d = rel[:, bb] - rel[:, a]
        print('   %-9s us  mean %.2f p50 %.2f p95 %.2f max %.2f' % (nm, d.mean(), np.median(d), np.percentile(d, 95), d.max()))
    dw_ = (t[:, 3] - t[:, 0]).astype(np.float64) / 100.0
    dc_ = (t[:, 5] - t[:, 4]).astype(np.float64)
    print('   shader clock (MHz): mean %.0f' % (dc_ / dw_).mean())
    T = rel[:, 3].max()
    bw = 2.0
    bins = np.arange(0, T + bw, bw)

    def occupancy(a, bb):
        h = np.zeros(len(bins))
        for s, e in zip(a, bb):
            h[int(s // bw):int(e // bw) + 1] += 1
        return h
    hp, hl, he = occupancy(rel[:, 0], rel[:, 1]), occupancy(rel[:, 1], rel[:, 2]), occupancy(rel[:, 2], rel[:, 3])
    print('   t(us): blocks in prologue / loop / epilogue ; starts in bin')
    starts = np.histogram(rel[:, 0], bins=np.append(bins, T + 2 * bw))[0]
    for i in range(len(bins)):
        print(f'   {bins[i]:6.0f}: {int(hp[i]):5d} {int(hl[i]):5d} {int(he[i]):5d} ; {int(starts[i]):5d}')
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, 'gpurun_out', 'trace.npz'), **out)
