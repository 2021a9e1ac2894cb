"""How much of a Winograd forward launch is per-workgroup overhead (prologue + epilogue) and how much the K loop?  Same output tile
grid (Co, H, W fixed), reduction channels Ci varied: T(Ci) = a + b Ci per launch.  Product library, horizontal and vertical forward."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())


def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000


_w = torch.randn(4096, 4096, device='cuda')
for _ in range(200): _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
for (N, Co, H, W) in [(32, 64, 120, 160), (32, 128, 60, 80), (32, 256, 30, 40)]:
    for (KH, KW) in ((1, 3), (3, 1)):
        rows = []
        for Ci in (Co // 2, Co, 2 * Co, 4 * Co):
            if Ci < 32:
                continue
            g = L.ConvGeom(N, Ci, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Ci)
            if not lib.dynmm_conv2d_wino_supported(C.byref(g), 0):
                continue
            x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, KH, KW, device='cuda') * 0.05
            b = torch.randn(Co, device='cuda'); y = torch.empty(N, Co, H, W, device='cuda')
            u = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, KH, KW), device='cuda')
            assert lib.dynmm_wino_pack(p(w), p(u), None, Co, Ci, KH, KW, 0, st) == 0
            call = lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(u), p(b), None, p(y), C.byref(g), 1, st)
            assert call() == 0
            rows.append((Ci, tm(call)))
        if len(rows) >= 2:
            (c0, t0), (c1, t1) = rows[0], rows[-1]
            slope = (t1 - t0) / (c1 - c0)
            a = t0 - slope * c0
            print(f'Co={Co} {H}x{W} {KH}x{KW}: ' + '  '.join(f'Ci={c}: {t:.1f} us' for c, t in rows) +
                  f'   fit: {a:.1f} us fixed + {slope:.3f} us per channel -> at Ci = Co the fixed part is {a / (a + slope * Co):.0%}', flush=True)
