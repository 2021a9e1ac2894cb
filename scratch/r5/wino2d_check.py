import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dynmm_amd import lib as L
HERE = os.path.dirname(os.path.abspath(__file__))
lib = L.load(); e = C.CDLL(os.path.join(HERE, 'libwino2d.so'))
v = C.c_void_p
e.exp_wino2d_packed_floats.restype = C.c_size_t; e.exp_wino2d_packed_floats.argtypes = [C.c_int, C.c_int]
e.exp_wino2d_pack.argtypes = [v, v, C.c_int, C.c_int, v]
e.exp_conv2d_wino2d_fwd.argtypes = [v, v, v, v, v] + [C.c_int] * 6 + [v]
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
torch.manual_seed(0)
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
def run(N, Ci, H, W, Co, timing=False):
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * (2.0 / (Ci * 9)) ** 0.5
    b = torch.randn(Co, device='cuda'); res = torch.randn(N, Co, H, W, device='cuda')
    ut = torch.empty(e.exp_wino2d_packed_floats(Co, Ci), device='cuda')
    e.exp_wino2d_pack(p(w), p(ut), Co, Ci, st)
    y = torch.full((N, Co, H, W), float('nan'), device='cuda')
    rc = e.exp_conv2d_wino2d_fwd(p(x), p(ut), p(b), p(res), p(y), N, Ci, H, W, Co, 1, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    if N * Ci * H * W < 3e7:
        yr = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + res.double())
        err = ((y.double() - yr).abs().max() / yr.abs().max()).item()
    else:
        err = float('nan')
    line = f'{(N, Ci, H, W, Co)}: 2-D err {err:.2e}'
    if timing:
        g = L.ConvGeom(N, Ci, H, W, Co, H, W, 3, 3, 1, 1, 1, 1, Ci)
        uf = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, 3, 3), device='cuda')
        lib.dynmm_wino_pack(p(w), p(uf), None, Co, Ci, 3, 3, 0, st)
        y1 = torch.empty_like(y)
        L.check(lib.dynmm_conv2d_wino_fwd(p(x), p(uf), p(b), None, p(y1), C.byref(g), 1, st), 'fwd')
        t2 = tm(lambda: e.exp_conv2d_wino2d_fwd(p(x), p(ut), p(b), None, p(y), N, Ci, H, W, Co, 1, st))
        t1 = tm(lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(uf), p(b), None, p(y1), C.byref(g), 1, st))
        fl = 2.0 * N * H * W * 9 * Ci * Co
        torch.cuda.synchronize()
        d = ((y - y1).abs().max() / y1.abs().max()).item()
        line += f' | 2-D {t2:.1f} us ({fl / t2 / 1e6:.0f} TF alg) vs looped F(2,3) {t1:.1f} us ({fl / t1 / 1e6:.0f}); diff {d:.1e}'
    print(line, flush=True)
for shp in [(2, 64, 8, 16, 64), (3, 128, 15, 20, 128), (2, 64, 12, 16, 128), (5, 64, 9, 12, 64), (1, 16, 6, 8, 64), (2, 192, 7, 24, 64)]:
    run(*shp)
if len(sys.argv) > 1:
    for shp in [(32, 64, 120, 160, 64), (32, 128, 60, 80, 128), (32, 256, 30, 40, 256), (32, 512, 15, 20, 512), (32, 128, 120, 160, 64)]:
        run(*shp, timing=True)
