"""product 2-D F(2x2,3x3) kernel vs the 1-D forms it replaces, isolated launches at batch 32 (us)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
torch.manual_seed(0)
for (N, Ci, H, W, Co) in [(32, 64, 120, 160, 64), (32, 128, 60, 80, 128), (32, 256, 30, 40, 256), (32, 512, 15, 20, 512), (32, 128, 120, 160, 40)]:
    g = L.ConvGeom(N, Ci, H, W, Co, H, W, 3, 3, 1, 1, 1, 1, Ci)
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    dy = torch.randn(N, Co, H, W, device='cuda'); mask = torch.randn_like(x); acc = torch.randn_like(x); y = torch.empty_like(dy); dx = torch.empty_like(x)
    u2f = torch.empty(lib.dynmm_wino2d_packed_floats(Co, Ci), device='cuda'); u2d = torch.empty_like(u2f)
    lib.dynmm_wino2d_pack(p(w), p(u2f), None, Co, Ci, 0, st)
    u1f = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, 3, 3), device='cuda'); lib.dynmm_wino_pack(p(w), p(u1f), None, Co, Ci, 3, 3, 0, st)
    fl = 2.0 * N * H * W * 9 * Ci * Co
    t2 = tm(lambda: lib.dynmm_conv2d_wino2d_fwd(p(x), p(u2f), p(b), None, p(y), None, 0, C.byref(g), 1, st))
    t1 = tm(lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(u1f), p(b), None, p(y), C.byref(g), 1, st))
    line = f'{(N, Ci, H, W, Co)}: fwd 2-D {t2:.0f} us ({fl / t2 / 1e6:.0f} TF) | F(2,3) looped {t1:.0f} ({fl / t1 / 1e6:.0f})'
    if Ci % 64 == 0 and Co % 8 == 0:
        lib.dynmm_wino2d_pack(p(w), p(u2d), None, Co, Ci, 1, st)
        u43 = torch.empty(lib.dynmm_wino43_packed_floats(Co, Ci, 3, 3), device='cuda'); lib.dynmm_wino43_pack(p(w), p(u43), Co, Ci, 3, 3, st)
        for nm, m_, a_ in (('plain', None, None), ('mask+acc', mask, acc)):
            d2 = tm(lambda: lib.dynmm_conv2d_wino2d_dgrad(p(dy), p(u2d), p(m_), p(a_), p(dx), C.byref(g), st))
            d4 = tm(lambda: lib.dynmm_conv2d_wino43_dgrad(p(dy), p(u43), p(m_), p(a_), p(dx), C.byref(g), st))
            line += f' | dgrad {nm} 2-D {d2:.0f} ({fl / d2 / 1e6:.0f}) F(4,3) {d4:.0f} ({fl / d4 / 1e6:.0f})'
    print(line, flush=True)
