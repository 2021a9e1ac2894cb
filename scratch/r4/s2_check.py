"""Stride-2 three-tap input gradients on the pair kernel (polyphase form) against fp64 and the round-2 tile kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
p = lambda t: None if t is None else t.data_ptr()


def run(N, Ci, H, W, Co, vert, timing=False):
    KH, KW, SH, SW, PH, PW = (3, 1, 2, 1, 1, 0) if vert else (1, 3, 1, 2, 0, 1)
    Ho, Wo = (H + 2 * PH - KH) // SH + 1, (W + 2 * PW - KW) // SW + 1
    g = L.ConvGeom(N, Ci, H, W, Co, Ho, Wo, KH, KW, SH, SW, PH, PW, Ci)
    assert lib.dynmm_conv2d_wino_supported(C.byref(g), 1) == 2, (N, Ci, H, W, Co, vert)
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, KH, KW, device='cuda') * (2.0 / (3 * Ci)) ** 0.5
    dy = torch.randn(N, Co, Ho, Wo, device='cuda'); mask = torch.randn(N, Ci, H, W, device='cuda'); acc = torch.randn(N, Ci, H, W, device='cuda')
    ut = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, KH, KW), device='cuda')
    L.check(lib.dynmm_wino_pack(p(w), p(ut), None, Co, Ci, KH, KW, 2, st), 'pack s2')
    wd = torch.empty(lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 1), device='cuda')
    lib.dynmm_pack_weight(p(w), None, p(wd), Co, Ci, KH, KW, st)
    dx = torch.full((N, Ci, H, W), float('nan'), device='cuda'); dx2 = torch.empty_like(dx)
    L.check(lib.dynmm_conv2d_wino_dgrad(p(dy), p(ut), p(mask), p(acc), p(dx), C.byref(g), st), 's2 dgrad')
    L.check(lib.dynmm_conv2d_dgrad(p(dy), p(wd), p(mask), p(acc), p(dx2), None, C.byref(g), st), 'tile dgrad')
    torch.cuda.synchronize()
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=(SH, SW), padding=(PH, PW)) * (mask > 0) + acc.double()
    e1 = ((dx.double() - dxr).abs().max() / dxr.abs().max()).item(); e2 = ((dx2.double() - dxr).abs().max() / dxr.abs().max()).item()
    line = f'{(N, Ci, H, W, Co, "3x1s2" if vert else "1x3s2")}: pair kernel {e1:.2e}  tile kernel {e2:.2e}'
    ok = e1 < 3e-6
    if timing:
        def tm(fn, n=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1_.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1_) / n * 1000
        fl = 2.0 * N * Ho * Wo * 3 * Ci * Co
        t1 = tm(lambda: lib.dynmm_conv2d_wino_dgrad(p(dy), p(ut), None, p(acc), p(dx), C.byref(g), st))
        t2 = tm(lambda: lib.dynmm_conv2d_dgrad(p(dy), p(wd), None, p(acc), p(dx2), None, C.byref(g), st))
        line += f' | pair {t1:.1f} us ({fl / t1 / 1e6:.0f} TF)  tile {t2:.1f} us ({fl / t2 / 1e6:.0f} TF)'
        return line, ok, (t1, t2)
    return line, ok, None


bad = 0
for shp in [(3, 64, 16, 24, 128, True), (3, 128, 16, 24, 128, False), (2, 128, 30, 40, 256, True), (2, 256, 15, 40, 256, False), (8, 64, 12, 8, 64, True), (4, 64, 9, 16, 40, False)]:
    line, ok, _ = run(*shp)
    bad += not ok
    print(('OK  ' if ok else 'BAD ') + line, flush=True)
if len(sys.argv) > 1:
    tot = [0.0, 0.0]
    for shp in [(32, 64, 120, 160, 128, True), (32, 128, 60, 160, 128, False), (32, 128, 60, 80, 256, True), (32, 256, 30, 80, 256, False),
                (32, 256, 30, 40, 512, True), (32, 512, 15, 40, 512, False)]:
        line, ok, tms = run(*shp, timing=True)
        bad += not ok
        tot = [a + b for a, b in zip(tot, tms)]
        print(('OK  ' if ok else 'BAD ') + line, flush=True)
    print(f'sum: pair kernel {tot[0]:.0f} us, tile kernel {tot[1]:.0f} us')
sys.exit(1 if bad else 0)
