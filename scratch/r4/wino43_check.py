"""conv_wino43.hip (input gradient, F(4,3)) against fp64 and against the F(2,3) kernel: correctness on awkward shapes, then
timing on the encoder / decoder shapes at batch 32."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
p = lambda t: None if t is None else t.data_ptr()


def pack43(w):
    Co, Ci, KH, KW = w.shape
    n = lib.dynmm_wino43_packed_floats(Co, Ci, KH, KW)
    ut = torch.empty(n, device='cuda')
    desc = torch.tensor([[0, 0, Co | (Ci << 32), KH | (KW << 8)]], dtype=torch.int64).cuda()
    L.check(lib.dynmm_wino43_pack_multi(p(w), p(ut), desc.data_ptr(), 1, lib.dynmm_wino43_pack_multi_blocks(Co, Ci, KH, KW), st), 'pack43')
    return ut


def run(N, Ci, H, W, Co, KH, KW, timing=False):
    g = L.ConvGeom(N, Ci, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Ci)
    assert lib.dynmm_conv2d_wino43_supported(C.byref(g)), (N, Ci, H, W, Co, KH, KW)
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, KH, KW, device='cuda') * (2.0 / (Ci * KH * KW)) ** 0.5
    dy = torch.randn(N, Co, H, W, device='cuda'); mask = torch.randn(N, Ci, H, W, device='cuda'); acc = torch.randn(N, Ci, H, W, device='cuda')
    ut = pack43(w)
    ud = torch.empty(lib.dynmm_wino_packed_floats(Co, Ci, KH, KW), device='cuda')
    L.check(lib.dynmm_wino_pack(p(w), p(ud), None, Co, Ci, KH, KW, 1, st), 'pack23')
    dx = torch.full((N, Ci, H, W), float('nan'), device='cuda'); dx2 = torch.empty_like(dx)
    L.check(lib.dynmm_conv2d_wino43_dgrad(p(dy), p(ut), p(mask), p(acc), p(dx), C.byref(g), st), 'wino43 dgrad')
    L.check(lib.dynmm_conv2d_wino_dgrad(p(dy), p(ud), p(mask), p(acc), p(dx2), C.byref(g), st), 'wino23 dgrad')
    torch.cuda.synchronize()
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=(KH // 2, KW // 2)) * (mask > 0) + acc.double()
    e43 = ((dx.double() - dxr).abs().max() / dxr.abs().max()).item(); e23 = ((dx2.double() - dxr).abs().max() / dxr.abs().max()).item()
    line = f'{(N, Ci, H, W, Co, KH, KW)}: F(4,3) {e43:.2e}  F(2,3) {e23:.2e}'
    ok = e43 < 2e-5
    if timing:
        def tm(fn, n=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
        fl = 2.0 * N * H * W * KH * KW * Ci * Co
        res = []
        for mk, ac in ((mask, None), (None, None)):
            t43 = tm(lambda: lib.dynmm_conv2d_wino43_dgrad(p(dy), p(ut), p(mk), p(ac), p(dx), C.byref(g), st))
            t23 = tm(lambda: lib.dynmm_conv2d_wino_dgrad(p(dy), p(ud), p(mk), p(ac), p(dx2), C.byref(g), st))
            res += [t43, t23]
        line += f' | mask: F(4,3) {res[0]:.1f} us ({fl / res[0] / 1e6:.0f} TF alg) F(2,3) {res[1]:.1f} ({fl / res[1] / 1e6:.0f}) | plain: {res[2]:.1f} ({fl / res[2] / 1e6:.0f}) vs {res[3]:.1f} ({fl / res[3] / 1e6:.0f})'
        return line, ok, res
    return line, ok, None


bad = 0
for shp in [(3, 128, 15, 20, 128, 1, 3), (5, 64, 17, 20, 64, 3, 1), (2, 128, 9, 16, 256, 3, 1), (2, 192, 8, 24, 64, 1, 3), (3, 64, 30, 16, 64, 3, 1),
            (2, 64, 12, 16, 128, 3, 3), (3, 128, 15, 20, 128, 3, 3), (7, 64, 6, 12, 64, 3, 1), (2, 256, 15, 20, 64, 1, 3), (3, 128, 24, 32, 40, 3, 3)]:
    line, ok, _ = run(*shp)
    bad += not ok
    print(('OK  ' if ok else 'BAD ') + line, flush=True)
if len(sys.argv) > 1:
    tot = [0.0] * 4
    for shp in [(32, 64, 120, 160, 64, 3, 1), (32, 64, 120, 160, 64, 1, 3), (32, 128, 60, 80, 128, 3, 1), (32, 128, 60, 80, 128, 1, 3),
                (32, 256, 30, 40, 256, 3, 1), (32, 256, 30, 40, 256, 1, 3), (32, 512, 15, 20, 512, 3, 1), (32, 512, 15, 20, 512, 1, 3),
                (32, 128, 60, 80, 128, 3, 3), (32, 128, 30, 40, 128, 1, 3)]:
        line, ok, tms = run(*shp, timing=True)
        bad += not ok
        tot = [a + b for a, b in zip(tot, tms)]
        print(('OK  ' if ok else 'BAD ') + line, flush=True)
    print(f'sum of the ten shapes: mask F(4,3) {tot[0]:.0f} us, F(2,3) {tot[1]:.0f} us | plain F(4,3) {tot[2]:.0f}, F(2,3) {tot[3]:.0f}')
sys.exit(1 if bad else 0)
