"""conv_wino.hip against torch (fp64 truth) and against the operand-ring kernels: correctness on awkward shapes, then
the ten encoder / decoder shapes at batch 32 timed back to back (same-box A/B)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
p = lambda t: None if t is None else t.data_ptr()


def geom(N, Ci, H, W, Co, KH, KW):
    return L.ConvGeom(N, Ci, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Ci)


def run(N, Ci, H, W, Co, KH, KW, timing=False):
    g = geom(N, Ci, H, W, Co, KH, KW)
    assert lib.dynmm_conv2d_wino_supported(C.byref(g), 0), (N, Ci, H, W, Co, KH, KW)
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, KH, KW, device='cuda') * (2.0 / (Ci * KH * KW)) ** 0.5
    b = torch.randn(Co, device='cuda'); res = torch.randn(N, Co, H, W, device='cuda')
    dy = torch.randn(N, Co, H, W, device='cuda'); mask = torch.randn(N, Ci, H, W, device='cuda'); acc = torch.randn(N, Ci, H, W, device='cuda')
    nf = lib.dynmm_wino_packed_floats(Co, Ci, KH, KW)
    uf = torch.empty(nf, device='cuda'); ud = torch.empty(nf, device='cuda')
    L.check(lib.dynmm_wino_pack(p(w), p(uf), None, Co, Ci, KH, KW, 0, st), 'pack f')
    L.check(lib.dynmm_wino_pack(p(w), p(ud), None, Co, Ci, KH, KW, 1, st), 'pack d')
    y = torch.full((N, Co, H, W), float('nan'), device='cuda'); dx = torch.full((N, Ci, H, W), float('nan'), device='cuda')
    L.check(lib.dynmm_conv2d_wino_fwd(p(x), p(uf), p(b), p(res), p(y), C.byref(g), 1, st), 'wino fwd')
    L.check(lib.dynmm_conv2d_wino_dgrad(p(dy), p(ud), p(mask), p(acc), p(dx), C.byref(g), st), 'wino dgrad')
    torch.cuda.synchronize()
    yr = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=(KH // 2, KW // 2)) + res.double())
    dxr = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=(KH // 2, KW // 2)) * (mask > 0) + acc.double()
    ef = ((y.double() - yr).abs().max() / yr.abs().max()).item(); ed = ((dx.double() - dxr).abs().max() / dxr.abs().max()).item()
    # the operand-ring kernels on the same problem
    wp = torch.empty(lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 0), device='cuda'); wd = torch.empty(lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 1), device='cuda')
    lib.dynmm_pack_weight(p(w), p(wp), p(wd), Co, Ci, KH, KW, st)
    y5 = torch.empty_like(y); dx5 = torch.empty_like(dx)
    L.check(lib.dynmm_conv2d_fwd(p(x), None, p(wp), None, p(b), p(res), p(y5), C.byref(g), 1, st), 'v5 fwd')
    L.check(lib.dynmm_conv2d_dgrad(p(dy), p(wd), p(mask), p(acc), p(dx5), None, C.byref(g), st), 'v5 dgrad')
    torch.cuda.synchronize()
    e5f = ((y5.double() - yr).abs().max() / yr.abs().max()).item(); e5d = ((dx5.double() - dxr).abs().max() / dxr.abs().max()).item()
    line = f'{(N, Ci, H, W, Co, KH, KW)}: wino fwd {ef:.2e} dgrad {ed:.2e} | ring fwd {e5f:.2e} dgrad {e5d:.2e}'
    ok = ef < 3e-6 and ed < 3e-6
    if timing:
        def tm(fn, n=20):
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
        fl = 2.0 * N * H * W * KH * KW * Ci * Co
        tw = tm(lambda: lib.dynmm_conv2d_wino_dgrad(p(dy), p(ud), p(mask), p(acc), p(dx), C.byref(g), st))
        t5 = tm(lambda: lib.dynmm_conv2d_dgrad(p(dy), p(wd), p(mask), p(acc), p(dx5), None, C.byref(g), st))
        twf = tm(lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(uf), p(b), None, p(y), C.byref(g), 1, st))
        t5f = tm(lambda: lib.dynmm_conv2d_fwd(p(x), None, p(wp), None, p(b), None, p(y5), C.byref(g), 1, st))
        line += f' | dgrad wino {tw:.1f} us ({fl / tw / 1e6:.0f} TF alg) ring {t5:.1f} us ({fl / t5 / 1e6:.0f}) | fwd wino {twf:.1f} ring {t5f:.1f}'
        return line, ok, (tw, t5, twf, t5f)
    return line, ok, None


bad = 0
for shp in [(3, 128, 15, 20, 128, 1, 3), (5, 64, 17, 20, 64, 3, 1), (2, 128, 9, 16, 256, 3, 1), (2, 192, 8, 24, 64, 1, 3),
            (2, 64, 12, 16, 128, 3, 3), (3, 128, 15, 20, 128, 3, 3), (7, 64, 6, 12, 64, 3, 1), (2, 256, 15, 20, 64, 1, 3)]:
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
    print(f'sum of the ten shapes: dgrad wino {tot[0]:.0f} us, ring {tot[1]:.0f} us | fwd wino {tot[2]:.0f} us, ring {tot[3]:.0f} us')
sys.exit(1 if bad else 0)
