"""conv_igemm_v8.hip against the operand-ring kernel: bit identity on awkward shapes, then timing on the training shapes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
p = lambda t: None if t is None else t.data_ptr()


def run(N, Ci, H, W, Co, KH, KW, timing=False):
    g = L.ConvGeom(N, Ci, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Ci)
    x = torch.randn(N, Ci, H, W, device='cuda'); w = torch.randn(Co, Ci, KH, KW, device='cuda') * (2.0 / (Ci * KH * KW)) ** 0.5
    b = torch.randn(Co, device='cuda'); res = torch.randn(N, Co, H, W, device='cuda')
    wp = torch.empty(lib.dynmm_packed_weight_floats(Co, Ci, KH, KW, 0), device='cuda')
    lib.dynmm_pack_weight(p(w), p(wp), None, Co, Ci, KH, KW, st)
    outs = []
    for mode in (0, 1):
        lib.dynmm_debug_set_igemm_v8(mode)
        y = torch.full((N, Co, H, W), float('nan'), device='cuda')
        L.check(lib.dynmm_conv2d_fwd(p(x), None, p(wp), None, p(b), p(res), p(y), C.byref(g), 1, st), 'fwd')
        outs.append(y)
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    line = f'{(N, Ci, H, W, Co, KH, KW)}: identical {same} maxdiff {(outs[0] - outs[1]).abs().max().item():.2e}'
    if timing:
        def tm(n=20):
            fn = lambda: lib.dynmm_conv2d_fwd(p(x), None, p(wp), None, p(b), None, p(outs[0]), C.byref(g), 1, st)
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
        fl = 2.0 * N * H * W * KH * KW * Ci * Co
        lib.dynmm_debug_set_igemm_v8(0); t5 = tm()
        lib.dynmm_debug_set_igemm_v8(1); t8 = tm()
        line += f' | ring {t5:.1f} us ({fl / t5 / 1e6:.0f} TF)  v8 {t8:.1f} us ({fl / t8 / 1e6:.0f} TF)'
        return line, same, (t5, t8)
    return line, same, None


bad = 0
for shp in [(3, 128, 15, 20, 128, 1, 3), (5, 64, 17, 20, 64, 3, 1), (2, 128, 9, 16, 256, 3, 1), (2, 192, 8, 24, 64, 1, 3), (2, 96, 8, 24, 64, 1, 3),
            (2, 64, 12, 16, 128, 3, 3), (3, 128, 15, 20, 128, 3, 3), (7, 64, 6, 12, 64, 3, 1), (2, 256, 15, 20, 64, 1, 1), (1, 64, 8, 8, 128, 1, 3)]:
    line, ok, _ = run(*shp)
    bad += not ok
    print(('OK  ' if ok else 'BAD ') + line, flush=True)
if len(sys.argv) > 1:
    tot = [0.0, 0.0]
    for shp in [(32, 64, 120, 160, 64, 3, 1), (32, 64, 120, 160, 64, 1, 3), (32, 128, 60, 80, 128, 3, 1), (32, 128, 60, 80, 128, 1, 3),
                (32, 256, 30, 40, 256, 3, 1), (32, 256, 30, 40, 256, 1, 3), (32, 512, 15, 20, 512, 3, 1), (32, 512, 15, 20, 512, 1, 3),
                (32, 128, 60, 80, 128, 3, 3), (32, 128, 30, 40, 128, 1, 3), (32, 64, 120, 160, 128, 1, 1)]:
        line, ok, tms = run(*shp, timing=True)
        bad += not ok
        tot = [a + b for a, b in zip(tot, tms)]
        print(('OK  ' if ok else 'BAD ') + line, flush=True)
    print(f'sum: ring {tot[0]:.0f} us, v8 {tot[1]:.0f} us')
lib.dynmm_debug_set_igemm_v8(-1)
sys.exit(1 if bad else 0)
