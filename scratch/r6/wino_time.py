"""F(2,3) pair kernel variants (scratch/r6/libwino_<tag>.so = csrc/conv_wino.hip built stand-alone): us per launch of the horizontal
forward (plain, + residual, with BatchNorm statistics) at batch 32, and bit-equality with the first variant."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
HERE = os.path.dirname(os.path.abspath(__file__))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
GP = C.POINTER(L.ConvGeom)
v = C.c_void_p
TAGS = sys.argv[1:] or ['v0', 'v2']
VERT = os.environ.get('TAPS', 'h') == 'v'
KH, KW = (3, 1) if VERT else (1, 3)


def load(tag):
    lib = C.CDLL(os.path.join(HERE, f'libwino_{tag}.so'))
    lib.dynmm_wino_packed_floats.restype = C.c_size_t
    lib.dynmm_wino_packed_floats.argtypes = [C.c_int] * 4
    lib.dynmm_wino_pack.argtypes = [v, v, v, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.dynmm_conv2d_wino_fwd.argtypes = [v, v, v, v, v, GP, C.c_int, v]
    lib.dynmm_conv2d_wino_dgrad.argtypes = [v, v, v, v, v, GP, v]
    lib.dynmm_conv2d_wino_fwd_stats.argtypes = [v, v, v, v, v, C.c_int, GP, v]
    lib.dynmm_conv2d_wino_fwd_stats_slots.argtypes = [GP]
    lib.dynmm_conv2d_wino_dgrad_bnred.argtypes = [v, v, v, v, v, v, v, v, v, GP, v]
    lib.dynmm_conv2d_wino_dgrad_bnred_slots.argtypes = [GP]
    return lib


def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000


SHAPES = [(32, 64, 120, 160), (32, 128, 60, 80), (32, 256, 30, 40), (32, 512, 15, 20), (32, 128, 30, 40), (3, 64, 24, 36), (5, 128, 17, 20)]
libs = {t: load(t) for t in TAGS}
# WARM-UP: the first launches of a process run ~10 % slow (clock ramp): measured columns would carry an order bias
_w = torch.randn(4096, 4096, device='cuda')
for _ in range(300): _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
torch.manual_seed(0)
print('shape | pass | ' + ' | '.join(TAGS) + '   (us; alg TF/s of each)')
for (N, Cc, H, W) in SHAPES:
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda'); res = torch.randn_like(x); mask = torch.randn_like(x)
    mean = torch.randn(Cc, device='cuda'); invstd = torch.rand(Cc, device='cuda') + 0.5
    gam = torch.randn(Cc, device='cuda'); bet = torch.randn(Cc, device='cuda')
    nf = libs[TAGS[0]].dynmm_wino_packed_floats(Cc, Cc, KH, KW)
    fl = 2.0 * N * H * W * 3 * Cc * Cc
    for name in ('fwd relu', 'fwd +res', 'fwd stats', 'dgrad pl', 'dgrad m+a', 'dgrad bnred'):
        ts, outs = [], []
        for t in TAGS:
            lib = libs[t]
            u = torch.empty(nf, device='cuda')
            assert lib.dynmm_wino_pack(p(w), p(u), None, Cc, Cc, KH, KW, 1 if name.startswith('dgrad') else 0, st) == 0
            y = torch.full_like(x, float('nan'))
            if name == 'fwd relu':
                call = lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(u), p(b), None, p(y), C.byref(g), 1, st)
            elif name == 'fwd +res':
                call = lambda: lib.dynmm_conv2d_wino_fwd(p(x), p(u), p(b), p(res), p(y), C.byref(g), 1, st)
            elif name == 'fwd stats':
                ns = 0 if VERT else lib.dynmm_conv2d_wino_fwd_stats_slots(C.byref(g))
                if ns <= 0:
                    continue
                stats = torch.zeros(ns * 2 * Cc, device='cuda', dtype=torch.float64)
                call = lambda: lib.dynmm_conv2d_wino_fwd_stats(p(x), p(u), p(b), p(y), p(stats), ns, C.byref(g), st)
            elif name == 'dgrad bnred':
                if not VERT:
                    continue
                ns = lib.dynmm_conv2d_wino_dgrad_bnred_slots(C.byref(g))
                sums = torch.zeros(ns * 2 * Cc, device='cuda', dtype=torch.float64)
                call = lambda: lib.dynmm_conv2d_wino_dgrad_bnred(p(x), p(u), p(mask), p(mean), p(invstd), p(gam), p(bet), p(sums), p(y), C.byref(g), st)
            elif name == 'dgrad pl':
                call = lambda: lib.dynmm_conv2d_wino_dgrad(p(x), p(u), None, None, p(y), C.byref(g), st)
            else:
                call = lambda: lib.dynmm_conv2d_wino_dgrad(p(x), p(u), p(mask), p(res), p(y), C.byref(g), st)
            r = call()
            assert r == 0, (t, name, r)
            torch.cuda.synchronize()
            outs.append(y.clone())
            ts.append(tm(call))
        if not ts:
            continue
        same = [bool(torch.equal(outs[0], o)) for o in outs]
        print(f'{(N, Cc, H, W)} | {name:9s} | ' + ' | '.join(f'{t:7.1f}' for t in ts) + '   (' + ' '.join(f'{fl / t / 1e6:.0f}' for t in ts) + f')  equal {same}', flush=True)
        assert all(same), 'variant differs from the first'
