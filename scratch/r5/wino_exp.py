"""Bound-splitting experiments on the F(2,3) pair kernel (scratch/r5/wino_exp.hip = csrc/conv_wino.hip + -DEXP=k):
  0 as shipped | 1 no epilogue | 2 no MFMA (one VALU fma instead) | 3 every tile stages tile 0's pixels (L2-resident B)
  4 no data transform | 5 no LDS fragment reads.   Prints us per launch for the encoder shapes at batch 32."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
HERE = os.path.dirname(os.path.abspath(__file__))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
GP = C.POINTER(L.ConvGeom)
v = C.c_void_p


def load(k):
    lib = C.CDLL(os.path.join(HERE, f'libwino_exp{k}.so'))
    lib.exp_wino_packed_floats.restype = C.c_size_t
    lib.exp_wino_packed_floats.argtypes = [C.c_int] * 4
    lib.exp_wino_pack.argtypes = [v, v, v, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.exp_conv2d_wino_fwd.argtypes = [v, v, v, v, v, GP, C.c_int, v]
    lib.exp_conv2d_wino_dgrad.argtypes = [v, v, v, v, v, GP, v]
    return lib


def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000


SHAPES = [(32, 64, 120, 160, 1, 3), (32, 64, 120, 160, 3, 1), (32, 128, 60, 80, 1, 3), (32, 128, 60, 80, 3, 1),
          (32, 256, 30, 40, 1, 3), (32, 256, 30, 40, 3, 1), (32, 512, 15, 20, 1, 3), (32, 512, 15, 20, 3, 1)]
KS = [int(k) for k in os.environ.get('EXPS', '0 1 2 3 4 5').split()]
libs = {k: load(k) for k in KS}
ONLY = os.environ.get('ONLY')
torch.manual_seed(0)
print('shape | pass | ' + ' | '.join(f'EXP{k}' for k in KS) + '   (us; alg TF/s of EXP0)')
for (N, Cc, H, W, KH, KW) in SHAPES:
    if ONLY == 'h' and KW != 3: continue
    if ONLY == 'v' and KW != 1: continue
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda'); res = torch.randn_like(x); mask = torch.randn_like(x); y = torch.empty_like(x)
    nf = libs[0].exp_wino_packed_floats(Cc, Cc, KH, KW)
    uf = torch.empty(nf, device='cuda'); ud = torch.empty(nf, device='cuda')
    libs[0].exp_wino_pack(p(w), p(uf), None, Cc, Cc, KH, KW, 0, st); libs[0].exp_wino_pack(p(w), p(ud), None, Cc, Cc, KH, KW, 1, st)
    fl = 2.0 * N * H * W * 3 * Cc * Cc
    for name, call in (('fwd bias+relu', lambda lib: lib.exp_conv2d_wino_fwd(p(x), p(uf), p(b), None, p(y), C.byref(g), 1, st)),
                       ('fwd +residual', lambda lib: lib.exp_conv2d_wino_fwd(p(x), p(uf), p(b), p(res), p(y), C.byref(g), 1, st)),
                       ('dgrad plain', lambda lib: lib.exp_conv2d_wino_dgrad(p(x), p(ud), None, None, p(y), C.byref(g), st)),
                       ('dgrad mask', lambda lib: lib.exp_conv2d_wino_dgrad(p(x), p(ud), p(mask), None, p(y), C.byref(g), st)),
                       ('dgrad mask+acc', lambda lib: lib.exp_conv2d_wino_dgrad(p(x), p(ud), p(mask), p(res), p(y), C.byref(g), st))):
        ts = []
        for k in KS:
            r = call(libs[k])
            assert r == 0, (name, k, r)
            ts.append(tm(lambda: call(libs[k])))
        print(f'{(N, Cc, H, W, KH, KW)} | {name:14s} | ' + ' | '.join(f'{t:7.1f}' for t in ts) + f'   ({fl / ts[0] / 1e6:.0f})', flush=True)
