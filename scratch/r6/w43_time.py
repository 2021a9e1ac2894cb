"""F(4,3) input-gradient kernel variants (scratch/r6/libw43_<tag>.so = csrc/conv_wino43.hip built stand-alone):
us per launch at batch 32 for the step's shapes, and bit-equality of every variant with the first one."""
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


def load(tag):
    lib = C.CDLL(os.path.join(HERE, f'libw43_{tag}.so'))
    lib.dynmm_wino43_packed_floats.restype = C.c_size_t
    lib.dynmm_wino43_packed_floats.argtypes = [C.c_int] * 4
    lib.dynmm_wino43_pack.argtypes = [v, v, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.dynmm_conv2d_wino43_dgrad.argtypes = [v, v, v, v, v, GP, v]
    return lib


def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000


SHAPES = [(32, 64, 120, 160), (32, 128, 60, 80), (32, 256, 30, 40), (32, 512, 15, 20), (32, 128, 30, 40), (32, 128, 15, 20),
          (3, 64, 24, 36), (5, 128, 17, 20)]
libs = {t: load(t) for t in TAGS}
# WARM-UP: the first launches of a process run ~10 % slow (clock ramp): measured columns would carry an order bias
_w = torch.randn(4096, 4096, device='cuda')
for _ in range(300): _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
torch.manual_seed(0)
print('shape | epilogue | ' + ' | '.join(TAGS) + '   (us; alg TF/s of each)')
for (N, Cc, H, W) in SHAPES:
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, 1, 3, 1, 1, 0, 1, Cc)
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, 1, 3, device='cuda') * 0.05
    res = torch.randn_like(x); mask = torch.randn_like(x)
    nf = libs[TAGS[0]].dynmm_wino43_packed_floats(Cc, Cc, 1, 3)
    fl = 2.0 * N * H * W * 3 * Cc * Cc
    for name, m_, r_ in (('plain', None, None), ('mask+acc', mask, res)):
        ts, outs = [], []
        for t in TAGS:
            lib = libs[t]
            ud = torch.empty(nf, device='cuda')
            assert lib.dynmm_wino43_pack(p(w), p(ud), Cc, Cc, 1, 3, st) == 0
            y = torch.full_like(x, float('nan'))
            call = lambda: lib.dynmm_conv2d_wino43_dgrad(p(x), p(ud), p(m_), p(r_), p(y), C.byref(g), st)
            r = call()
            assert r == 0, (t, r)
            torch.cuda.synchronize()
            outs.append(y.clone())
            ts.append(tm(call))
        same = [bool(torch.equal(outs[0], o)) for o in outs]
        print(f'{(N, Cc, H, W)} | {name:8s} | ' + ' | '.join(f'{t:7.1f}' for t in ts) + '   (' + ' '.join(f'{fl / t / 1e6:.0f}' for t in ts) + f')  equal {same}', flush=True)
        assert all(same), 'variant differs from the first'
