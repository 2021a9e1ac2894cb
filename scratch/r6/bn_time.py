"""BatchNorm normalise / backward-apply passes through the C ABI: us per launch and TB/s at the encoder shapes (batch 32), product
library against an alternative build; outputs compared bit for bit.  python scratch/r6/bn_time.py <libA> <libB> ..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
v, ci, cf = C.c_void_p, C.c_int, C.c_float
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
TAGS = sys.argv[1:]


def load(path):
    lib = C.CDLL(os.path.join(ROOT, path))
    lib.dynmm_bn_apply.argtypes = [v] * 11 + [ci, ci, ci, cf, cf, ci, ci, v, v]
    lib.dynmm_bn_bwd_apply.argtypes = [v] * 12 + [ci, ci, ci, ci, ci, v, v]
    lib.dynmm_bn_relu_bits_words.restype = C.c_size_t
    lib.dynmm_bn_relu_bits_words.argtypes = [ci, ci, ci]
    return lib


def tm(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000


libs = [load(t) for t in TAGS]
_w = torch.randn(4096, 4096, device='cuda')
for _ in range(200): _w = (_w @ _w).clamp_(-1, 1)
torch.cuda.synchronize()
torch.manual_seed(0)
for (N, Cc, H, W) in [(32, 64, 120, 160), (32, 128, 60, 80), (32, 256, 30, 40), (32, 512, 15, 20), (3, 64, 24, 36)]:
    HW = H * W
    x = torch.randn(N, Cc, H, W, device='cuda'); res = torch.randn_like(x); g = torch.randn_like(x)
    sums = torch.stack([x.double().sum((0, 2, 3)), (x.double() ** 2).sum((0, 2, 3))]).contiguous()
    gam = torch.randn(Cc, device='cuda'); bet = torch.randn(Cc, device='cuda')
    nb = (N * Cc * HW * 4) / 1e6
    for name in ('apply relu', 'apply +res+bits', 'bwd relu(remask)', 'bwd bits+dres'):
        ts, outs = [], []
        for lib in libs:
            rm, rv = torch.zeros(Cc, device='cuda'), torch.ones(Cc, device='cuda')
            sm, si = torch.empty(Cc, device='cuda'), torch.empty(Cc, device='cuda')
            y = torch.full_like(x, float('nan'))
            words = lib.dynmm_bn_relu_bits_words(N, Cc, HW)
            bits = torch.zeros(words, device='cuda', dtype=torch.int64)
            lib.dynmm_bn_apply(p(x), p(sums), p(gam), p(bet), p(rm), p(rv), p(sm), p(si), p(res), p(y), None, N, Cc, HW, 1e-3, 0.1, 1, 1, p(bits), st)
            bsum = torch.randn(2 * Cc, device='cuda', dtype=torch.float64)
            dx = torch.full_like(x, float('nan')); dres = torch.full_like(x, float('nan'))
            dg, db = torch.empty(Cc, device='cuda'), torch.empty(Cc, device='cuda')
            if name == 'apply relu':
                call = lambda: lib.dynmm_bn_apply(p(x), p(sums), p(gam), p(bet), p(rm), p(rv), p(sm), p(si), None, p(y), None, N, Cc, HW, 1e-3, 0.1, 1, 1, None, st)
                out, mb = y, 2 * nb
            elif name == 'apply +res+bits':
                call = lambda: lib.dynmm_bn_apply(p(x), p(sums), p(gam), p(bet), p(rm), p(rv), p(sm), p(si), p(res), p(y), None, N, Cc, HW, 1e-3, 0.1, 1, 1, p(bits), st)
                out, mb = y, 3 * nb
            elif name == 'bwd relu(remask)':
                call = lambda: lib.dynmm_bn_bwd_apply(p(g), None, p(x), p(sm), p(si), p(gam), p(bet), p(bsum), p(dx), None, p(dg), p(db), N, Cc, HW, 1, 1, None, st)
                out, mb = dx, 3 * nb
            else:
                call = lambda: lib.dynmm_bn_bwd_apply(p(g), None, p(x), p(sm), p(si), p(gam), p(bet), p(bsum), p(dx), p(dres), p(dg), p(db), N, Cc, HW, 1, 1, p(bits), st)
                out, mb = dx, 4 * nb
            r = call(); assert r == 0, (name, r)
            torch.cuda.synchronize()
            outs.append((out.clone(), bits.clone(), dres.clone()))
            ts.append(tm(call))
        same = [bool(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) and torch.equal(outs[0][2].nan_to_num(), o[2].nan_to_num())) for o in outs]
        print(f'{(N, Cc, H, W)} | {name:18s} | ' + ' | '.join(f'{t:7.1f}' for t in ts) + '   (' + ' '.join(f'{mb / t:.2f}' for t in ts) + f' TB/s)  equal {same}', flush=True)
        assert all(same)
