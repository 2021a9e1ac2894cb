import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
HERE = os.path.dirname(os.path.abspath(__file__))
st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
GP = C.POINTER(L.ConvGeom); v = C.c_void_p
def load(tag):
    lib = C.CDLL(os.path.join(HERE, f'libwino_{tag}.so'))
    lib.dynmm_wino_packed_floats.restype = C.c_size_t
    lib.dynmm_wino_packed_floats.argtypes = [C.c_int] * 4
    lib.dynmm_wino_pack.argtypes = [v, v, v, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.dynmm_conv2d_wino_dgrad_bnred.argtypes = [v, v, v, v, v, v, v, v, v, GP, v]
    lib.dynmm_conv2d_wino_dgrad_bnred_slots.argtypes = [GP]
    return lib
libs = {t: load(t) for t in ('cur', 'o5')}
torch.manual_seed(0)
for (N, Cc, H, W) in [(32, 64, 120, 160), (32, 128, 60, 80), (4, 64, 24, 32)]:
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, 3, 1, 1, 1, 1, 0, Cc)
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, 3, 1, device='cuda') * 0.05
    c = torch.randn_like(x)
    mean = torch.randn(Cc, device='cuda'); invstd = torch.rand(Cc, device='cuda') + 0.5
    gam = torch.randn(Cc, device='cuda'); bet = torch.randn(Cc, device='cuda')
    outs = {}
    for t in ('cur', 'cur', 'o5', 'o5'):
        lib = libs[t]
        nf = lib.dynmm_wino_packed_floats(Cc, Cc, 3, 1)
        u = torch.empty(nf, device='cuda')
        assert lib.dynmm_wino_pack(p(w), p(u), None, Cc, Cc, 3, 1, 1, st) == 0
        ns = lib.dynmm_conv2d_wino_dgrad_bnred_slots(C.byref(g))
        sums = torch.zeros(ns * 2 * Cc, device='cuda', dtype=torch.float64)
        y = torch.full_like(x, float('nan'))
        r = lib.dynmm_conv2d_wino_dgrad_bnred(p(x), p(u), p(c), p(mean), p(invstd), p(gam), p(bet), p(sums), p(y), C.byref(g), st)
        torch.cuda.synchronize()
        outs.setdefault(t, []).append((y.clone(), sums.clone(), r))
    a0, a1 = outs['cur']; b0, b1 = outs['o5']
    print((N, Cc, H, W), 'rc', a0[2], b0[2], 'cur==cur', torch.equal(a0[0], a1[0]), 'o5==o5', torch.equal(b0[0], b1[0]),
          'cur==o5', torch.equal(a0[0], b0[0]), 'nan', int(torch.isnan(a0[0]).sum()), int(torch.isnan(b0[0]).sum()),
          'maxdiff', float((a0[0] - b0[0]).nan_to_num().abs().max()), 'ndiff', int((a0[0] != b0[0]).sum()),
          'sums rel', float(((a0[1] - b0[1]).abs().max() / a0[1].abs().max())))
    d = (a0[0] != b0[0]).nonzero()
    if len(d):
        print('  first diffs', d[:6].tolist(), 'cur', a0[0][tuple(d[0])].item(), 'o5', b0[0][tuple(d[0])].item())
