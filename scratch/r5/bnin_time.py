"""Isolated cost of normalising on load (scratch; needs scratch/r5/bnin_async_apply.patch applied and the library rebuilt):
dynmm_conv2d_wino_fwd on z against dynmm_conv2d_wino_fwd_bnin on c, encoder shapes at batch 32, 3x1 taps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import ops, lib as L
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
for (N, Cc, H, W) in [(32, 64, 120, 160), (32, 128, 60, 80), (32, 256, 30, 40), (32, 512, 15, 20)]:
    c = torch.randn(N, Cc, H, W, device='cuda')
    w = torch.randn(Cc, Cc, 3, 1, device='cuda') * (3 * Cc) ** -0.5
    b = torch.randn(Cc, device='cuda') * 0.1
    gamma, beta = torch.rand(Cc, device='cuda') + 0.5, torch.randn(Cc, device='cuda') * 0.3
    sums = torch.zeros(2, Cc, device='cuda', dtype=torch.float64)
    L.check(lib.dynmm_bn_stats(ops._p(c), ops._p(sums), N, Cc, H * W, 1, st), 'bn_stats')
    g = ops._geom(c, None, w, (1, 1), (1, 0))
    ut = torch.empty(lib.dynmm_wino_packed_floats(Cc, Cc, 3, 1), device='cuda')
    L.check(lib.dynmm_wino_pack(ops._p(w), ops._p(ut), None, Cc, Cc, 3, 1, 0, st), 'pack')
    z = torch.empty_like(c); mean = torch.empty(Cc, device='cuda'); invstd = torch.empty(Cc, device='cuda')
    y = torch.empty_like(c)
    def apply():
        L.check(lib.dynmm_bn_apply(ops._p(c), ops._p(sums), ops._p(gamma), ops._p(beta), None, None, ops._p(mean), ops._p(invstd), None,
                                   ops._p(z), None, N, Cc, H * W, 1e-3, 0.1, 1, L.ACT_RELU, None, st), 'bn_apply')
    def plain():
        L.check(lib.dynmm_conv2d_wino_fwd(ops._p(z), ops._p(ut), ops._p(b), None, ops._p(y), C.byref(g), L.ACT_RELU, st), 'fwd')
    def bnin():
        L.check(lib.dynmm_conv2d_wino_fwd_bnin(ops._p(c), ops._p(ut), ops._p(b), ops._p(sums), 1, ops._p(gamma), ops._p(beta), 1e-3,
                                               ops._p(y), C.byref(g), L.ACT_RELU, st), 'bnin')
    apply()
    res = {}
    for name, fn in (('bn_apply', apply), ('plain', plain), ('bnin', bnin), ('plain', plain), ('bnin', bnin)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(1000 * e0.elapsed_time(e1) / 20)
    print((N, Cc, H, W), {k: [round(v, 1) for v in vs] for k, vs in res.items()}, flush=True)
