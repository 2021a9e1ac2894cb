"""The product library's Winograd forward / input-gradient kernels in isolation on the encoder shapes (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
p = lambda t: None if t is None else t.data_ptr()
torch.manual_seed(0)
for (N, Cc, H, W, KH, KW) in [(32, 128, 60, 80, 1, 3), (32, 128, 60, 80, 3, 1), (32, 256, 30, 40, 1, 3), (32, 256, 30, 40, 3, 1)]:
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda'); res = torch.randn_like(x); mask = torch.randn_like(x); y = torch.empty_like(x)
    nf = lib.dynmm_wino_packed_floats(Cc, Cc, KH, KW)
    uf = torch.empty(nf, device='cuda'); ud = torch.empty(nf, device='cuda')
    lib.dynmm_wino_pack(p(w), p(uf), None, Cc, Cc, KH, KW, 0, st); lib.dynmm_wino_pack(p(w), p(ud), None, Cc, Cc, KH, KW, 1, st)
    n43 = lib.dynmm_wino43_packed_floats(Cc, Cc, KH, KW); u43 = torch.empty(n43, device='cuda')
    if KW == 3:
        lib.dynmm_wino43_pack(p(w), p(u43), Cc, Cc, KH, KW, st)
    for _ in range(4):
        L.check(lib.dynmm_conv2d_wino_fwd(p(x), p(uf), p(b), None, p(y), C.byref(g), 1, st), 'fwd')
        L.check(lib.dynmm_conv2d_wino_dgrad(p(x), p(ud), p(mask), None, p(y), C.byref(g), st), 'dgrad')
        if KW == 3:
            L.check(lib.dynmm_conv2d_wino43_dgrad(p(x), p(u43), p(mask), None, p(y), C.byref(g), st), 'dgrad43')
    torch.cuda.synchronize()
print('done')
