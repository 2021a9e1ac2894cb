import ctypes as C, os, sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for (N, Cc, H, W, KH, KW) in [(2, 128, 12, 16, 3, 1), (2, 128, 12, 16, 1, 3), (2, 64, 24, 32, 3, 1)]:
    dy = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    mask = torch.randn(N, Cc, H, W, device='cuda'); acc = torch.randn(N, Cc, H, W, device='cuda')
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    wp = torch.empty(KH * KW * Cc * Cc, device='cuda'); wd = torch.empty(KH * KW * Cc * Cc, device='cuda')
    lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), wd.data_ptr(), Cc, Cc, KH, KW, st)
    ref0 = F.conv_transpose2d(dy, w, padding=(KH // 2, KW // 2))
    for name, m, a_ in (('plain', None, None), ('mask', mask, None), ('accum', None, acc), ('both', mask, acc)):
        dx = torch.empty_like(dy)
        rc = lib.dynmm_conv2d_dgrad(dy.data_ptr(), wd.data_ptr(), m.data_ptr() if m is not None else None,
                                    a_.data_ptr() if a_ is not None else None, dx.data_ptr(), None, C.byref(g), st)
        ref = ref0 * ((m > 0).float() if m is not None else 1.0) + (a_ if a_ is not None else 0.0)
        print(N, Cc, H, W, KH, KW, name, rc, float((dx - ref).abs().max() / ref.abs().max()))
