import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
lib.dynmm_debug_set_igemm_v5(1)
for (N, Cc, H, W, KH, KW) in [(2, 128, 12, 16, 1, 3), (2, 128, 12, 16, 3, 1), (2, 64, 24, 32, 1, 3)]:
    n = N * Cc * H * W
    G = 1 << 16
    def guarded():
        b = torch.full((n + 2 * G,), 7.0, device='cuda'); return b, b[G:G + n]
    bd, dy = guarded(); dy.normal_()
    bm, mask = guarded(); mask.normal_()
    ba, acc = guarded(); acc.normal_()
    bx, dx = guarded()
    w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    wp = torch.full((KH * KW * Cc * Cc + 2 * G,), 7.0, device='cuda'); wdb = torch.full((KH * KW * Cc * Cc + 2 * G,), 7.0, device='cuda')
    lib.dynmm_pack_weight(w.data_ptr(), wp[G:].data_ptr(), wdb[G:].data_ptr(), Cc, Cc, KH, KW, st)
    snap = [t.clone() for t in (bd, bm, ba, wdb)]
    rc = lib.dynmm_conv2d_dgrad(dy.data_ptr(), wdb[G:].data_ptr(), mask.data_ptr(), acc.data_ptr(), dx.data_ptr(), None, C.byref(g), st)
    torch.cuda.synchronize()
    print((N, Cc, H, W, KH, KW), 'rc', rc, 'inputs intact:', [bool(torch.equal(a, b)) for a, b in zip(snap, (bd, bm, ba, wdb))],
          'dx guards intact:', bool((bx[:G] == 7).all() and (bx[G + n:] == 7).all()), 'dx all written:', bool((dx != 7).all()))
