"""Weight gradient of three-tap convs through the C ABI: error vs an fp64 einsum, bias gradient, run-to-run bits, time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
shapes = [(32, 128, 60, 80, 1, 3), (32, 128, 60, 80, 3, 1), (32, 256, 30, 40, 1, 3), (32, 256, 30, 40, 3, 1),
          (32, 512, 15, 20, 1, 3), (32, 512, 15, 20, 3, 1), (32, 64, 120, 160, 1, 3), (32, 64, 120, 160, 3, 1),
          (3, 128, 15, 20, 1, 3), (5, 64, 17, 16, 3, 1), (6, 128, 30, 40, 3, 1), (1, 64, 15, 20, 1, 3)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    shapes = shapes[8:]
for (N, Cc, H, W, KH, KW) in shapes:
    Co = Cc
    x = torch.randn(N, Cc, H, W, device='cuda'); dy = torch.randn(N, Co, H, W, device='cuda')
    g = L.ConvGeom(N, Cc, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    nws = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
    ws = torch.empty(max(nws // 4, 1), device='cuda')
    outs = []
    for _ in range(2):
        dw = torch.full((Co, Cc, KH, KW), float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
        rc = lib.dynmm_conv2d_wgrad(x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, C.byref(g), st)
        assert rc == 0, rc
        outs.append((dw, db))
    torch.cuda.synchronize()
    xd, dyd = x.double(), dy.double()
    ref = torch.empty(Co, Cc, KH, KW, dtype=torch.float64, device='cuda')
    xp = F.pad(xd, (KW // 2, KW // 2, KH // 2, KH // 2))
    for r in range(KH):
        for s in range(KW):
            ref[:, :, r, s] = torch.einsum('nkhw,nchw->kc', dyd, xp[:, :, r:r + H, s:s + W])
    err = float((outs[0][0].double() - ref).abs().max() / ref.abs().max())
    berr = float((outs[0][1].double() - dyd.sum((0, 2, 3))).abs().max() / dyd.sum((0, 2, 3)).abs().max())
    same = bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]))
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
    dw, db = outs[0]
    us = t(lambda: lib.dynmm_conv2d_wgrad(x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, C.byref(g), st))
    fl = 2.0 * N * H * W * KH * KW * Cc * Co
    print((N, Cc, H, W, KH, KW), f'rel err {err:.2e} bias {berr:.2e} reproducible {same}  {us:.1f} us  {fl / us / 1e6:.1f} TF (incl. slab reduction)', flush=True)
    if os.environ.get('GROUP'):
        n = int(os.environ['GROUP'])
        xs_t = [torch.randn_like(x) for _ in range(n)]; dys_t = [torch.randn_like(dy) for _ in range(n)]
        dws_t = [torch.empty_like(dw) for _ in range(n)]; dbs_t = [torch.empty_like(db) for _ in range(n)]
        nb = lib.dynmm_conv2d_wgrad_group_workspace_bytes(C.byref(g), n)
        wsg = torch.empty(max(nb // 4, 1), device='cuda')
        xs = (C.c_void_p * n)(*[v.data_ptr() for v in xs_t]); dys = (C.c_void_p * n)(*[v.data_ptr() for v in dys_t])
        dwa = (C.c_void_p * n)(*[v.data_ptr() for v in dws_t]); dba = (C.c_void_p * n)(*[v.data_ptr() for v in dbs_t])
        assert lib.dynmm_conv2d_wgrad_group(n, xs, dys, dwa, dba, wsg.data_ptr(), nb, C.byref(g), st) == 0
        torch.cuda.synchronize()
        # group result == single-problem result up to the split boundaries (different pixel ranges): compare with fp64
        xp = F.pad(xs_t[1].double(), (KW // 2, KW // 2, KH // 2, KH // 2))
        for r in range(KH):
            for s_ in range(KW):
                ref[:, :, r, s_] = torch.einsum('nkhw,nchw->kc', dys_t[1].double(), xp[:, :, r:r + H, s_:s_ + W])
        gerr = float((dws_t[1].double() - ref).abs().max() / ref.abs().max())
        gus = t(lambda: lib.dynmm_conv2d_wgrad_group(n, xs, dys, dwa, dba, wsg.data_ptr(), nb, C.byref(g), st))
        print(f'    group of {n}: rel err {gerr:.2e}  {gus:.1f} us = {gus / n:.1f} us per conv  {n * fl / gus / 1e6:.1f} TF', flush=True)
