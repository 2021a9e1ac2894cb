"""K-split operand-ring launches: result vs the un-split launch (same kernel, no workspace) and run-to-run bit-reproducibility."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import lib as L
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for (N, Cc, H, W, KH, KW) in [(32, 512, 15, 20, 3, 1), (32, 512, 15, 20, 1, 3), (16, 256, 30, 40, 3, 1), (32, 128, 15, 20, 3, 3), (6, 512, 15, 20, 1, 3), (32, 128, 30, 40, 1, 3)]:
    x = torch.randn(N, Cc, H, W, device='cuda'); w = torch.randn(Cc, Cc, KH, KW, device='cuda') * 0.05
    b = torch.randn(Cc, device='cuda'); res = torch.randn(N, Cc, H, W, device='cuda')
    g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Cc)
    wp = torch.empty(KH * KW * Cc * Cc, device='cuda'); wd = torch.empty(KH * KW * Cc * Cc, device='cuda')
    lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), wd.data_ptr(), Cc, Cc, KH, KW, st)
    nws = lib.dynmm_conv2d_workspace_bytes(C.byref(g), 0)
    ws = torch.empty(max(nws // 4, 1), device='cuda')
    outs = []
    for use in (False, True, True):
        y = torch.empty_like(x)
        rc = lib.dynmm_conv2d_fwd_ws(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), res.data_ptr(), y.data_ptr(), C.byref(g), 1,
                                     ws.data_ptr() if use else None, nws if use else 0, st)
        assert rc == 0
        outs.append(y)
    torch.cuda.synchronize()
    def t(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1000
    y = torch.empty_like(x)
    t0 = t(lambda: lib.dynmm_conv2d_fwd_ws(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), res.data_ptr(), y.data_ptr(), C.byref(g), 1, None, 0, st))
    t1 = t(lambda: lib.dynmm_conv2d_fwd_ws(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), res.data_ptr(), y.data_ptr(), C.byref(g), 1, ws.data_ptr(), nws, st))
    fl = 2.0 * N * H * W * KH * KW * Cc * Cc
    print((N, Cc, H, W, KH, KW), 'ws MB', round(nws / 1e6, 1), 'split-vs-unsplit rel', float((outs[0] - outs[1]).abs().max() / outs[0].abs().max()),
          'reproducible', bool(torch.equal(outs[1], outs[2])), f'unsplit {t0:.1f} us ({fl / t0 / 1e6:.0f} TF) split {t1:.1f} us ({fl / t1 / 1e6:.0f} TF)')
