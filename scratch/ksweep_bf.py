import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import torch.nn.functional as F
from dynmm_amd import ops, lib as L
lib = L.load()
N = 32
st = torch.cuda.current_stream().cuda_stream
for (Cc, H, W) in ((128, 60, 80), (256, 30, 40), (512, 15, 20), (64, 120, 160)):
    x = torch.randn(N, Cc, H, W, device='cuda')
    b = torch.zeros(Cc, device='cuda')
    y = torch.empty_like(x); y2 = torch.empty_like(x)
    for KH in (3,):
        w = torch.randn(Cc, Cc, KH, 1, device='cuda') * (3 * Cc) ** -0.5
        g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, 1, 1, 1, KH // 2, 0, Cc)
        nel = KH * Cc * Cc
        fh = torch.empty(nel, device='cuda', dtype=torch.int16); fl = torch.empty_like(fh)
        lib.dynmm_pack_weight_bf16x3(w.data_ptr(), fh.data_ptr(), fl.data_ptr(), None, None, Cc, Cc, KH, 1, st)
        wp = torch.empty(nel, device='cuda')
        lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), None, Cc, Cc, KH, 1, st)
        def run_bf():
            return lib.dynmm_conv2d_fwd_bf16x3(x.data_ptr(), fh.data_ptr(), fl.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), C.byref(g), 0, st)
        def run_32():
            return lib.dynmm_conv2d_fwd(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), None, y2.data_ptr(), C.byref(g), 0, st)
        assert run_bf() == 0 and run_32() == 0
        torch.cuda.synchronize()
        ref = F.conv2d(x[:2].double().cpu(), w.double().cpu(), None, 1, (KH // 2, 0))
        e_bf = ((y[:2].double().cpu() - ref).abs().max() / ref.abs().max()).item()
        e_32 = ((y2[:2].double().cpu() - ref).abs().max() / ref.abs().max()).item()
        res = []
        for fn in (run_bf, run_32):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1000)
        fl_ = 2.0 * N * H * W * KH * Cc * Cc
        print(f'TPIX={os.environ.get("DYNMM_BF16_TPIX","128")} C={Cc} KH={KH}: bf16x3 {res[0]:.1f} us ({fl_/res[0]/1e6:.1f} TF-equiv, err {e_bf:.2e}) | fp32 {res[1]:.1f} us ({fl_/res[1]/1e6:.1f} TF, err {e_32:.2e})')
