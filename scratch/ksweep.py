import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from dynmm_amd import ops, lib as L
lib = L.load()
N = 32
st = torch.cuda.current_stream().cuda_stream
for (Cc, H, W) in ((128, 60, 80), (256, 30, 40), (512, 15, 20)):
    x = torch.randn(N, Cc, H, W, device='cuda')
    b = torch.zeros(Cc, device='cuda')
    y = torch.empty_like(x)
    for KH in (1, 3):
        w = torch.randn(Cc, Cc, KH, 1, device='cuda') * 0.05
        g = L.ConvGeom(N, Cc, H, W, Cc, H, W, KH, 1, 1, 1, KH // 2, 0, Cc)
        wp = torch.empty(KH * Cc * Cc, device='cuda')
        lib.dynmm_pack_weight(w.data_ptr(), wp.data_ptr(), None, Cc, Cc, KH, 1, st)
        def run():
            lib.dynmm_conv2d_fwd(x.data_ptr(), None, wp.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), C.byref(g), 0, st)
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1000
        fl = 2.0 * N * H * W * KH * Cc * Cc
        print(f'ABL={os.environ.get("DYNMM_ABLATE","0")} C={Cc} KH={KH} nk={KH*Cc//16}: {us:.1f} us  {fl/us/1e6:.1f} TF')
