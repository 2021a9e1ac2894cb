"""Micro-benchmark of the weight-gradient kernel through the C ABI: median time per launch (HIP events on the launch
stream) for the encoder shapes, with the knobs of conv_igemm.hip set from the environment by the caller.

    python scratch/wgrad_micro.py [iters]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dynmm_amd import lib as L  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib = L.load()
SHAPES = [  # N, Ci, H, W, Co, KH, KW
    (32, 128, 60, 80, 128, 3, 1), (32, 128, 60, 80, 128, 1, 3), (32, 256, 30, 40, 256, 3, 1),
    (32, 256, 30, 40, 256, 1, 3), (32, 512, 15, 20, 512, 3, 1), (32, 512, 15, 20, 512, 1, 3),
    (32, 64, 120, 160, 64, 3, 1), (32, 64, 120, 160, 64, 1, 3), (32, 128, 120, 160, 128, 3, 3),
]
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for (N, Ci, H, W, Co, KH, KW) in SHAPES:
    g = L.ConvGeom(N, Ci, H, W, Co, H, W, KH, KW, 1, 1, KH // 2, KW // 2, Ci)
    x = torch.randn(N, Ci, H, W, device='cuda')
    dy = torch.randn(N, Co, H, W, device='cuda')
    dw = torch.empty(Co, Ci, KH, KW, device='cuda')
    db = torch.empty(Co, device='cuda')
    nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g))
    ws = torch.empty(max(nbytes // 4, 1), device='cuda')
    ts = []
    for i in range(iters + 3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dynmm_conv2d_wgrad(x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                       nbytes, C.byref(g), st), 'wgrad')
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    fl = 2.0 * N * H * W * KH * KW * Ci * Co
    ref = torch.nn.grad.conv2d_weight(x, dw.shape, dy, padding=(KH // 2, KW // 2)) if i < 0 else None
    tot += med
    print(f'{(N, Ci, H, W, Co, KH, KW)}  {med:7.1f} us  {fl / med / 1e6:6.1f} TFLOP/s  (min {ts[0]:.1f})')
print(f'sum of medians {tot:.1f} us')
