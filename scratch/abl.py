import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops
N = 32
for (C, H, W) in ((128, 60, 80), (256, 30, 40), (512, 15, 20)):
    x = torch.randn(N, C, H, W, device='cuda')
    w = torch.randn(C, C, 3, 1, device='cuda') * 0.05
    b = torch.zeros(C, device='cuda')
    with torch.no_grad():
        for _ in range(3): y = ops.conv2d(x, w, b, 1, (1, 0), None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y = ops.conv2d(x, w, b, 1, (1, 0), None)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * N * H * W * 3 * C * C
    print(f'ABL={os.environ.get("DYNMM_ABLATE","0")} C={C}: {ms*1000:.1f} us  {fl/ms/1e9:.1f} TF (incl. pack kernel)')
