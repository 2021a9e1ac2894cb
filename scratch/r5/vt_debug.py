import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from dynmm_amd import lib as L, ops
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for (N, Ci, H, W, Co) in [(1, 64, 4, 32, 64), (1, 64, 8, 32, 64), (2, 64, 4, 32, 64), (1, 64, 24, 32, 64), (2, 64, 24, 32, 64), (2, 64, 24, 16, 64), (1, 64, 24, 64, 64)]:
    x = torch.randn(N, Ci, H, W, device='cuda'); dy = torch.randn(N, Co, H, W, device='cuda')
    w0 = torch.empty(Co, Ci, 3, 1)
    g = ops._geom(x, None, w0, (1, 1), (1, 0))
    wd = torch.zeros(Co, Ci, 3, 1, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wd, None, 1, (1, 0)).backward(dy.double().cpu())
    dw = torch.full((Co, Ci, 3, 1), float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    nbytes = lib.dynmm_conv2d_wgrad_workspace_bytes(C.byref(g)); ws = torch.empty(max(nbytes // 4, 1), device='cuda')
    L.check(lib.dynmm_conv2d_wgrad(x.data_ptr(), None, dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nbytes, C.byref(g), st), 'wgrad')
    torch.cuda.synchronize()
    ref = wd.grad
    err = (dw.double().cpu() - ref).abs()
    bad_ci = (err[:, :, 1, 0].max(0).values > 1e-3).nonzero().flatten().tolist()
    bad_co = (err[:, :, 1, 0].max(1).values > 1e-3).nonzero().flatten().tolist()
    print((N, Ci, H, W, Co), 'max err per tap', ['%.2e' % err[:, :, r].max().item() for r in range(3)], 'bad ci', bad_ci[:12], len(bad_ci), 'bad co', len(bad_co), 'ws', nbytes)
