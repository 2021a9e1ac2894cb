"""micro-benchmark of the fused up-sampling + CE tail (csrc/tail.hip) at the benchmark shape."""
import sys
import torch
from dynmm_amd import lib as L
from dynmm_amd.ops import _p

import os
if os.environ.get("DYNMM_LIB"):
    L.LIB_PATH = os.environ["DYNMM_LIB"]
lib = L.load()
N, C, H, W = 32, 40, 240, 320
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = 'cuda'
x = torch.randn(N, C, H, W, device=dev)
w = torch.randn(C, 9, device=dev) * 0.3
b = torch.randn(C, device=dev) * 0.1
t = torch.randint(0, C + 1, (N, 2 * H, 2 * W), device=dev, dtype=torch.uint8)
cw = torch.rand(C, device=dev) + 0.5
lse = torch.empty(N, 2 * H, 2 * W, device=dev)
acc = torch.zeros(2, device=dev, dtype=torch.float64)
gs = torch.full((1,), 1e-7, device=dev)
dx = torch.empty_like(x)
dw, db = torch.empty(C, 9, device=dev), torch.empty(C, device=dev)
ws = torch.empty(lib.dynmm_up2ce_bwd_workspace_bytes(N, C, H, W) // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream


def fwd():
    L.check(lib.dynmm_up2ce_fwd(_p(x), _p(w), _p(b), t.data_ptr(), _p(cw), _p(lse), acc.data_ptr(), N, C, H, W, 0, st), 'f')


def bwd():
    L.check(lib.dynmm_up2ce_bwd(_p(x), _p(w), _p(b), t.data_ptr(), _p(cw), _p(lse), _p(gs), _p(dx), _p(dw), _p(db), _p(ws),
                                N, C, H, W, st), 'b')


for name, f in (('fwd', fwd), ('bwd', bwd)):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / reps * 1e3, 1), 'us')
