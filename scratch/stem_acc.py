import os, sys, torch
import torch.nn.functional as F
from dynmm_amd import ops
torch.manual_seed(0)
for ci in (3, 1):
    x = torch.randn(2, ci, 480, 640)
    w = torch.randn(64, ci, 7, 7) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    with torch.no_grad():
        y = ops.conv2d(x.cuda(), w.cuda(), None, 2, 3).cpu().double()
    y32 = F.conv2d(x, w, None, 2, 3).double()
    print('stem ci', ci, 'hip rms err', ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), 'cpu fp32', ((y32 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
r, d = torch.randn(2, 64, 120, 160), torch.randn(2, 64, 120, 160)
w = torch.randn(8, 128, 5, 5) * 0.02
b = torch.randn(8) * 0.1
ref = F.conv2d(torch.cat([r, d], 1).double(), w.double(), b.double(), 2, 0)
with torch.no_grad():
    y = ops.conv2d(r.cuda(), w.cuda(), b.cuda(), 2, 0, x2=d.cuda()).cpu().double()
y32 = F.conv2d(torch.cat([r, d], 1), w, b, 2, 0).double()
print('gate hip rms err', ((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item(), 'cpu fp32', ((y32 - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
