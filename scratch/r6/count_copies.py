"""Where do the ~190 __amd_rocclr_copyBuffer dispatches per step come from?  torch.profiler on one eager step: every op that launched
a Memcpy DtoD, with its Python call site."""
import os, sys, collections
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd import engine, ops
import bench
dev = torch.device('cuda:0')
N, H, W = 32, 480, 640
rgb, depth, labels = bench.make_batch(N, H, W, dev, 1234)
m = bench.make_model('P', H, W, dev).train()
m.temp, m.hard_gate = 1.0, False
ts = engine.TrainStep(m, np.linspace(0.5, 2.0, 40), lr=1e-4, momentum=0.9, weight_decay=1e-4, loss_ratio=1.0, flop_budget=0.0)
for _ in range(3):
    ts(rgb, depth, labels)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts(rgb, depth, labels)
    torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    if e.device_type.name == 'CUDA' or 'emcpy' in e.name or 'copy' in e.name.lower():
        cnt[e.name[:80]] += 1
for k, v in cnt.most_common(25):
    print(v, k)
# aten ops that led to copies, with stacks
stacks = collections.Counter()
for e in ev:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::zero_', 'aten::fill_', 'aten::zeros', 'aten::empty_like'):
        st = [s for s in (e.stack or []) if 'dynmm_amd' in s or 'bench.py' in s]
        stacks[(e.name, st[0] if st else '?')] += 1
for (n, s), v in stacks.most_common(40):
    print(v, n, s[-110:])
