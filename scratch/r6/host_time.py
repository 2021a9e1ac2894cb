"""How long does the HOST take to enqueue one step (GPU idle at the start, no synchronisation inside)?  If that is close to the
step time the step is host-bound on that box and Python overhead matters; if it is half, it does not."""
import os, sys, time
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
for _ in range(4):
    ts(rgb, depth, labels)
torch.cuda.synchronize()
host, total = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts(rgb, depth, labels)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print('host enqueue ms per step', [round(v, 1) for v in host], 'step from idle ms', [round(v, 1) for v in total])
# forward only / backward only split of the host time
import torch.autograd.profiler as prof
fw, bw = [], []
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with engine.direct_gradients(True), ts._prepacked():
        ops.touched_reset(); ts.reducer.zero()
        m.decoder.defer_tail = True
        outs, lf = m(rgb, depth)
        m.decoder.defer_tail = False
        t1 = time.perf_counter()
        last = ops.multi_scale_loss_backward(outs, [t for t in labels], ts.cw, lf, 1.0, 0.0)
        ops.join_async()
        t2 = time.perf_counter()
    torch.cuda.synchronize()
    fw.append((t1 - t0) * 1e3); bw.append((t2 - t1) * 1e3)
print('host ms forward', [round(v, 1) for v in fw], 'loss + backward', [round(v, 1) for v in bw])
