"""prove or kill the hypothesis that the 17 % gap between configs[3] dense and configs[2] is the
depth-encoder stream being a different torch pool stream for every model built in a process.

    python scratch/r6/stream_lottery.py pool      # round-5 behaviour: every model takes torch.cuda.Stream() (next pool stream)
    python scratch/r6/stream_lottery.py plan      # round-6: ops.side_stream() singleton
    python scratch/r6/stream_lottery.py sweep     # ONE model, depth stream := torch pool stream #k, k = 0..7, then the plan's
"""
import os
import sys
import time
import json

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd import engine, ops, synth                    # noqa: E402
import bench                                                # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'plan'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
N, H, W = 32, 480, 640
rgb, depth, labels = bench.make_batch(N, H, W, dev, 1234)
cw = np.linspace(0.5, 2.0, 40)


def build(hard):
    m = bench.make_model('P', H, W, dev).train()
    m.temp, m.hard_gate = 1.0, hard
    ts = engine.TrainStep(m, cw, lr=1e-4, momentum=0.9, weight_decay=1e-4, loss_ratio=1.0, flop_budget=0.0)
    return m, ts


def time_steps(ts, k=steps, warm=3):
    for _ in range(warm):
        ts(rgb, depth, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        ts(rgb, depth, labels)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


real_side = ops.side_stream
out = []
if mode in ('pool', 'plan'):
    for i in range(4):
        m, ts = build(hard=False)
        if mode == 'pool':
            s = torch.cuda.Stream()                         # what nn/net.py did per model instance in round 5
            ops.side_stream = lambda s=s: s
        ms = time_steps(ts)
        side = ops.side_stream()
        out.append({'model': i, 'ms': round(ms, 2), 'side': hex(side.cuda_stream), 'census': ts.census if mode == 'plan' else None})
        print(out[-1], flush=True)
        ts.reducer.remove_hooks()
        del m, ts
        torch.cuda.empty_cache()
elif mode == 'sweep':
    m, ts = build(hard=False)
    pool = [torch.cuda.Stream() for _ in range(8)]
    order = [None] + list(range(8)) + [None]
    for k in order:
        if k is None:
            ops.side_stream = real_side
        else:
            ops.side_stream = lambda s=pool[k]: s
        try:
            ms = time_steps(ts, warm=2)
        except Exception as e:                              # the census refuses a fifth stream only for plan streams it knows
            ms = repr(e)
        out.append({'side': 'plan' if k is None else f'pool[{k}]', 'handle': hex(ops.side_stream().cuda_stream), 'ms': ms if isinstance(ms, str) else round(ms, 2)})
        print(out[-1], flush=True)
print(json.dumps({'mode': mode, 'rows': out}))
