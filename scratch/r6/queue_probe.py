"""Can a cheap probe tell that two streams share a hardware queue?  For the caller's stream + the plan's streams + torch pool
streams 0..7: all-pairs concurrency of two spin kernels (torch.cuda._sleep), then the step time with that pool stream as the
depth-encoder stream (the r6a sweep) to compare."""
import os, sys, time, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dynmm_amd import engine, ops
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(0)
plan = ops.stream_plan()
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(8)]
CYC = 400000          # ~0.2 ms


def pair_time(a, b, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for s in (a, b):
            if s is not main:
                s.wait_stream(main)
        for s in (a, b):
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)
        for s in (a, b):
            if s is not main:
                main.wait_stream(s)
        e1.record(main)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


torch.cuda._sleep(CYC); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(CYC); e1.record(); torch.cuda.synchronize()
single = e0.elapsed_time(e1)
print('single spin', round(single, 3), 'ms')
names = {'main': main, 'side': plan.side, 'w0': plan.wgrad[0], 'w1': plan.wgrad[1]}
for i, s in enumerate(pool):
    names[f'pool{i}'] = s
keys = list(names)
print('pair concurrency (time of two spins / one spin; ~1 = concurrent, ~2 = serialised)')
for i, a in enumerate(keys):
    row = []
    for b in keys[i + 1:]:
        row.append(f'{b}:{pair_time(names[a], names[b]) / single:.2f}')
    print(a, ' '.join(row), flush=True)
