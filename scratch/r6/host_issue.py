"""How long does the HOST take to issue one training step of the headline workload, and where does it spend it?

    python scratch/r6/host_issue.py [steps]

Prints (a) the wall time per step with a synchronisation after every step, (b) the time Python needs to ISSUE a step when it is
never made to wait (a queue of steps is issued, one synchronisation at the end), and (c) the 25 functions with the largest own
time of a cProfile over 3 issued steps.  If (b) is close to (a) the step is co-bound by the host and every microsecond of Python
per launch shows up as idle GPU time in the regions of short kernels (profiles/r06_exposed_3stream.txt: 2.2 ms of idle)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = [sys.argv[0]] + ['--no-cpu-baseline', '--no-extra']
import torch  # noqa: E402
import bench  # noqa: E402

steps = 8
args = bench.parse()
dev = torch.device('cuda', 0)
step, ts, model = bench.train_workload(args, dev, 0, 1)
for _ in range(4):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
    torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
t0 = time.perf_counter()
for _ in range(steps):
    step()
issue = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / steps
print(f'wall per step (sync each) {1e3 * wall:.2f} ms | host issue per step {1e3 * issue:.2f} ms | {steps} queued steps {1e3 * total:.2f} ms per step')
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(25)
