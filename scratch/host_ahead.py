"""How far ahead of the GPU does the host run?  Per step: time for step() to RETURN (enqueue only) vs the GPU's step time;
with gc enabled / disabled / frozen."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
sys.argv = ['bench.py', '--no-cpu-baseline']
args = bench.parse(); args.gpus = 1
dev = torch.device('cuda', 0)
step, ts, model = bench.train_workload(args, dev, 0, 1)
for _ in range(4): step()
torch.cuda.synchronize()
def run(n=10):
    host = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); step(); host.append((time.perf_counter() - a) * 1e3)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) * 1e3 / n
    return el, host
for mode in ('gc on', 'gc off', 'gc frozen', 'gc on'):
    if mode == 'gc off': gc.disable()
    elif mode == 'gc frozen': gc.enable(); gc.collect(); gc.freeze()
    else: gc.enable()
    el, host = run()
    print(f'{mode:10s} step {el:6.2f} ms   host enqueue per step: ' + ' '.join(f'{h:5.1f}' for h in host), flush=True)
print('gc counts', gc.get_count(), 'threshold', gc.get_threshold())
