"""cProfile of the host side of one inference forward (configs[1], batch 16) and one training step: where the enqueue time goes."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
sys.argv = ['bench.py', '--no-cpu-baseline']
args = bench.parse(); args.gpus = 1
dev = torch.device('cuda', 0)
step, model = bench.fwd_workload(args, dev, 16, 'all4', True)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'inference batch 16: host enqueue {(t1 - t0) * 100:.2f} ms per forward, wall {(t2 - t0) * 100:.2f} ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
