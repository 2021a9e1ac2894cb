"""One rank, RCCL process group, the bucket all-reduces FORCED (DYNMM_DP_FORCE_COLLECTIVES) — what each exchange arrangement of
dp.GradBucketReducer costs the step when the collective itself is trivial: python scratch/r6/dp_exchange_ab.py <plain|wgrad|depth|comm>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
if mode != 'plain':
    os.environ['DYNMM_DP_FORCE_COLLECTIVES'] = '1'
import torch
import torch.distributed as dist
import bench
sys.argv = ['bench.py', '--no-cpu-baseline'] + ([] if mode == 'plain' else ['--dp-exchange', mode])
args = bench.parse(); args.gpus = 1
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
if mode != 'plain':
    dist.init_process_group('nccl', rank=0, world_size=1)
step, ts, model = bench.train_workload(args, dev, 0, 1)
for _ in range(4): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(12): step()
torch.cuda.synchronize()
print(f'{mode}: {(time.perf_counter() - t0) / 12 * 1e3:.3f} ms per step; buckets launched in backward: '
      f'{getattr(ts.reducer, "launched_in_backward", None)} of {len(ts.reducer.buckets)}; census {ts.census}', flush=True)
if mode != 'plain':
    dist.destroy_process_group()
