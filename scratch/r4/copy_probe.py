"""Which host-side operations of a train step enqueue copy kernels?  python scratch/r4/copy_probe.py"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import bench  # noqa: E402

sys.argv = ['bench.py', '--no-cpu-baseline', '--no-extra', '--no-kernel-timing']
args = bench.parse()
dev = torch.device('cuda', 0)
step, ts, model = bench.train_workload(args, dev, 0, 1, False, args.branches, False)
for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
names = Counter()
for e in ev:
    n = e.name
    if 'memcpy' in n.lower() or 'copyBuffer' in n or 'Memcpy' in n or 'Memset' in n:
        names[n] += 1
print(names.most_common(10))
# CPU-side ops that launched copies: aten::copy_ / aten::to / aten::clone with their python stacks
stacks = Counter()
for e in ev:
    if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::clone', 'aten::fill_', 'aten::zero_'):
        st = [s for s in (e.stack or []) if 'dynmm_amd' in s or 'bench.py' in s]
        stacks[(e.name, st[0] if st else '?')] += 1
for k, v in stacks.most_common(25):
    print(v, k)
