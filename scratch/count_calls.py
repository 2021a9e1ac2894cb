"""How many BatchNorm backward reductions does one training step still launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dynmm_amd import ops, lib as L
sys.argv = ['bench.py', '--no-cpu-baseline']
args = bench.parse(); args.gpus = 1
step, ts, model = bench.train_workload(args, torch.device('cuda', 0), 0, 1)
for _ in range(2): step()
torch.cuda.synchronize()
handle = L.load()
counts = {}
class Counting:
    def __init__(self, inner): self.inner = inner
    def __getattr__(self, name):
        f = getattr(self.inner, name)
        if name in ('dynmm_bn_bwd_reduce', 'dynmm_conv2d_dgrad_bnstats', 'dynmm_bn_stats_from_partials', 'dynmm_bn_bwd_apply', 'dynmm_conv2d_dgrad_ws'):
            def g(*a, **k):
                counts[name] = counts.get(name, 0) + 1
                return f(*a, **k)
            return g
        return f
orig = ops._lib
ops._lib = lambda: Counting(handle)
step(); torch.cuda.synchronize()
ops._lib = orig
print(counts)
