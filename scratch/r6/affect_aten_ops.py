"""Which ATen kernels does one eager ModalityDynMM train step launch, and from where?  python scratch/r6/affect_aten_ops.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from dynmm_amd.nn import affect as A  # noqa: E402

batch = 128
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = A.DynMMNetV2(1.0, False, freeze=False).to(dev).train()
g = torch.Generator().manual_seed(7)
xs = [torch.randn(batch, 50, f, generator=g).to(dev) for f in (35, 74, 300)]
inputs = [xs, [torch.full((batch,), 50, dtype=torch.long)] * 3]
y = torch.randn(batch, 1, generator=g).to(dev)
step = A.AffectTrainStep(model, lr=1e-5, weight_decay=1e-4, lossw=0.1, use_graph=False)
for _ in range(2):
    step(inputs, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(inputs, y)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith('aten::') and e.key in (
    'aten::add', 'aten::add_', 'aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::zeros',
    'aten::zeros_like', 'aten::cat', 'aten::mul', 'aten::sum')]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print(f'{e.count:5d} {e.key:18s}', ' <- '.join(s.split('/')[-1] for s in e.stack[:5]))
