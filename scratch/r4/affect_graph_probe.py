"""Bisect a capture failure of the multi-stream ModalityDynMM step: python affect_graph_probe.py BATCH [train|eval] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from dynmm_amd.nn import affect as A  # noqa: E402

batch = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else 'train'
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device('cuda', 0)
A.BRANCH_STREAMS = os.environ.get('STREAMS', '1') == '1'
torch.manual_seed(0)
model = A.DynMMNetV2(1.0, False, freeze=False).to(dev)
g = torch.Generator().manual_seed(7)
xs = [torch.randn(batch, 50, f, generator=g).to(dev) for f in (35, 74, 300)]
inputs = [xs, [torch.full((batch,), 50, dtype=torch.long)] * 3]
y = torch.randn(batch, 1, generator=g).to(dev)
step = A.AffectTrainStep(model, lr=1e-5, weight_decay=1e-4, lossw=0.1, use_graph=os.environ.get('GRAPH', '1') == '1')
if os.environ.get('WGRAD_GROUP'):
    step.wgrad_group = int(os.environ['WGRAD_GROUP'])       # (A/B: 4 = the library's default group)
model.train(mode == 'train')
for _ in range(3):
    out = step(inputs, y)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(steps):
    out = step(inputs, y)
torch.cuda.synchronize()
el = time.perf_counter() - t
print(f'batch {batch} {mode} streams={int(A.BRANCH_STREAMS)} graph={int(step.use_graph)}: {1000 * el / steps:.3f} ms/step, '
      f'{batch * steps / el:.0f} samples/s, total {out["total"].item():.6f}', flush=True)
