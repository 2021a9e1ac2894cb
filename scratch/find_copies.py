"""Which Python lines issue device copies during one training step?  (aten::copy_ / clone / _to_copy by caller)"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynmm_amd import engine
from dynmm_amd.nn.net import SkipGateESANet
from dynmm_amd import synth
torch.manual_seed(0)
m = SkipGateESANet(height=96, width=128, pretrained_on_imagenet=False).cuda().train()
step = engine.TrainStep(m, np.ones(40), lr=0.01, momentum=0.9, weight_decay=1e-4)
rgb = torch.randn(2, 3, 96, 128, device='cuda'); depth = torch.randn(2, 1, 96, 128, device='cuda')
lab = synth.synth_labels(2, 96, 128, seed=1, device='cuda')
import torch.nn.functional as F
tg = [lab] + [F.interpolate(lab[:, None].float(), scale_factor=1 / r, mode='nearest')[:, 0].to(lab.dtype) for r in (8, 16, 32)]
for _ in range(2): step(rgb, depth, tg)
torch.cuda.synchronize()
cnt = collections.Counter()
orig_copy, orig_clone = torch.Tensor.copy_, torch.Tensor.clone
def where():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'dynmm_amd' in fr.filename: return f'{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}'
    return 'outside'
def copy_(self, *a, **k): cnt['copy_ ' + where()] += 1; return orig_copy(self, *a, **k)
def clone(self, *a, **k): cnt['clone ' + where()] += 1; return orig_clone(self, *a, **k)
torch.Tensor.copy_, torch.Tensor.clone = copy_, clone
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(rgb, depth, tg); torch.cuda.synchronize()
torch.Tensor.copy_, torch.Tensor.clone = orig_copy, orig_clone
for k, v in cnt.most_common(10): print(v, k)
ev = collections.Counter()
for e in prof.events():
    if 'copy' in e.name.lower() or 'Memcpy' in e.name or 'clone' in e.name: ev[e.name] += 1
print(ev.most_common(12))
# who calls aten::copy_ : parent op names
par = collections.Counter()
for e in prof.events():
    if e.name == 'aten::copy_':
        p = e.cpu_parent
        while p is not None and p.name.startswith('aten::'): p = p.cpu_parent
        par[p.name if p is not None else 'top'] += 1
print(par.most_common(12))
