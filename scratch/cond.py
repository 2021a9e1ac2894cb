import sys, os
sys.path.insert(0, '.')
import torch
from dynmm_amd import synth
from tests import helpers as Hh
from oracle import dynmm_oracle as O
cfg=sys.argv[1]; h,w,n = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rgb, depth = synth.synth_inputs(n, h, w, seed=99)
res={}
for dt in (torch.float32, torch.float64):
    sd = Hh.filled_state_dict(Hh.CFGS[cfg], seed=5)
    sd = {k:(v.to(dt) if v.dtype.is_floating_point else v) for k,v in sd.items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
    outs, lf = O.forward(sd, rgb.to(dt), depth.to(dt), Hh.CFGS[cfg], training=True, temp=0.7)
    tot = 3.0*lf
    for i,o in enumerate(outs): tot = tot + (o*Hh.grad_probe(tuple(o.shape), f's{i}').to(dt)).mean()
    tot.backward()
    res[dt]=(outs, {k:p.grad for k,p in params.items()})
o32,g32=res[torch.float32]; o64,g64=res[torch.float64]
print('logits fp32 vs fp64', [Hh.rel_err(a.detach(), b.detach()) for a,b in zip(o32,o64)])
errs=sorted([(Hh.rel_err(g32[k], g64[k]), k) for k in g32 if g64[k].abs().max()>1e-6], reverse=True)
print(errs[:6]); import numpy as np; print('median', np.median([e for e,_ in errs]))
