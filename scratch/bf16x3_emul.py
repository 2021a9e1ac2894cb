"""Would a 3xbf16 split-precision MFMA path meet the 1e-3 logits bar?  Emulate it in the CPU oracle:
every conv computes conv(xh,wh)+conv(xh,wl)+conv(xl,wh) with hi/lo bf16 operands and fp32 accumulate."""
import sys
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
from dynmm_amd import synth
from tests import helpers as Hh
from oracle import dynmm_oracle as O

def split(t):
    hi = t.bfloat16().float()
    lo = (t - hi).bfloat16().float()
    return hi, lo

orig = F.conv2d
mode = {'v': 'fp32'}
def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if mode['v'] == 'fp32' or groups != 1:
        return orig(x, w, b, stride, padding, dilation, groups)
    xh, xl = split(x); wh, wl = split(w)
    if mode['v'] == 'bf16x3':
        y = orig(xh, wh, None, stride, padding, dilation, groups) + orig(xh, wl, None, stride, padding, dilation, groups) + orig(xl, wh, None, stride, padding, dilation, groups)
    elif mode['v'] == 'bf16x2':   # hi*hi + lo*hi + hi*lo without lo rounding... here: only 2 products
        y = orig(xh, wh, None, stride, padding, dilation, groups) + orig(xl, wh, None, stride, padding, dilation, groups)
    else:  # bf16x1
        y = orig(xh, wh, None, stride, padding, dilation, groups)
    return y if b is None else y + b.view(1, -1, 1, 1)
F.conv2d = conv
for (h, w, n) in ((96, 128, 3), (480, 640, 1)):
    sd = Hh.filled_state_dict(Hh.CFGS['P_se'], seed=0)
    rgb, depth = synth.synth_inputs(n, h, w, seed=5)
    res = {}
    for m in ('fp32', 'bf16x3', 'bf16x2', 'bf16x1'):
        mode['v'] = m
        with torch.no_grad():
            res[m] = O.forward(sd, rgb, depth, Hh.CFGS['P_se'], test=True, hard_gate=False)
    for m in ('bf16x3', 'bf16x2', 'bf16x1'):
        print(h, w, m, 'eval logits rel err vs fp32:', Hh.rel_err(res[m], res['fp32']))
    if h == 96:
        for m in ('fp32', 'bf16x3'):
            mode['v'] = m
            sd2 = {k: v.clone() for k, v in sd.items()}
            outs, lf = O.forward(sd2, rgb, depth, Hh.CFGS['P_se'], training=True)
            res['t' + m] = outs[0].detach()
        print('train logits rel err bf16x3 vs fp32:', Hh.rel_err(res['tbf16x3'], res['tfp32']))
