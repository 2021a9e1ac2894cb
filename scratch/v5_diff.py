"""Find the first tensor that differs between the two implicit-GEMM kernel generations inside a model-level pass.
   DYNMM_IGEMM_V5=0 python scratch/v5_diff.py /tmp/a.pt ; DYNMM_IGEMM_V5=1 python scratch/v5_diff.py /tmp/b.pt ; python scratch/v5_diff.py /tmp/a.pt /tmp/b.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    for (ka, va), (kb, vb) in zip(a, b):
        assert ka == kb, (ka, kb)
        d = float((va - vb).abs().max() / (va.abs().max() + 1e-30))
        print(f'{ka:60s} {tuple(va.shape)} rel diff {d:.3e}' + ('   <<<<' if d > 1e-4 else ''))
    sys.exit(0)
from dynmm_amd import ops, synth, lib as L
from dynmm_amd.nn.blocks import ResNetEncoder
import ctypes as C
rec = []
lib = L.load()
orig_fwd, orig_dg = lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad
class Wrap(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.e = ResNetEncoder('resnet34', 'NonBottleneck1D', 1)
    def forward(self, x):
        y = ops.max_pool_3x3_s2(self.e.forward_first_conv(x))
        return self.e.forward_layer2(self.e.forward_layer1(y))
m = Wrap(); del m.e.layer3, m.e.layer4
synth.fill_state_dict(m.state_dict(), seed=3)
m = m.cuda().train()
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 96, 128, generator=g).cuda().requires_grad_(True)
# record every tensor autograd saves / produces through hooks on module outputs
def hook(name):
    def f(mod, inp, out):
        if torch.is_tensor(out):
            rec.append((name + '.out', out.detach().cpu().clone()))
            if out.requires_grad:
                out.register_hook(lambda gr, n=name: rec.append((n + '.gout', gr.detach().cpu().clone())))
    return f
for n, mod in m.named_modules():
    if n:
        mod.register_forward_hook(hook(n))
y = m(x)
gy = torch.randn(y.shape, generator=g).cuda()
y.backward(gy)
torch.cuda.synchronize()
rec.append(('dx', x.grad.cpu()))
for n, p in m.named_parameters():
    rec.append(('grad.' + n, p.grad.detach().cpu()))
torch.save(rec, sys.argv[1])
print(len(rec), 'tensors')
