import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops, synth, lib as L
from dynmm_amd.nn.blocks import ResNetEncoder
lib = L.load()
hip = C.CDLL('libamdhip64.so')
real_fwd, real_dg, real_wg = lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad, lib.dynmm_conv2d_wgrad
def grab(ptr, n):
    t = torch.empty(n, device='cuda')
    p = ptr if isinstance(ptr, int) else ptr.value
    hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(p), C.c_size_t(4 * n), 3)
    return t.cpu()
class Wrap(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.e = ResNetEncoder('resnet34', 'NonBottleneck1D', 1)
    def forward(self, x):
        y = ops.max_pool_3x3_s2(self.e.forward_first_conv(x))
        return self.e.forward_layer2(self.e.forward_layer1(y))
def run(mode):
    lib.dynmm_debug_set_igemm_v5(mode)
    log = []
    def dg(dy, wd, mask, accum, dx, dx2, g, st):
        go = g._obj
        nin, nout = go.N * go.Co * go.Ho * go.Wo, go.N * go.Ci * go.H * go.W
        torch.cuda.synchronize()
        ins = {'dy': grab(dy, nin), 'mask': grab(mask, nout) if mask else None, 'accum': grab(accum, nout) if accum else None}
        rc = real_dg(dy, wd, mask, accum, dx, dx2, g, st)
        torch.cuda.synchronize()
        log.append((f'dgrad Ci{go.Ci} {go.H}x{go.W} Co{go.Co} k{go.KH}x{go.KW} s{go.SH}{go.SW}', ins, grab(dx, nout)))
        return rc
    def wg(x, x2, dy, dw, db, ws, wsb, g, st):
        go = g._obj
        torch.cuda.synchronize()
        ins = {'x': grab(x, go.N * go.Ci * go.H * go.W), 'dy': grab(dy, go.N * go.Co * go.Ho * go.Wo)}
        rc = real_wg(x, x2, dy, dw, db, ws, wsb, g, st)
        torch.cuda.synchronize()
        log.append((f'wgrad Ci{go.Ci} {go.H}x{go.W} Co{go.Co} k{go.KH}x{go.KW} s{go.SH}{go.SW}', ins, grab(dw, go.Co * go.Ci * go.KH * go.KW)))
        return rc
    lib.dynmm_conv2d_dgrad, lib.dynmm_conv2d_wgrad = dg, wg
    m = Wrap(); del m.e.layer3, m.e.layer4
    synth.fill_state_dict(m.state_dict(), seed=3)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 96, 128, generator=g).cuda().requires_grad_(True)
    y = m(x)
    y.backward(torch.randn(y.shape, generator=g).cuda())
    torch.cuda.synchronize()
    return log
a, b = run(0), run(1)
def rd(u, v):
    return float((u - v).abs().max() / (u.abs().max() + 1e-30))
for (na, ia, oa), (nb, ib, ob) in list(zip(a, b))[:14]:
    assert na == nb
    print(na, ' inputs:', {k: (None if ia[k] is None else f'{rd(ia[k], ib[k]):.1e}') for k in ia}, ' output:', f'{rd(oa, ob):.1e}')
(na, ia, oa), (nb, ib, ob) = a[0], b[0]
d = (oa - ob).abs()
idx = (d > 1e-3 * oa.abs().max()).nonzero().flatten()
print('differing elements', len(idx), 'of', oa.numel())
for i in idx[:10].tolist():
    print(i, 'old out', float(oa[i]), 'v5 out', float(ob[i]), 'mask old', float(ia['mask'][i]), 'mask v5', float(ib['mask'][i]))
mo, mv = ia['mask'], ib['mask']
print('mask zeros old/v5', int((mo == 0).sum()), int((mv == 0).sum()), 'sign mismatches', int(((mo > 0) != (mv > 0)).sum()))
