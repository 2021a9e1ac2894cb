"""the skip-model gate-gradient test under both implicit-GEMM generations + a per-call A/B."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynmm_amd import synth, lib as L
from tests import helpers as Hh
from tests.test_skip_esanet import _hip_skip
lib = L.load(); hip = C.CDLL('libamdhip64.so')
real_fwd, real_dg = lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad
def grab(ptr, n):
    t = torch.empty(n, device='cuda'); p = ptr if isinstance(ptr, int) else ptr.value
    hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(p), C.c_size_t(4 * n), 3); return t
AB = [False]
def gs(g):
    g = g._obj; return f'N{g.N} Ci{g.Ci} {g.H}x{g.W} Co{g.Co} k{g.KH}x{g.KW} s{g.SH}{g.SW} p{g.PH}{g.PW}'
def fwd(x, x2, wp, sc, sh, res, y, g, act, st):
    if not AB[0]: return real_fwd(x, x2, wp, sc, sh, res, y, g, act, st)
    go = g._obj; n = go.N * go.Co * go.Ho * go.Wo
    resc = grab(res, n) if res else None       # the residual may alias y
    lib.dynmm_debug_set_igemm_v5(0); ref = torch.empty(n, device='cuda')
    real_fwd(x, x2, wp, sc, sh, resc.data_ptr() if res else None, ref.data_ptr(), g, act, st)
    lib.dynmm_debug_set_igemm_v5(1); rc = real_fwd(x, x2, wp, sc, sh, res, y, g, act, st)
    torch.cuda.synchronize(); got = grab(y, n); d = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
    if d > 1e-5: print(f'DIFF fwd {gs(g)} act{act} res{int(bool(res))} alias{int(bool(res) and (res if isinstance(res,int) else res.value) == (y if isinstance(y,int) else y.value))} sc{int(bool(sc))} sh{int(bool(sh))} rel {d:.2e}')
    return rc
def dg(dy, wd, mask, accum, dx, dx2, g, st):
    if not AB[0]: return real_dg(dy, wd, mask, accum, dx, dx2, g, st)
    go = g._obj; n = go.N * go.Ci * go.H * go.W
    acc_c = grab(accum, n) if accum else None
    lib.dynmm_debug_set_igemm_v5(0); ref = torch.empty(n, device='cuda')
    real_dg(dy, wd, mask, acc_c.data_ptr() if accum else None, ref.data_ptr(), dx2, g, st)
    lib.dynmm_debug_set_igemm_v5(1); rc = real_dg(dy, wd, mask, accum, dx, dx2, g, st)
    torch.cuda.synchronize(); got = grab(dx, n); d = float((got - ref).abs().max() / (ref.abs().max() + 1e-30))
    if d > 1e-5: print(f'DIFF dgrad {gs(g)} mask{int(bool(mask))} accum{int(bool(accum))} rel {d:.2e}')
    return rc
lib.dynmm_conv2d_fwd, lib.dynmm_conv2d_dgrad = fwd, dg
h, w, n, temp = 96, 128, 2, 0.7
rgb, depth = synth.synth_inputs(n, h, w, seed=1234)
r = np.random.Generator(np.random.PCG64(11))
noise = [torch.from_numpy(r.exponential(size=(n, 2)).astype(np.float32)) for _ in range(4)]
def run(mode):
    lib.dynmm_debug_set_igemm_v5(mode)
    m = _hip_skip(temp, (2, 2, 2, 2)).eval(); m.freeze()
    m.gumbel_noise = [e.cuda() for e in noise]
    out = m(rgb.cuda(), depth.cuda())
    (out * Hh.grad_probe(tuple(out.shape), 's0').cuda()).mean().backward()
    torch.cuda.synchronize()
    return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
o0, g0 = run(0); o1, g1 = run(1)
print('out old-vs-v5', float((o0 - o1).abs().max() / o0.abs().max()))
for k in g0: print(k, float((g0[k] - g1[k]).abs().max() / (g0[k].abs().max() + 1e-30)))
AB[0] = True
run(1)
