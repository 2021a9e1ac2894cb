import numpy as np, torch
from dynmm_amd import engine
from tests.test_hip_model import hip_model
torch.manual_seed(0)
h, w, n = 96, 128, 3
rgb, depth = torch.randn(n, 3, h, w).cuda(), torch.randn(n, 1, h, w).cuda()
labels = [torch.randint(0, 41, (n, h // s, w // s), dtype=torch.uint8).cuda() for s in (1, 8, 16, 32)]
cw = np.linspace(0.5, 1.5, 40).astype(np.float32)
out = {}
for fused in (True, False, 'again'):
    m = hip_model('P_se', h, w, seed=3); m.train(); m.temp, m.hard_gate = 1.0, False
    step = engine.TrainStep(m, cw, lr=0.0, loss_ratio=1e-3, fuse_tail=bool(fused is True))
    step._body(rgb, depth, labels); torch.cuda.synchronize()
    out[fused] = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
gmax = max(g.abs().max().item() for g in out[False].values())
for a, b in ((True, False), ('again', False)):
    rows = []
    for k, ga in out[a].items():
        gb = out[b][k]
        rows.append((((ga - gb).abs().max() / max(gb.abs().max().item(), 1e-4 * gmax)).item(), k, gb.abs().max().item()))
    rows.sort(reverse=True)
    print(a, 'gmax', gmax)
    for r in rows[:8]: print('  %.3e %s %.3e' % r)
    va = torch.cat([g.flatten() for g in out[a].values()]).double(); vb = torch.cat([g.flatten() for g in out[b].values()]).double()
    print('  cos-1', (torch.nn.functional.cosine_similarity(va, vb, dim=0) - 1).item(), 'norm ratio', (va.norm() / vb.norm()).item())
