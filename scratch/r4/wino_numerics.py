"""fp32 1-D Winograd F(2,3) against direct fp32 conv, both vs fp64 truth (error class of the arithmetic)."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
for C in (64, 128, 512):
    x = torch.randn(2, C, 16, 40)
    w = torch.randn(C, C, 1, 3) * (2.0 / (3 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=(0, 1))
    direct = F.conv2d(x, w, padding=(0, 1))
    g0, g1, g2 = w[..., 0, 0], w[..., 0, 1], w[..., 0, 2]
    U = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    xp = F.pad(x, (1, 1))
    d0, d1, d2, d3 = xp[..., 0:-3:2], xp[..., 1:-2:2], xp[..., 2:-1:2], xp[..., 3::2]
    V = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
    M = [torch.einsum('oc,nchw->nohw', u, v) for u, v in zip(U, V)]
    y0, y1 = M[0] + M[1] + M[2], M[1] - M[2] - M[3]
    y = torch.stack([y0, y1], -1).reshape(ref.shape)
    e = lambda a: ((a.double() - ref).abs().max() / ref.abs().max()).item()
    r = lambda a: ((a.double() - ref).norm() / ref.norm()).item()
    print(C, 'direct max/rms', f'{e(direct):.2e} {r(direct):.2e}', 'winograd', f'{e(y):.2e} {r(y):.2e}')
