"""1-D Winograd F(4,3) in fp32 (input gradient) and its transposed form F(3,4) (weight gradient) against fp64, beside the
direct fp32 sums and F(2,3): is the arithmetic good enough for the BACKWARD passes (bars: 2e-4 max-norm per tensor)?"""
import torch, torch.nn.functional as F
torch.manual_seed(0)
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
e = lambda a, ref: ((a.double() - ref).abs().max() / ref.abs().max()).item()
for C, N, H, W in ((64, 8, 60, 80), (128, 8, 30, 80), (512, 8, 15, 20)):
    x = torch.randn(N, C, H, W); w = torch.randn(C, C, 1, 3) * (2.0 / (3 * C)) ** 0.5; dy = torch.randn(N, C, H, W)
    ref = F.conv2d(x.double(), w.double(), padding=(0, 1))
    direct = F.conv2d(x, w, padding=(0, 1))
    g = w[..., 0, :]                                                   # [co, ci, 3]
    U = torch.einsum('ij,ocj->ioc', G.float(), g)                      # 6 x [co, ci]
    xp = F.pad(x, (1, 1))
    d = torch.stack([xp[..., j:j + W:4] if False else xp[..., j::4][..., :W // 4] for j in range(6)], 0)   # d_j of quad t = xp[4t + j]
    V = torch.einsum('ij,jnchw->inchw', BT.float(), d)
    M = torch.stack([torch.einsum('oc,nchw->nohw', U[i], V[i]) for i in range(6)], 0)
    Y = torch.einsum('ij,jnohw->inohw', AT.float(), M)                 # 4 x [n, o, h, W/4]
    y = Y.permute(1, 2, 3, 4, 0).reshape(ref.shape)
    # weight gradient, taps s: dW[o,c,s] = sum dy[o, p] x[c, p + s - 1]
    dwr = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=(0, 1))[..., 0, :]
    dwd = torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=(0, 1))[..., 0, :]
    E = torch.stack([dy[..., j::4] for j in range(4)], 0)             # e_j of quad t
    AE = torch.einsum('ij,jnohw->inohw', AT.t().float().contiguous(), E)       # A (6x4) e
    Mw = torch.stack([torch.einsum('nohw,nchw->oc', AE[i], V[i]) for i in range(6)], 0)
    dw = torch.einsum('ij,ioc->ocj', G.float(), Mw)                    # G^T m
    print(f'C={C} {N}x{H}x{W}: fwd/dgrad max-norm err direct {e(direct, ref):.2e}  F(4,3) {e(y, ref):.2e} | wgrad direct {e(dwd, dwr):.2e}  F(3,4) {e(dw, dwr):.2e}')
