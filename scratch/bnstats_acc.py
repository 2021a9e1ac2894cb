"""Accuracy of the BatchNorm batch statistics: conv-epilogue partials vs the pass over y, both against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynmm_amd import ops
torch.manual_seed(0)
for (N, C_, H, W, k) in [(4, 64, 120, 160, (1, 3)), (8, 128, 60, 80, (1, 3)), (3, 128, 12, 16, (3, 3)), (32, 512, 15, 20, (1, 3))]:
    conv = torch.nn.Conv2d(C_, C_, k, padding=(k[0] // 2, k[1] // 2)).cuda()
    with torch.no_grad(): conv.bias.normal_(0, 2.0)          # means of order 2: var = E[x^2] - mean^2 cancels
    x = torch.randn(N, C_, H, W, device='cuda')
    res = {}
    orig = ops._stats_tiles
    for fused in (True, False):
        ops._stats_tiles = orig if fused else (lambda g: 0)
        bn = torch.nn.BatchNorm2d(C_).cuda().train(); bn.momentum = 1.0
        with torch.no_grad():
            y = ops.conv2d(x.requires_grad_(True), conv.weight, conv.bias, 1, conv.padding, bn_stats=True).detach() if False else None
        yy = ops.conv2d(x.clone().requires_grad_(True), conv.weight, conv.bias, 1, conv.padding, bn_stats=True)
        out = ops.batch_norm_act(yy, bn, None)
        y64 = yy.detach().double()
        m64 = y64.mean((0, 2, 3)); v64 = y64.var((0, 2, 3), unbiased=True)
        res[fused] = ((bn.running_mean.double() - m64).abs().max().item() / m64.abs().max().item(),
                      ((bn.running_var.double() - v64).abs() / v64).max().item(), hasattr(yy, '_dynmm_stats'))
    ops._stats_tiles = orig
    print((N, C_, H, W, k), 'fused  mean err %.1e var err %.1e (partials used: %s)' % res[True], '| pass over y  mean err %.1e var err %.1e' % res[False][:2])
