"""Pyramid pooling context module, bins (1,5), nearest upsampling (context_modules.py:47-87)."""
import torch.nn as nn

from .. import ops
from .blocks import ConvBNAct


class PyramidPoolingModule(nn.Module):
    def __init__(self, in_dim, out_dim, bins=(1, 5)):
        super().__init__()
        self.bins = tuple(bins)
        red = in_dim // len(bins)
        # index 0 of each branch is the parameter-free adaptive pool of the reference Sequential
        self.features = nn.ModuleList([nn.Sequential(nn.Identity(), ConvBNAct(in_dim, red, 1)) for _ in bins])
        self.final_conv = ConvBNAct(in_dim + red * len(bins), out_dim, 1)

    def forward(self, x):
        branches = [f[1](ops.adaptive_avg_pool(x, b)) for f, b in zip(self.features, self.bins)]
        return self.final_conv(ops.nearest_concat(x, *branches))


def get_context_module(name, channels_in, channels_out):
    if 'appm' in name:
        raise NotImplementedError('appm context modules are outside the HIP hot path (SURVEY.md §2.1 #5)')
    if 'ppm' in name:
        bins = (1, 2, 4, 8) if name == 'ppm-1-2-4-8' else (1, 5)
        return PyramidPoolingModule(channels_in, channels_out, bins), channels_out
    return nn.Identity(), channels_in
