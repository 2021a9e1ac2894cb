"""Test infrastructure: a SEVENTH evaluation order for the CPU oracle.  `winograd_convolutions()` makes every stride-1 three-tap
/ 3x3 convolution of oracle/dynmm_oracle.py evaluate in the 1-D Winograd F(2,3) form — four channel contractions per output
pair, `y0 = M0 + M1 + M2, y1 = M1 - M2 - M3` with `M_i = U_i V_i` — built from torch ops, so autograd differentiates through the
same form.  Mathematically identical to F.conv2d; in fp32 it is another correct rounding of the same sums, which is what the
noise calibration (tests/golden/make_grad_noise.py) and the oracle-vs-oracle tests need: the HIP path evaluates these
convolutions in this form."""
import contextlib

import torch
import torch.nn.functional as F

from oracle import dynmm_oracle as O


def _wino_1x3(x, g0, g1, g2):
    """sum_k g_k * x[..., w + k - 1] along the last axis (zero padding 1), g_k [Co, Ci]."""
    W = x.shape[-1]
    P = (W + 1) // 2
    xp = F.pad(x, (1, 2 * P + 2 - (W + 1)))                   # index w + 1 holds x[w]; total length 2P + 2
    d0, d1, d2, d3 = (xp[..., k:k + 2 * P:2] for k in range(4))
    v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    u = (g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2)
    m = [F.conv2d(vi, ui[:, :, None, None]) for vi, ui in zip(v, u)]
    y = torch.stack((m[0] + m[1] + m[2], m[1] - m[2] - m[3]), dim=-1).flatten(-2)
    return y[..., :W]


def wino_conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    st = (stride, stride) if isinstance(stride, int) else tuple(stride)
    pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
    kh, kw = w.shape[2:]
    ok = st == (1, 1) and groups == 1 and dilation in (1, (1, 1)) and x.shape[1] >= 16 and \
        (kh, kw) + pd in ((1, 3, 0, 1), (3, 1, 1, 0), (3, 3, 1, 1))
    if not ok:
        return _orig_conv2d(x, w, bias, stride, padding, dilation, groups)
    if (kh, kw) == (1, 3):
        y = _wino_1x3(x, w[:, :, 0, 0], w[:, :, 0, 1], w[:, :, 0, 2])
    elif (kh, kw) == (3, 1):
        y = _wino_1x3(x.transpose(2, 3), w[:, :, 0, 0], w[:, :, 1, 0], w[:, :, 2, 0]).transpose(2, 3)
    else:                                                       # 3x3: horizontal transform, the vertical taps are summed
        H = x.shape[2]
        xp = F.pad(x, (0, 0, 1, 1))
        y = sum(_wino_1x3(xp[:, :, r:r + H], w[:, :, r, 0], w[:, :, r, 1], w[:, :, r, 2]) for r in range(3))
    return y if bias is None else y + bias[None, :, None, None]


_orig_conv2d = F.conv2d


@contextlib.contextmanager
def winograd_convolutions():
    saved = O.F.conv2d
    O.F.conv2d = wino_conv2d
    try:
        yield
    finally:
        O.F.conv2d = saved
