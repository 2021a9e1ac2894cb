"""CPU oracle for the modality-level DynMM path (ModalityDynMM/affect/affect_dyn.py).
*** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***  Only tests/ and bench.py's checker legs may import it.

PARITY UNPINNED.  The arithmetic of the reference's experts lives in MultiBench (`unimodals.common_models`,
`fusions.common_fusions`), which /root/reference neither vendors nor pins to a commit (ModalityDynMM README;
affect_dyn.py:12-15), and the reference holds no golden vector, known-answer test or fixture for this path.  This
file restates MultiBench's published module definitions with the real torch.nn layers they wrap (dropout set to 0:
the deterministic, eval-mode arithmetic; the training-mode arithmetic with EXPLICIT keep flags is `encoder_layer_dropout`,
pinned to torch's own layer in tests/test_affect.py) and the reference's own DynMMNetV2 / DynMMNet / DiffSoftmax
(affect_dyn.py:18-28, 31-104, 107-175) and training objective (Supervised_Learning.py:120-144).  State-dict keys
equal those of dynmm_amd.nn.affect, so one deterministic fill drives both sides.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def encoder_layer_dropout(layer, x, p, next_mask):
    """nn.TransformerEncoderLayer.forward (post-norm, ReLU; torch/nn/modules/transformer.py) in training mode, with the four
    dropout decisions supplied by the caller instead of drawn: x [T, B, D]; next_mask(name, shape) -> keep flags in the
    HIP path's layouts — 'attn' [B*heads, T, T], 'dropout1' / 'dropout2' [B, D, T], 'dropout' [B, dim_feedforward, T]."""
    T, B, D = x.shape
    sa = layer.self_attn
    H = sa.num_heads
    dh = D // H
    keep = lambda name, shape: next_mask(name, shape).to(x.dtype) / (1.0 - p)
    q, k, v = F.linear(x, sa.in_proj_weight, sa.in_proj_bias).chunk(3, dim=-1)
    q, k, v = (t.reshape(T, B * H, dh).transpose(0, 1) for t in (q, k, v))          # [B*H, T, dh], batch index b*H + h
    att = torch.softmax((q / math.sqrt(dh)) @ k.transpose(1, 2), dim=-1) * keep('attn', (B * H, T, T))
    o = (att @ v).transpose(0, 1).reshape(T, B, D)
    o = F.linear(o, sa.out_proj.weight, sa.out_proj.bias)
    x = layer.norm1(x + o * keep('dropout1', (B, D, T)).permute(2, 0, 1))
    f = F.relu(layer.linear1(x))
    f = layer.linear2(f * keep('dropout', (B, f.shape[-1], T)).permute(2, 0, 1))
    return layer.norm2(x + f * keep('dropout2', (B, D, T)).permute(2, 0, 1))


class Transformer(nn.Module):
    """MultiBench unimodals/common_models.py `Transformer(n_features, dim)`."""

    def __init__(self, n_features, dim, nhead=5, num_layers=5):
        super().__init__()
        self.conv = nn.Conv1d(n_features, dim, kernel_size=1, padding=0, bias=False)
        layer = nn.TransformerEncoderLayer(d_model=dim, nhead=nhead, dropout=0.0)
        self.transformer = nn.TransformerEncoder(layer, num_layers=num_layers, enable_nested_tensor=False)

    dropout_masks = None      # (p, next_mask): training-mode arithmetic with injected keep flags (see encoder_layer_dropout)

    def forward(self, x):
        if isinstance(x, (list, tuple)):
            x = x[0]
        x = self.conv(x.permute([0, 2, 1]))
        x = x.permute([2, 0, 1])
        if Transformer.dropout_masks is not None:
            p, next_mask = Transformer.dropout_masks
            for layer in self.transformer.layers:
                x = encoder_layer_dropout(layer, x, p, next_mask)
            return x[-1]
        return self.transformer(x)[-1]


class MLP(nn.Module):
    def __init__(self, indim, hiddim, outdim):
        super().__init__()
        self.fc = nn.Linear(indim, hiddim)
        self.fc2 = nn.Linear(hiddim, outdim)

    def forward(self, x):
        return self.fc2(F.relu(self.fc(x)))


class Concat(nn.Module):
    def forward(self, modalities):
        return torch.cat([torch.flatten(m, start_dim=1) for m in modalities], dim=1)


class MMDL(nn.Module):
    """training_structures/Supervised_Learning.py:16-51, has_padding=True, tensor-valued encoders."""

    def __init__(self, encoders, fusion, head):
        super().__init__()
        self.encoders = nn.ModuleList(encoders)
        self.fuse, self.head = fusion, head

    def forward(self, inputs):
        outs = [enc([inputs[0][i], inputs[1][i]]) for i, enc in enumerate(self.encoders)]
        return self.head(self.fuse(outs))


def diff_softmax(logits, tau=1.0, hard=False, dim=-1):
    """affect_dyn.py:18-28."""
    y_soft = (logits / tau).softmax(dim)
    if not hard:
        return y_soft
    index = y_soft.max(dim, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(dim, index, 1.0)
    return y_hard - y_soft.detach() + y_soft


class DynMMNetV2(nn.Module):
    """affect_dyn.py:107-175 (experts constructed instead of unpickled)."""

    def __init__(self, temp=1.0, hard_gate=False):
        super().__init__()
        self.text_encoder = Transformer(300, 120)
        self.text_head = MLP(120, 64, 1)
        self.branch2 = MMDL([Transformer(35, 60), Transformer(74, 120), Transformer(300, 120)], Concat(), MLP(300, 128, 1))
        self.gate = nn.Sequential(Transformer(409, 10), nn.Linear(10, 2))
        self.temp, self.hard_gate = temp, hard_gate

    def forward(self, inputs):
        x = torch.cat(inputs[0], dim=2)
        weight = diff_softmax(self.gate([x, inputs[1][0]]), tau=self.temp, hard=self.hard_gate)
        preds = [self.text_head(self.text_encoder([inputs[0][2], inputs[1][2]])), self.branch2(inputs)]
        out = weight[:, 0:1] * preds[0] + weight[:, 1:2] * preds[1]
        return out, weight[:, 1].mean(), weight


class DynMMNet(nn.Module):
    """affect_dyn.py:31-104, forward2."""

    def __init__(self, temp=1.0, hard_gate=False):
        super().__init__()
        self.encoders = nn.ModuleList([Transformer(f, 120) for f in (35, 74, 300)])
        self.heads = nn.ModuleList([MLP(120, 64, 1) for _ in range(3)])
        self.gate = nn.Sequential(Transformer(409, 10), nn.Linear(10, 3))
        self.temp, self.hard_gate = temp, hard_gate

    def forward(self, inputs):
        x = torch.cat(inputs[0], dim=2)
        weight = diff_softmax(self.gate([x, inputs[1][0]]), tau=self.temp, hard=self.hard_gate)
        preds = [self.heads[i](self.encoders[i]([inputs[0][i], inputs[1][i]])) for i in range(3)]
        out = weight[:, 0:1] * preds[0] + weight[:, 1:2] * preds[1] + weight[:, 2:3] * preds[2]
        return out, weight[:, 2].mean(), weight


def train_objective(out, aux, target, lossw):
    """Supervised_Learning.py:135-136 with objective = nn.L1Loss()."""
    loss1 = F.l1_loss(out, target)
    return loss1 + lossw * aux, loss1


def fill_(module, seed=0):
    """Deterministic, well-conditioned weights (keyed on the state_dict key; dynmm_amd.synth is image-model specific)."""
    import zlib

    import numpy as np
    with torch.no_grad():
        for k, t in module.state_dict().items():
            r = np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(k.encode())]))
            if k.endswith('norm1.weight') or k.endswith('norm2.weight'):
                v = r.uniform(0.8, 1.2, size=tuple(t.shape))
            elif t.dim() >= 2:
                fan_in = int(np.prod(t.shape[1:]))
                v = r.standard_normal(size=tuple(t.shape)) * np.sqrt(1.0 / fan_in)
            else:
                v = 0.05 * r.standard_normal(size=tuple(t.shape))
            t.copy_(torch.from_numpy(v.astype(np.float32)))
    return module


def synth_batch(batch, T=50, seed=0, device='cpu'):
    """CMU-MOSEI-shaped batch: [[visual [B,T,35], audio [B,T,74], text [B,T,300]], [lengths]*3], target [B,1]."""
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(batch, T, f, generator=g) for f in (35, 74, 300)]
    lens = [torch.full((batch,), T, dtype=torch.long)] * 3
    y = torch.randn(batch, 1, generator=g)
    return [[x.to(device) for x in xs], lens], y.to(device)
