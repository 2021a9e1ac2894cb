"""Shared helpers for the parity tests (CPU oracle side)."""
import numpy as np
import torch

from dynmm_amd import synth
from oracle import dynmm_oracle as O

CFGS = {
    'P_se': O.Config(encoder_block='NonBottleneck1D', fuse='SE-add'),
    'P_add': O.Config(encoder_block='NonBottleneck1D', fuse='add'),
    'S_se': O.Config(encoder_block='BasicBlock', fuse='SE-add'),
    'S_add': O.Config(encoder_block='BasicBlock', fuse='add'),
    'R50_se': O.Config(encoder='resnet50', encoder_block='BasicBlock', fuse='SE-add'),
    'R18_se': O.Config(encoder='resnet18', encoder_block='BasicBlock', fuse='SE-add', nr_decoder_blocks=[1, 1, 1]),
}


def state_dict_template(cfg: O.Config):
    """Key -> shape of SkipGateESANet's state_dict for `cfg`, derived from the build's own module
    (the reference is not importable on the GPU box).  Checked against the reference's 907-entry
    contract in tests/test_contract.py."""
    from dynmm_amd.nn.net import SkipGateESANet
    m = SkipGateESANet(height=96, width=128, encoder_rgb=cfg.encoder, encoder_depth=cfg.encoder,
                       encoder_block=cfg.encoder_block,
                       fuse_depth_in_rgb_encoder=cfg.fuse, channels_decoder=cfg.channels_decoder,
                       nr_decoder_blocks=cfg.nr_decoder_blocks, num_classes=cfg.num_classes)
    return {k: v.clone() for k, v in m.state_dict().items()}


def filled_state_dict(cfg: O.Config, seed=0):
    sd = state_dict_template(cfg)
    synth.fill_state_dict(sd, seed)
    return sd


def ini_index(n):
    return torch.tensor([(3 * i + 1) % 5 for i in range(n)])


def ini_weight(n):
    w = torch.zeros(n, 5)
    w[range(n), ini_index(n)] = 1
    return w


def grad_probe(shape, tag):
    r = np.random.Generator(np.random.PCG64([99, sum(shape), len(tag)]))
    return torch.from_numpy(r.standard_normal(size=shape).astype(np.float32))


def train_loss(outs, loss_flop):
    total = 3.0 * loss_flop
    for i, o in enumerate(outs):
        total = total + (o * grad_probe(tuple(o.shape), f's{i}').to(o.device)).mean()
    return total


MODE_KW = {
    'eval_baseline': dict(baseline=True),
    'eval_soft': dict(),
    'eval_hard': dict(hard_gate=True),
    'eval_ini': dict(ini_stage=True),
    'train_soft': dict(training=True),
    'train_hard': dict(training=True, hard_gate=True, temp=0.5),
}


def rel_err(a, b):
    """max |a-b| / max |b|  (the measure SURVEY.md §0-6 / north_star's 1e-3 bar is stated in)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
