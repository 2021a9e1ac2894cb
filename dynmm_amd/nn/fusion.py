"""SE fusion containers (rgb_depth_fusion.py:13-26, model_utils.py:36-51).  The arithmetic (GAP,
excitation MLPs, channel scaling, modality sum AND the gate blend) runs in ops.se_fuse_blend."""
import torch.nn as nn


class SqueezeAndExcitation(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        # indices 1 and 3 are the (parameter-free) ReLU / Sigmoid slots of the reference Sequential
        self.fc = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1), nn.Identity(),
                                nn.Conv2d(channel // reduction, channel, 1), nn.Identity())

    def mlp_params(self):
        return [self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias]


class SqueezeAndExciteFusionAdd(nn.Module):
    def __init__(self, channels_in):
        super().__init__()
        self.se_rgb = SqueezeAndExcitation(channels_in)
        self.se_depth = SqueezeAndExcitation(channels_in)

    def params8(self):
        return self.se_rgb.mlp_params() + self.se_depth.mlp_params()


class SqueezeAndExcitationWeight(nn.Module):
    """Parameter container of model_utils.py:54-70 (the `linear` layer is unused by the reference's
    forward but part of its state_dict)."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(channel, channel // reduction, 1), nn.Identity(),
                                nn.Conv2d(channel // reduction, channel, 1), nn.Identity())
        self.linear = nn.Linear(channel, 2)

    def mlp_params(self):
        return [self.fc[0].weight, self.fc[0].bias, self.fc[2].weight, self.fc[2].bias]


class SqueezeAndExciteReweigh(nn.Module):
    """Per-stage 2-way Gumbel gate (rgb_depth_fusion.py:29-65).  Stand-alone call evaluates the gate only;
    inside SkipESANet the gate is fused with the stage blend (ops.reweigh_fuse)."""

    def __init__(self, temp, channels_in):
        super().__init__()
        self.temp = temp
        self.se = SqueezeAndExcitationWeight(channels_in * 2)
        self.act = nn.Identity()          # nn.Sigmoid() slot: parameter-free

    def random_weights(self, bs, device, prev_weight=None):
        import torch
        b0 = torch.randint(0, 2, (bs,))                       # CPU RNG, as the reference (…fusion.py:39-42)
        w = torch.stack([b0, 1 - b0], dim=1).to(device=device, dtype=torch.float32)
        if prev_weight is not None:
            b1 = w[:, 1] * prev_weight
            w = torch.stack([1 - b1, b1], dim=1)
        return w

    def forward(self, rgb, depth, hard=False, prev_weight=None, random=False, test=False, noise=None):
        from .. import ops
        if random:
            return self.random_weights(rgb.shape[0], rgb.device, prev_weight).view(-1, 2, 1, 1)
        _, w, _ = ops.reweigh_fuse(rgb, depth, None, 0, self.se.mlp_params(), self.temp, hard or test,
                                   prev_weight, noise)
        return w.view(-1, 2, 1, 1)
