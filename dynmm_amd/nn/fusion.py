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
