"""Encoder building blocks.  Modules here are *parameter containers* (ordinary nn.Conv2d /
nn.BatchNorm2d children so that state_dict keys and shapes equal the reference's,
FusionDynMM/src/models/resnet.py and model_utils.py) — their own forward is never called; the
block-level forward below drives the HIP kernels through dynmm_amd.ops.
"""
import torch
import torch.nn as nn

from .. import ops


def conv_bn_act(x, conv, bn, act=None, residual=None, x2=None, mask_input=False, conv_link=None, res_link=None, bwd_link=None):
    """act(BN(conv(x|x2)) + residual).

    inference (eval, no grad): ONE kernel — BN folded into the implicit-GEMM epilogue.
    training: conv(+bias) kernel, batch statistics pass, normalise(+residual+act) pass.
    """
    if not bn.training and not torch.is_grad_enabled():
        return ops.conv2d_fused_eval(x, conv.weight, conv.bias, bn, act, residual,
                                     conv.stride, conv.padding, x2)
    y = ops.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, None, x2, mask_input=mask_input, link=conv_link,
                   bn_stats=bn.training)
    return ops.batch_norm_act(y, bn, act, residual, link=res_link, bwd_link=bwd_link)


class ConvBNAct(nn.Sequential):
    """conv (no bias) -> BN -> ReLU; keys '<p>.conv.weight', '<p>.bn.*' (model_utils.py:11-23)."""

    def __init__(self, cin, cout, kernel_size):
        super().__init__()
        self.add_module('conv', nn.Conv2d(cin, cout, kernel_size, padding=kernel_size // 2, bias=False))
        self.add_module('bn', nn.BatchNorm2d(cout))

    def forward(self, x):
        return conv_bn_act(x, self.conv, self.bn, 'relu')


class NonBottleneck1D(nn.Module):
    """ERFNet factorised residual block (resnet.py:87-147): 3x1 -> ReLU -> 1x3 -> BN(eps 1e-3) ->
    ReLU -> 3x1 -> ReLU -> 1x3 -> BN -> (+identity) -> ReLU; stride is split (s,1)/(1,s)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(inplanes, planes, (3, 1), stride=(stride, 1), padding=(1, 0))
        self.conv1x3_1 = nn.Conv2d(planes, planes, (1, 3), stride=(1, stride), padding=(0, 1))
        self.bn1 = nn.BatchNorm2d(planes, eps=1e-3)
        self.conv3x1_2 = nn.Conv2d(planes, planes, (3, 1), padding=(1, 0))
        self.conv1x3_2 = nn.Conv2d(planes, planes, (1, 3), padding=(0, 1))
        self.bn2 = nn.BatchNorm2d(planes, eps=1e-3)
        self.downsample = downsample

    def forward(self, x, chain=False):
        """chain: x is the output of the previous NonBottleneck1D block and this block is its only consumer (the stage loops
        below say so) — the previous block's bn2 backward reductions then come out of this block's first input-gradient launch."""
        # Backward fusions (forward results unchanged): the ReLU backward of each 3x1 conv is applied in
        # the dgrad epilogue of the 1x3 conv that consumes it, and the identity branch's gradient is
        # added in the dgrad epilogue of the first conv instead of a separate autograd add pass.
        fuse_bwd = torch.is_grad_enabled() and x.requires_grad
        link = ops.GradLink() if (fuse_bwd and self.downsample is None) else None
        in_link = getattr(x, '_bn_out_link', None) if (chain and link is not None) else None
        xd = x
        if self.downsample is not None:
            x, xd = ops.fan_out(x, 2)            # x feeds the first conv AND the down-sample conv: one fused gradient sum
        c = self.conv3x1_1
        y = ops.conv2d(x, c.weight, c.bias, c.stride, c.padding, 'relu', defer_mask=fuse_bwd, link=link, bn_link=in_link)
        # bn1 + ReLU feed conv3x1_2 and nothing else: its input-gradient launch also does bn1's backward reductions (ops.BNLink)
        bnl = ops.BNLink() if (fuse_bwd and self.bn1.training) else None
        y = conv_bn_act(y, self.conv1x3_1, self.bn1, 'relu', mask_input=fuse_bwd, bwd_link=bnl)
        c = self.conv3x1_2
        y = ops.conv2d(y, c.weight, c.bias, c.stride, c.padding, 'relu', defer_mask=fuse_bwd, bn_link=bnl)
        idt = x if self.downsample is None else conv_bn_act(xd, self.downsample[0], self.downsample[1])
        out_link = ops.BNLink() if (fuse_bwd and self.bn2.training) else None
        out = conv_bn_act(y, self.conv1x3_2, self.bn2, 'relu', residual=idt, mask_input=fuse_bwd, res_link=link, bwd_link=out_link)
        if out_link is not None and out_link.bits is not None:
            out._bn_out_link = out_link            # picked up by the next block of the stage (chain=True), by nothing else
        return out


class BasicBlock(nn.Module):
    """Two 3x3 conv + BN, residual, ReLU (resnet.py:42-84)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x, chain=False):
        fuse_bwd = torch.is_grad_enabled() and x.requires_grad
        link = ops.GradLink() if (fuse_bwd and self.downsample is None) else None
        xd = x
        if self.downsample is not None:
            x, xd = ops.fan_out(x, 2)
        y = conv_bn_act(x, self.conv1, self.bn1, 'relu', conv_link=link)
        idt = x if self.downsample is None else conv_bn_act(xd, self.downsample[0], self.downsample[1])
        return conv_bn_act(y, self.conv2, self.bn2, 'relu', residual=idt, res_link=link)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride) -> 1x1 (x4 channels), each conv + BN, residual, ReLU — the ResNet-50
    block (resnet.py:150-192); `--encoder resnet50` is the reference CLI's default (src/args.py:105)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.downsample = downsample

    def forward(self, x, chain=False):
        fuse_bwd = torch.is_grad_enabled() and x.requires_grad
        link = ops.GradLink() if (fuse_bwd and self.downsample is None) else None
        xd = x
        if self.downsample is not None:
            x, xd = ops.fan_out(x, 2)
        y = conv_bn_act(x, self.conv1, self.bn1, 'relu', conv_link=link)
        y = conv_bn_act(y, self.conv2, self.bn2, 'relu')
        idt = x if self.downsample is None else conv_bn_act(xd, self.downsample[0], self.downsample[1])
        return conv_bn_act(y, self.conv3, self.bn3, 'relu', residual=idt, res_link=link)


def chain_ok(prev, blk):
    """May `blk` treat `prev`'s output as consumed by itself alone (ResNetEncoder._stage)?  Not when a forward hook — on `prev`, a
    pre-hook on `blk`, or a global module hook — gets to see that tensor: it may keep it in the autograd graph."""
    import torch.nn.modules.module as M
    if prev._forward_hooks or blk._forward_pre_hooks:
        return False
    if M._global_forward_hooks or M._global_forward_pre_hooks:
        return False
    return True


BLOCKS = {'NonBottleneck1D': NonBottleneck1D, 'BasicBlock': BasicBlock, 'Bottleneck': Bottleneck}
LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet50': (3, 4, 6, 3)}


class ResNetEncoder(nn.Module):
    """ResNet-18/34 trunk with stage-wise entry points (resnet.py:195-379)."""

    def __init__(self, name, block, input_channels=3):
        super().__init__()
        if name not in LAYERS:
            raise NotImplementedError(f'Only {sorted(LAYERS)} encoders are implemented on the HIP path. Got {name}')
        if block not in BLOCKS and name != 'resnet50':
            raise NotImplementedError(f'Block {block} is not implemented')
        if name == 'resnet50':
            block = 'Bottleneck'          # ResNet50() ignores encoder_block (…globalgate.py:93-96, resnet.py:450-452)
        blk = BLOCKS[block]
        ex = blk.expansion
        self.conv1 = nn.Conv2d(input_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for j, (planes, n) in enumerate(zip((64, 128, 256, 512), LAYERS[name]), start=1):
            stride = 1 if j == 1 else 2
            down = None
            if stride != 1 or inplanes != planes * ex:
                down = nn.Sequential(nn.Conv2d(inplanes, planes * ex, 1, stride=stride, bias=False),
                                     nn.BatchNorm2d(planes * ex))
            stage = [blk(inplanes, planes, stride, down)] + [blk(planes * ex, planes) for _ in range(n - 1)]
            setattr(self, f'layer{j}', nn.Sequential(*stage))
            inplanes = planes * ex
        self.down_2_channels_out = 64
        self.down_4_channels_out, self.down_8_channels_out = 64 * ex, 128 * ex
        self.down_16_channels_out, self.down_32_channels_out = 256 * ex, 512 * ex
        for m in self.modules():           # resnet.py:264-270
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_first_conv(self, x):
        return conv_bn_act(x, self.conv1, self.bn1, 'relu')

    def _stage(self, x, j):
        """The blocks of stage j in sequence.  THE CHAIN CONTRACT (ops.BNLink, BNRED = 2): `chain=True` tells block i that the
        output of block i - 1 feeds block i's first convolution and identity branch AND NOTHING ELSE — block i - 1's `bn2`
        backward then takes its ReLU-masked gradient and both reductions from block i's first input-gradient launch
        (`out._bn_out_link`, a tensor attribute that any op between the blocks would drop = the safe direction).  A second
        consumer of an intermediate block output (a forward hook that keeps the feature map in the graph, a feature tap) would add
        an UNMASKED gradient to a tensor the BatchNorm backward treats as already masked: silently wrong gradients.
        So the chain is only offered when no hook can see the intermediate tensor (`chain_ok`); `ops.BN_BWD_FUSE = False` switches
        the whole mechanism off (every BatchNorm backward then does its own reductions)."""
        blocks = list(getattr(self, f'layer{j}'))
        for i, blk in enumerate(blocks):
            x = blk(x, chain=i > 0 and chain_ok(blocks[i - 1], blk))
        return x

    def forward_layer1(self, x):
        return self._stage(x, 1)

    def forward_layer2(self, x):
        return self._stage(x, 2)

    def forward_layer3(self, x):
        return self._stage(x, 3)

    def forward_layer4(self, x):
        return self._stage(x, 4)
