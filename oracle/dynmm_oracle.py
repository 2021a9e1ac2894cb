"""CPU oracle for the fusion-level DynMM hot path.  *** TEST INFRASTRUCTURE — NOT PRODUCT CODE. ***

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the
product path (dynmm_amd/) never does and fails loudly if its HIP library is missing.

This is a functional (state_dict -> tensors) restatement, in plain PyTorch fp32 on the CPU, of what
the reference computes on its hot path.  Every function cites the reference lines it follows
(paths relative to /root/reference/FusionDynMM).  Parity is PINNED: tests/test_oracle_golden.py
checks this file against fixtures in tests/golden/ that were produced by importing the reference
itself in the build container (tests/golden/make_goldens.py), plus the reference's own
known-answer values (confusion-matrix example src/confusion_matrix.py:181-198; MAC tables
src/models/model_skip_mod_globalgate.py:217-223).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

RESNET_LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet50': (3, 4, 6, 3)}   # src/models/resnet.py:392,425,452
STAGE_PLANES = (64, 128, 256, 512)                                       # src/models/resnet.py:247-266
# src/models/model_skip_mod_globalgate.py:219 (ResNet-34 table) and :222 (otherwise)
DEPTH_ENC_FLOP_R34 = (0.2506752, 3.1113216, 6.9470208, 12.66432, 15.538944)
DEPTH_ENC_FLOP_OTHER = (0.2506752, 4.39420573, 10.72382115, 19.71582947, 24.679084)


@dataclass
class Config:
    """Constructor arguments of SkipGateESANet (src/models/model_skip_mod_globalgate.py:34-51)."""
    num_classes: int = 40
    encoder: str = 'resnet34'
    encoder_block: str = 'NonBottleneck1D'
    channels_decoder: List[int] = field(default_factory=lambda: [128, 128, 128])
    nr_decoder_blocks: List[int] = field(default_factory=lambda: [3, 3, 3])
    fuse: str = 'SE-add'          # 'add' | 'SE-add'
    bn_momentum: float = 0.1


# --------------------------------------------------------------------------------------------
# micro blocks
# --------------------------------------------------------------------------------------------
def _conv(sd, p, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[p + '.weight'], sd.get(p + '.bias'), stride, padding, 1, groups)


def _bn(sd, p, x, training, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d forward; in training mode updates running stats in `sd` in place."""
    rm, rv = sd[p + '.running_mean'], sd[p + '.running_var']
    if training and (p + '.num_batches_tracked') in sd:
        sd[p + '.num_batches_tracked'] += 1
    return F.batch_norm(x, rm, rv, sd[p + '.weight'], sd[p + '.bias'], training, momentum, eps)


def conv_bn_act(sd, p, x, training, padding=0):
    """ConvBNAct (src/models/model_utils.py:11-23): conv(no bias) -> BN(eps 1e-5) -> ReLU."""
    return F.relu(_bn(sd, p + '.bn', _conv(sd, p + '.conv', x, 1, padding), training))


def non_bottleneck_1d(sd, p, x, training, stride=1):
    """ERFNet block, src/models/resnet.py:124-147 (BN eps 1e-3 at :110,:118; conv bias=True)."""
    y = F.relu(_conv(sd, p + '.conv3x1_1', x, (stride, 1), (1, 0)))
    y = _conv(sd, p + '.conv1x3_1', y, (1, stride), (0, 1))
    y = F.relu(_bn(sd, p + '.bn1', y, training, eps=1e-3))
    y = F.relu(_conv(sd, p + '.conv3x1_2', y, 1, (1, 0)))
    y = _conv(sd, p + '.conv1x3_2', y, 1, (0, 1))
    y = _bn(sd, p + '.bn2', y, training, eps=1e-3)
    idt = x
    if (p + '.downsample.0.weight') in sd:               # src/models/resnet.py:293-297
        idt = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x, stride), training)
    return F.relu(y + idt)


def basic_block(sd, p, x, training, stride=1):
    """src/models/resnet.py:66-84: two 3x3 conv(no bias)+BN, residual, ReLU."""
    y = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, stride, 1), training))
    y = _bn(sd, p + '.bn2', _conv(sd, p + '.conv2', y, 1, 1), training)
    idt = x
    if (p + '.downsample.0.weight') in sd:
        idt = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x, stride), training)
    return F.relu(y + idt)


def bottleneck(sd, p, x, training, stride=1):
    """src/models/resnet.py:172-192: 1x1 -> 3x3 (stride) -> 1x1 (x4), each conv(no bias)+BN, residual, ReLU."""
    y = F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x), training))
    y = F.relu(_bn(sd, p + '.bn2', _conv(sd, p + '.conv2', y, stride, 1), training))
    y = _bn(sd, p + '.bn3', _conv(sd, p + '.conv3', y), training)
    idt = x
    if (p + '.downsample.0.weight') in sd:
        idt = _bn(sd, p + '.downsample.1', _conv(sd, p + '.downsample.0', x, stride), training)
    return F.relu(y + idt)


def encoder_stem(sd, p, x, training):
    """ResNet.forward_first_conv (src/models/resnet.py:352-358)."""
    return F.relu(_bn(sd, p + '.bn1', _conv(sd, p + '.conv1', x, 2, 3), training))


def encoder_stage(sd, p, x, training, cfg, j):
    """ResNet.forward_layer{j} (src/models/resnet.py:360-379); stride 2 in the first block of j>=2."""
    blk = bottleneck if cfg.encoder == 'resnet50' else \
        (non_bottleneck_1d if cfg.encoder_block == 'NonBottleneck1D' else basic_block)
    for i in range(RESNET_LAYERS[cfg.encoder][j - 1]):
        stride = 2 if (i == 0 and j > 1) else 1
        x = blk(sd, f'{p}.layer{j}.{i}', x, training, stride)
    return x


def squeeze_excite(sd, p, x):
    """SqueezeAndExcitation (src/models/model_utils.py:47-51)."""
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.relu(_conv(sd, p + '.fc.0', s))
    s = torch.sigmoid(_conv(sd, p + '.fc.2', s))
    return x * s


def fuse_rgbd(sd, p, rgb, depth, cfg):
    """'add' or SqueezeAndExciteFusionAdd (src/models/rgb_depth_fusion.py:22-26)."""
    if cfg.fuse == 'add':
        return rgb + depth
    return squeeze_excite(sd, p + '.se_rgb', rgb) + squeeze_excite(sd, p + '.se_depth', depth)


def diff_softmax(logits, tau=1.0, hard=False, dim=-1):
    """DiffSoftmax (src/models/model_skip_mod_globalgate.py:20-30): temperature softmax, optional
    straight-through one-hot.  No Gumbel noise."""
    y_soft = (logits / tau).softmax(dim)
    if not hard:
        return y_soft
    idx = y_soft.max(dim, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(dim, idx, 1.0)
    return y_hard - y_soft.detach() + y_soft


def global_gate(sd, p, rgb, depth, training, temp, hard):
    """GlobalGate.forward (src/models/model_skip_mod_globalgate.py:388-394)."""
    x = torch.cat([rgb, depth], dim=1)
    y = torch.tanh(_bn(sd, p + '.conv.1', _conv(sd, p + '.conv.0', x, 2), training))
    y = torch.tanh(_bn(sd, p + '.conv.4', _conv(sd, p + '.conv.3', y, 2), training))
    y = _conv(sd, p + '.fc', F.adaptive_avg_pool2d(y, 1))
    return diff_softmax(y, tau=temp, hard=hard, dim=1).squeeze(-1).squeeze(-1)


def pyramid_pooling(sd, p, x, training, bins=(1, 5)):
    """PyramidPoolingModule.forward with nearest upsampling (src/models/context_modules.py:67-87;
    'learned-3x3' upsampling falls back to nearest in the context module,
    src/models/model_skip_mod_globalgate.py:173-181)."""
    h, w = x.shape[2:]
    outs = [x]
    for i, b in enumerate(bins):
        y = conv_bn_act(sd, f'{p}.features.{i}.1', F.adaptive_avg_pool2d(x, b), training)
        outs.append(F.interpolate(y, (h, w), mode='nearest'))
    return conv_bn_act(sd, p + '.final_conv', torch.cat(outs, 1), training)


def learned_upsample(sd, p, x):
    """Upsample 'learned-3x3-zeropad' (src/models/model.py:404-410): nearest x2, depthwise 3x3."""
    x = F.interpolate(x, (x.shape[2] * 2, x.shape[3] * 2), mode='nearest')
    return _conv(sd, p + '.conv', x, 1, 1, groups=x.shape[1])


def decoder_module(sd, p, x, skip, training, n_blocks):
    """DecoderModule.forward (src/models/model.py:343-357)."""
    y = conv_bn_act(sd, p + '.conv3x3', x, training, padding=1)
    for i in range(n_blocks):
        y = non_bottleneck_1d(sd, f'{p}.decoder_blocks.{i}', y, training)
    side = _conv(sd, p + '.side_output', y) if training else None
    y = learned_upsample(sd, p + '.upsample', y) + skip
    return y, side


def decoder(sd, p, enc_outs, training, cfg):
    """Decoder.forward (src/models/model.py:295-308)."""
    out, s16, s8, s4 = enc_outs
    out, o32 = decoder_module(sd, p + '.decoder_module_1', out, s16, training, cfg.nr_decoder_blocks[0])
    out, o16 = decoder_module(sd, p + '.decoder_module_2', out, s8, training, cfg.nr_decoder_blocks[1])
    out, o8 = decoder_module(sd, p + '.decoder_module_3', out, s4, training, cfg.nr_decoder_blocks[2])
    out = _conv(sd, p + '.conv_out', out, 1, 1)
    out = learned_upsample(sd, p + '.upsample1', out)
    out = learned_upsample(sd, p + '.upsample2', out)
    return (out, o8, o16, o32) if training else out


def _skip(sd, p, x, training):
    """skip_layer{1,2,3}: ConvBNAct 1x1 or empty Sequential (…globalgate.py:145-171)."""
    if (p + '.0.conv.weight') in sd:
        return conv_bn_act(sd, p + '.0', x, training)
    return x


# --------------------------------------------------------------------------------------------
# whole model
# --------------------------------------------------------------------------------------------
def forward(sd, rgb, depth, cfg: Config, training=False, test=False, return_weight=False,
            baseline=False, ini_stage=False, hard_gate=False, temp=1.0,
            ini_weight: Optional[torch.Tensor] = None, detail=None):
    """SkipGateESANet.forward (src/models/model_skip_mod_globalgate.py:255-322).

    `ini_weight` replaces the reference's CPU torch.randint one-hots (:267-270) so that ini_stage
    runs are reproducible.  `detail` (a dict) receives intermediate tensors for per-stage tests.
    Return values follow the reference: test -> out | (out, weight); else (out, flop_loss) where
    `out` is a 4-tuple in training mode (src/models/model.py:306-308).
    """
    r = encoder_stem(sd, 'encoder_rgb', rgb, training)
    d = encoder_stem(sd, 'encoder_depth', depth, training)
    fuse = fuse_rgbd(sd, 'se_layer0', r, d, cfg)
    r = F.max_pool2d(fuse, 3, 2, 1)
    d = F.max_pool2d(d, 3, 2, 1)
    bs = r.shape[0]
    if baseline:
        weight = torch.zeros(bs, 5)
        weight[:, 4] = 1
    elif ini_stage:
        weight = ini_weight
    else:
        weight = global_gate(sd, 'gate_layer', r, d, training, temp, hard_gate)

    skips = []
    for j in (1, 2, 3, 4):
        r = encoder_stage(sd, 'encoder_rgb', fuse if j > 1 else r, training, cfg, j)
        d = encoder_stage(sd, 'encoder_depth', d, training, cfg, j)
        fused = fuse_rgbd(sd, f'se_layer{j}', r, d, cfg)
        if j < 4:
            w = weight[:, :j].sum(1).view(-1, 1, 1, 1)          # :282, :291, :300
            fuse = w * r + (1 - w) * fused
            skips.append(_skip(sd, f'skip_layer{j}', fuse, training))
        else:
            w = weight[:, 4].view(-1, 1, 1, 1)                   # :309-310 (roles swapped)
            fuse = (1 - w) * r + w * fused
        if detail is not None:
            detail[f'fuse{j}'] = fuse
    ctx = pyramid_pooling(sd, 'context_module', fuse, training)
    out = decoder(sd, 'decoder', [ctx, skips[2], skips[1], skips[0]], training, cfg)

    tab = DEPTH_ENC_FLOP_R34 if cfg.encoder == 'resnet34' else DEPTH_ENC_FLOP_OTHER
    loss = (weight.mean(dim=0) * torch.tensor(tab)).mean()       # :314-315, :322
    if detail is not None:
        detail['weight'] = weight
    if test:
        return (out, weight) if return_weight else out
    return out, loss


def forward_esanet(sd, rgb, depth, cfg: Config, training=False):
    """ESANet.forward (src/models/model.py:189-241): the static network — depth fused into the RGB encoder at the stem and
    after every stage, no gate, `out` only (a 4-tuple in training mode, model.py:306-308).  With weight = e_4 the blend of
    `forward` above is w = 0 for stages 1-3 (fuse = fused, :282-301) and w = 1 in the swapped stage-4 rule (:309-310)."""
    return forward(sd, rgb, depth, cfg, training=training, test=True, baseline=True)


# --------------------------------------------------------------------------------------------
# SkipESANet: per-stage Gumbel gates
# --------------------------------------------------------------------------------------------
def gumbel_softmax(logits, exp_noise, tau=1.0, hard=False):
    """torch.nn.functional.gumbel_softmax(dim=-1) with the Exp(1) draw made explicit:
    gumbels = -log(E), E = empty_like(logits).exponential_()  (torch/nn/functional.py, v2.x)."""
    y_soft = ((logits - exp_noise.log()) / tau).softmax(-1)
    if not hard:
        return y_soft
    idx = y_soft.max(-1, keepdim=True)[1]
    y_hard = torch.zeros_like(logits).scatter_(-1, idx, 1.0)
    return y_hard - y_soft.detach() + y_soft


def reweigh_gate(sd, p, rgb, depth, temp, exp_noise, hard=False, prev_weight=None, test=False):
    """SqueezeAndExciteReweigh.forward (src/models/rgb_depth_fusion.py:36-65) with
    SqueezeAndExcitationWeight (src/models/model_utils.py:66-70); `exp_noise` [N,2] replaces the
    global-generator draw inside gumbel_softmax.  Returns [N,2]."""
    x = torch.cat([rgb, depth], dim=1)
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(_conv(sd, p + '.se.fc.0', g))
    g = torch.sigmoid(_conv(sd, p + '.se.fc.2', g))
    w = torch.sigmoid((x * g.expand_as(x)).mean(dim=(1, 2, 3)))
    w = torch.stack([w, 1 - w], dim=1)
    w_norm = gumbel_softmax(w / temp, exp_noise, hard=True if test else hard)
    if prev_weight is not None:
        b1 = w_norm[:, 1] * prev_weight
        w_norm = torch.stack([1 - b1, b1], dim=1)
    return w_norm


def forward_skip(sd, rgb, depth, cfg: Config, exp_noise, training=False, test=False, hard_gate=False,
                 ini_stage=False, temp=1.0, block_rule=(2, 2, 2, 2), detail=None):
    """SkipESANet.forward (src/models/model_skip_mod.py:235-311).  `exp_noise`: 4 tensors [N,2] of
    Exp(1) samples, one per gate, in call order.  Note the reference never uses its se_layer* here:
    the modality fusion is a plain sum.  Returns the decoder output (4-tuple in training mode)."""
    r = encoder_stem(sd, 'encoder_rgb', rgb, training)
    d = encoder_stem(sd, 'encoder_depth', depth, training)
    fuse = r + d
    weights = [reweigh_gate(sd, 'gate_layer0', r, d, temp, exp_noise[0], hard_gate, None, test)]
    r = F.max_pool2d(fuse, 3, 2, 1)
    d = F.max_pool2d(d, 3, 2, 1)
    prev = None
    skips = []
    for j in (1, 2, 3, 4):
        r = encoder_stage(sd, 'encoder_rgb', fuse if j > 1 else r, training, cfg, j)
        d = encoder_stage(sd, 'encoder_depth', d, training, cfg, j)
        b0, b1 = r, r + d
        rule = block_rule[j - 1]
        if rule == 0:
            fuse = b0
        elif rule == 1:
            fuse = b1
        else:
            w = weights[j - 1].view(-1, 2, 1, 1)
            fuse = w[:, 0:1] * b0 + w[:, 1:2] * b1
            prev = None if ini_stage else w[:, 1, 0, 0]
        if j < 4:
            weights.append(reweigh_gate(sd, f'gate_layer{j}', r, d, temp, exp_noise[j], hard_gate, prev, test))
            skips.append(_skip(sd, f'skip_layer{j}', fuse, training))
        if detail is not None:
            detail[f'fuse{j}'] = fuse
    if detail is not None:
        detail['weights'] = weights
    ctx = pyramid_pooling(sd, 'context_module', fuse, training)
    return decoder(sd, 'decoder', [ctx, skips[2], skips[1], skips[0]], training, cfg)


# --------------------------------------------------------------------------------------------
# callers of the path: loss, eval post-processing, mIoU, schedules
# --------------------------------------------------------------------------------------------
def cross_entropy_2d(logits_scales, target_scales, class_weight):
    """CrossEntropyLoss2d.forward (src/utils.py:34-50).  targets: 0 = void, 1..C = classes."""
    losses = []
    for x, t in zip(logits_scales, target_scales):
        cw = torch.as_tensor(class_weight).to(x.dtype)       # float32 as the reference; float64 for the fp64 truth runs
        per_px = F.cross_entropy(x, t.long() - 1, weight=cw, reduction='none', ignore_index=-1)
        counts = torch.bincount(t.flatten().long(), minlength=len(cw) + 1)
        losses.append(per_px.sum() / (counts[1:] * cw).sum())
    return losses


def validation_losses(logits_batches, target_batches, class_weight, weighted_pixel_sum=None):
    """validate()'s two losses over a validation run (train.py:104-115, :432-440): CrossEntropyLoss2dForValidData
    (src/utils.py:53-74: sum over batches of the weighted CE with reduction='sum', divided by
    weighted_pixel_sum = sum_c pixels_c * w_c of the validation labels) and CrossEntropyLoss2dForValidDataUnweighted
    (:77-97: plain CE sum / number of non-void pixels).  targets: 0 = void."""
    cw = torch.as_tensor(class_weight, dtype=torch.float32)
    tw, tu, npx, wps = 0.0, 0.0, 0, 0.0
    for x, t in zip(logits_batches, target_batches):
        tm = t.long() - 1
        tw = tw + F.cross_entropy(x, tm, weight=cw.to(x.dtype), reduction='sum', ignore_index=-1)
        tu = tu + F.cross_entropy(x, tm, reduction='sum', ignore_index=-1)
        npx += int((tm >= 0).sum())
        wps += float((torch.bincount(t.flatten().long(), minlength=len(cw) + 1)[1:].double() * cw.double()).sum())
    wps = wps if weighted_pixel_sum is None else float(weighted_pixel_sum)
    return float(tw) / wps, float(tu) / npx


def confusion_matrix(label, pred, num_classes):
    """ConfusionMatrixPytorch.update (src/confusion_matrix.py:118-130): rows = label."""
    idx = num_classes * label.long() + pred.long()
    return torch.bincount(idx, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def iou_from_cm(cm):
    """iou_pytorch / miou_pytorch (src/confusion_matrix.py:147-178)."""
    cm = cm.double()
    iou = cm.diag() / (cm.sum(1) + cm.sum(0) - cm.diag() + 1e-15)
    return iou, iou.mean()


def eval_postprocess(logits, label_orig):
    """eval.py:117-134: bilinear resize to label size, argmax, drop void, shift labels by -1."""
    h, w = label_orig.shape[-2:]
    pred = F.interpolate(logits, (h, w), mode='bilinear', align_corners=False).argmax(1)
    mask = label_orig > 0
    return label_orig[mask] - 1, pred[mask]


def exp_decay_temp(start, end, length, epoch):
    """ExpDecayTemp (src/utils.py:203-214): start * b**epoch, b = exp(log(end/start)/len) (b=1 if
    len == 0).  Not clamped: the reference keeps decaying past `length`."""
    import math
    b = 1.0 if length == 0 else math.exp(1.0 / length * math.log(end / start))
    return start * b ** epoch


class OracleNet:
    """Small stateful wrapper (bench.py cpu_baseline and tests): holds a float32 CPU state_dict."""

    def __init__(self, state_dict, cfg: Config):
        self.cfg = cfg
        self.sd = {k: v.detach().clone().cpu() for k, v in state_dict.items()}

    def params(self):
        return {k: v for k, v in self.sd.items()
                if v.dtype.is_floating_point and 'running_' not in k}

    def __call__(self, rgb, depth, **kw):
        return forward(self.sd, rgb, depth, self.cfg, **kw)
