"""SkipESANet — the per-stage Gumbel-gated variant (FusionDynMM/src/models/model_skip_mod.py:20-324)
on the HIP path.

Same constructor signature, `forward(rgb, depth, test=False)`, caller-visible attributes
(`hard_gate, ini_stage, random_policy, save_weight_info, weight_list, block_rule, temp`), methods
(`freeze, start_weight, end_weight`) and state_dict as the reference.  Each fusion point is ONE fused op
(ops.reweigh_fuse): the 2-way blend of the stage with the previous gate's weights and the next gate
evaluated on the same two feature maps (GAP shared, feature maps read once forward).

Gumbel noise: the reference draws `-log(Exp(1))` from torch's global generator
(torch.nn.functional.gumbel_softmax); here the gate kernel draws it with Philox4x32-10 keyed by
(ops.manual_seed, call counter, sample) — parity is exact given the same noise (`noise=` hook used by the
tests) and distributional otherwise.
"""
import warnings

import torch
import torch.nn as nn

from .. import ops
from .blocks import ConvBNAct, ResNetEncoder
from .context import get_context_module
from .decoder import Decoder
from .fusion import SqueezeAndExciteFusionAdd, SqueezeAndExciteReweigh
from .net import encoder_stage_pair


class SkipESANet(nn.Module):
    def __init__(self, height=480, width=640, num_classes=37, encoder_rgb='resnet18',
                 encoder_depth='resnet18', encoder_block='BasicBlock', channels_decoder=None,
                 pretrained_on_imagenet=False, pretrained_dir='./trained_models/imagenet',
                 activation='relu', encoder_decoder_fusion='add', context_module='ppm',
                 nr_decoder_blocks=None, fuse_depth_in_rgb_encoder='SE-add',
                 upsampling='learned-3x3-zeropad', temp=1, block_rule=None):
        super().__init__()
        channels_decoder = [128, 128, 128] if channels_decoder is None else list(channels_decoder)
        nr_decoder_blocks = [1, 1, 1] if nr_decoder_blocks is None else list(nr_decoder_blocks)
        if activation.lower() != 'relu':
            raise NotImplementedError('Only relu is implemented as activation on the HIP path. '
                                      'Got {}'.format(activation))
        if upsampling != 'learned-3x3-zeropad':
            raise NotImplementedError('Only learned-3x3-zeropad upsampling is implemented. Got {}'.format(upsampling))
        if encoder_decoder_fusion != 'add':
            raise NotImplementedError('Only encoder_decoder_fusion="add" is implemented')
        self.fuse_depth_in_rgb_encoder = fuse_depth_in_rgb_encoder
        self.block_rule = block_rule if block_rule else [1, 1, 1, 1]
        self.height, self.width = height, width

        self.encoder_rgb = ResNetEncoder(encoder_rgb, encoder_block, input_channels=3)
        self.encoder_depth = ResNetEncoder(encoder_depth, encoder_block, input_channels=1)
        if pretrained_on_imagenet:                    # resnet.py:395-509, local files only (loud when absent)
            from ..src.pretrained import load_imagenet_encoder
            load_imagenet_encoder(self.encoder_rgb, encoder_rgb, encoder_block, 3, pretrained_dir)
            load_imagenet_encoder(self.encoder_depth, encoder_depth, encoder_block, 1, pretrained_dir)
        enc = self.encoder_rgb
        self.channels_decoder_in = enc.down_32_channels_out
        stage_ch = (64, enc.down_4_channels_out, enc.down_8_channels_out, enc.down_16_channels_out,
                    enc.down_32_channels_out)

        if fuse_depth_in_rgb_encoder == 'SE-add':
            # constructed (and checkpointed) by the reference but never used by its forward
            # (model_skip_mod.py:113-130 vs :235-311) — kept for state_dict parity only
            for j, ch in enumerate(stage_ch):
                setattr(self, f'se_layer{j}', SqueezeAndExciteFusionAdd(ch))

        self.temp = temp
        for j in range(4):
            setattr(self, f'gate_layer{j}', SqueezeAndExciteReweigh(self.temp, stage_ch[j]))

        for j, (cin, cout) in enumerate(((enc.down_4_channels_out, channels_decoder[2]),
                                         (enc.down_8_channels_out, channels_decoder[1]),
                                         (enc.down_16_channels_out, channels_decoder[0])), start=1):
            setattr(self, f'skip_layer{j}', nn.Sequential(*([ConvBNAct(cin, cout, 1)] if cin != cout else [])))

        self.context_module, ch_ctx = get_context_module(context_module, self.channels_decoder_in,
                                                         channels_decoder[0])
        self.decoder = Decoder(ch_ctx, channels_decoder, nr_decoder_blocks, num_classes)

        self.hard_gate = False
        self.ini_stage = False
        self.random_policy = False
        self.save_weight_info = False
        self.weight_list = [torch.Tensor() for _ in range(4)]
        self.gumbel_noise = None      # optional list of 4 [N,2] Exp(1) tensors (tests / reproducibility)
        self.last_aux = None          # gate internals of the last forward (w, ysoft, y1, E) per stage
        self.dual_stream = False      # depth encoder on a second HIP stream (see nn/net.py)
        # Inference with hard gates and block_rule 2222: the chained weights (b1_j = y1_j * b1_{j-1}) make a
        # skip permanent, so the depth encoder (and the later gates) run only on the samples still fusing.
        # Exact up to fp32 rounding of the straight-through one-hots; never used when gradients are recorded.
        self.compact = True
        self.last_stage_batch = None

    # ---- caller protocol ------------------------------------------------------------------------
    def freeze(self):
        for name, param in self.named_parameters():
            if 'gate' not in name:
                param.requires_grad = False

    def start_weight(self):
        self.save_weight_info = True
        self.weight_list = [torch.Tensor() for _ in range(4)]

    def end_weight(self, print_each=False, thre=None):
        self.save_weight_info = False
        avg = []
        for i in range(4):
            if self.block_rule[i] != 2:
                continue
            if thre:
                print('-' * 40, 'layer ', i, '-' * 40)
                cnt1 = (self.weight_list[i][:, 0] < thre).sum()
                cnt2 = (self.weight_list[i][:, 1] < thre).sum()
                print(f'Skip {cnt1} branch 1 | {cnt2} branch 2')
            weight_mean = torch.mean(self.weight_list[i], axis=0)
            if print_each:
                print(self.weight_list[i])
                print(weight_mean)
            avg.append(weight_mean)
        self.weight_list = [torch.Tensor() for _ in range(4)]
        return avg

    # ---- forward --------------------------------------------------------------------------------
    def _fuse(self, j, r, d, wblend, mode, prev, test, record=True):
        """Fusion point j (0 = stem … 4): blend with `wblend`, and for j < 4 evaluate gate j."""
        if j == 4:
            fuse, _, _ = ops.reweigh_fuse(r, d, wblend, mode)
            return fuse, None
        gate = getattr(self, f'gate_layer{j}')
        if self.random_policy:
            fuse, _, _ = ops.reweigh_fuse(r, d, wblend, mode)
            w = gate.random_weights(r.shape[0], r.device, prev)
            self.last_aux[j] = None
        else:
            noise = None if self.gumbel_noise is None else self.gumbel_noise[j]
            fuse, w, aux = ops.reweigh_fuse(r, d, wblend, mode, gate.se.mlp_params(), gate.temp,
                                            self.hard_gate or test, prev, noise)
            self.last_aux[j] = aux
        if self.save_weight_info and record:
            self.weight_list[j] = torch.cat((self.weight_list[j], w.detach().cpu()))
        return fuse, w

    def _forward_compact(self, rgb, depth, test):
        er, ed = self.encoder_rgb, self.encoder_depth
        dev = rgb.device
        r = er.forward_first_conv(rgb)
        d = ed.forward_first_conv(depth)
        fuse, w = self._fuse(0, r, d, None, 1, None, test, record=False)
        r = ops.max_pool_3x3_s2(fuse)
        d = ops.max_pool_3x3_s2(d)
        bs = r.shape[0]
        alive = list(range(bs))            # samples whose depth features are still computed (rows of d)
        weights = [w]
        self.last_stage_batch = []
        skips = []
        for j in (1, 2, 3, 4):
            r = getattr(er, f'forward_layer{j}')(r if j == 1 else fuse)
            take = (weights[-1][:, 1] > 0.5).tolist()                 # host sync: this stage's decisions
            keep = [pos for pos, n in enumerate(alive) if take[n]]
            if keep and len(keep) < len(alive):
                d = ops.batch_gather(d, torch.tensor(keep, dtype=torch.int32, device=dev))
            alive = [alive[pos] for pos in keep]
            self.last_stage_batch.append(len(alive))
            wj = torch.zeros(bs, 2, device=dev)
            wj[:, 0] = 1.0                                             # skipped samples: (b0, b1) = (1, 0) for good
            if not alive:
                fuse, d = r, None
            else:
                d = getattr(ed, f'forward_layer{j}')(d)
                full = len(alive) == bs
                idx = None if full else torch.tensor(alive, dtype=torch.long, device=dev)
                r_a = r if full else ops.batch_gather(r, idx.to(torch.int32))
                if j < 4:
                    gate = getattr(self, f'gate_layer{j}')
                    noise = None if self.gumbel_noise is None else self.gumbel_noise[j]
                    if noise is not None and not full:
                        noise = noise[idx].contiguous()
                    fused, w_a, _ = ops.reweigh_fuse(r_a, d, None, 1, gate.se.mlp_params(), gate.temp, True, None, noise)
                    if full:
                        wj = w_a
                    else:
                        wj[idx] = w_a
                else:
                    fused, _, _ = ops.reweigh_fuse(r_a, d, None, 1)
                if full:
                    fuse = fused
                else:
                    mapping = torch.full((bs,), -1, dtype=torch.int32)
                    mapping[alive] = torch.arange(len(alive), dtype=torch.int32)
                    fuse = ops.batch_merge(r, fused, mapping.to(dev))
            if j < 4:
                weights.append(wj)
                sk = getattr(self, f'skip_layer{j}')
                skips.append(sk[0](fuse) if len(sk) else fuse)
        if self.save_weight_info:
            for j in range(4):
                self.weight_list[j] = torch.cat((self.weight_list[j], weights[j].detach().cpu()))
        out = self.context_module(fuse)
        return self.decoder([out, skips[2], skips[1], skips[0]])

    def forward(self, rgb, depth, test=False):
        er, ed = self.encoder_rgb, self.encoder_depth
        self.last_aux = [None] * 4
        self.last_stage_batch = None
        if self.training:
            ops.begin_step()
        if (self.compact and (test or self.hard_gate) and not self.training and not torch.is_grad_enabled()
                and not self.ini_stage and not self.random_policy and list(self.block_rule) == [2, 2, 2, 2]):
            return self._forward_compact(rgb, depth, test)
        r = er.forward_first_conv(rgb)
        d = ed.forward_first_conv(depth)
        fuse, w = self._fuse(0, r, d, None, 1, None, test)          # stem: rgb + depth; gate 0
        r = ops.max_pool_3x3_s2(fuse)
        d = ops.max_pool_3x3_s2(d)

        prev = None
        skips = []
        for j in (1, 2, 3, 4):
            if self.dual_stream:
                r, d = encoder_stage_pair(self, j, r if j == 1 else fuse, d)
            else:
                r = getattr(er, f'forward_layer{j}')(r if j == 1 else fuse)
                d = getattr(ed, f'forward_layer{j}')(d)
            rule = self.block_rule[j - 1]
            # gate j is chained on `prev` as it stands BEFORE this stage updates it for j >= 2, but
            # AFTER the update for the stage's own weight (model_skip_mod.py:252-256, 271-275)
            mode = rule if rule in (0, 1) else 2
            wb = w if mode == 2 else None
            if mode == 2 and not self.ini_stage:
                prev = w[:, 1]
            fuse, w = self._fuse(j, r, d, wb, mode, prev, test)
            if j < 4:
                sk = getattr(self, f'skip_layer{j}')
                skips.append(sk[0](fuse) if len(sk) else fuse)
        out = self.context_module(fuse)
        return self.decoder([out, skips[2], skips[1], skips[0]])
