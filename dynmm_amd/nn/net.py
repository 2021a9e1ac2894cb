"""Fusion-level DynMM network on the HIP path.

Drop-in for FusionDynMM/src/models/model_skip_mod_globalgate.py: same constructor signature, same
forward contract `(rgb, depth, test=False, return_weight=False)`, same caller-visible attributes
(`baseline, ini_stage, hard_gate, temp, save_weight_info, weight_list, flop, depth_enc_flop,
total_flop`), same methods (`freeze, start_weight, end_weight`) and an identical state_dict
(907 entries for ResNet-34/NonBottleneck1D/SE-add).  Unlike the reference (…globalgate.py:218-223)
construction does not touch a GPU; forward requires a HIP device and libdynmm_hip.so.
"""
import warnings

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .blocks import ConvBNAct, ResNetEncoder, conv_bn_act
from .context import get_context_module
from .decoder import Decoder
from .fusion import SqueezeAndExciteFusionAdd


def DiffSoftmax(logits, tau=1.0, hard=False, dim=-1):
    """Temperature softmax with optional straight-through arg-max (…globalgate.py:20-30) for callers
    that use it stand-alone on [N,5,1,1]/[N,5] logits; the model itself uses the fused gate head."""
    n = logits.shape[0]
    if logits.dim() < 2 or logits.shape[1] != 5 or logits.numel() != 5 * n or dim not in (1, 1 - logits.dim()):
        raise NotImplementedError('HIP DiffSoftmax handles [N,5] / [N,5,1,1] logits along dim 1')
    eye = torch.eye(5, device=logits.device, dtype=torch.float32)
    tab = torch.zeros(5, device=logits.device, dtype=torch.float32)
    w, _, _ = ops.gate_head(logits.reshape(n, 5), eye, tab, tau, hard)
    return w.reshape(logits.shape)


class GlobalGate(nn.Module):
    """cat(rgb, depth) -> conv5x5 s2 -> BN -> tanh -> conv5x5 s2 -> BN -> tanh -> GAP -> 1x1 fc ->
    DiffSoftmax (…globalgate.py:375-394).  The cat is never materialised (dual-input conv)."""

    def __init__(self, branch_num, hidden_dim=8):
        super().__init__()
        if branch_num != 5:
            raise NotImplementedError('the fused gate head implements the 5-branch global gate')
        self.bnum = branch_num
        self.conv = nn.Sequential(
            nn.Conv2d(128, hidden_dim, 5, stride=2), nn.BatchNorm2d(hidden_dim), nn.Identity(),
            nn.Conv2d(hidden_dim, hidden_dim, 5, stride=2), nn.BatchNorm2d(hidden_dim), nn.Identity())
        self.fc = nn.Conv2d(hidden_dim, branch_num, 1, bias=False)

    def features(self, rgb, depth):
        y = conv_bn_act(rgb, self.conv[0], self.conv[1], 'tanh', x2=depth)
        y = conv_bn_act(y, self.conv[3], self.conv[4], 'tanh')
        return ops.adaptive_avg_pool(y, 1)

    def forward(self, rgb, depth, temp=1.0, hard=False, flop_table=None):
        if flop_table is None:
            flop_table = torch.zeros(5, device=rgb.device, dtype=torch.float32)
        weight, _, _ = ops.gate_head(self.features(rgb, depth), self.fc.weight, flop_table, temp, hard)
        return weight


def encoder_stage_pair(model, j, r_in, d_in):
    """Stage j of both encoders, the depth one on a second HIP stream so the kernels' ramp-up /
    store-burst / tail phases of the two independent chains overlap.  Autograd replays each backward
    node on its forward stream, so the backward gets the same concurrency.  The stream is THE depth-encoder
    stream of the device (ops.side_stream(): one per process and device, not one per model — round 5's
    per-instance torch pool stream made every model after the first run 17 % slower, profiles/r06_stream_plan.md)."""
    side = ops.side_stream()
    main = torch.cuda.current_stream()
    capturing = torch.cuda.is_current_stream_capturing()
    # (Round 5 measured letting the depth chain run a fusion AHEAD of the RGB chain — its stream waiting for `main` at stage 1 only,
    # its convolutions beside the fusion kernels: 63.45 against 63.21 ms in lockstep on 4 alternating pairs; two same-shape
    # launches side by side fill the chip better than staggered ones.  Not kept.)
    side.wait_stream(main)
    if not capturing:
        d_in.record_stream(side)         # allocated on `main`, read on the side stream
    with torch.cuda.stream(side):
        d = getattr(model.encoder_depth, f'forward_layer{j}')(d_in)
    r = getattr(model.encoder_rgb, f'forward_layer{j}')(r_in)
    main.wait_stream(side)
    if not capturing:
        d.record_stream(main)            # allocated on the side stream, read by the fusion on `main`
    return r, d


R34_FLOP = [0, 3.27, 7.27, 13.15, 16.02]
R34_DEPTH_ENC_FLOP = [0.2506752, 3.1113216, 6.9470208, 12.66432, 15.538944]
R34_TOTAL_FLOP = [22.37101509, 25.23166149, 29.06736069, 34.78465989, 37.65928389]
OTHER_DEPTH_ENC_FLOP = [0.2506752, 4.39420573, 10.72382115, 19.71582947, 24.679084]
OTHER_TOTAL_FLOP = [32.5854654, 36.728995928, 43.058611352, 52.050619672, 57.0138742]


class SkipGateESANet(nn.Module):
    def __init__(self, height=480, width=640, num_classes=40, encoder_rgb='resnet34',
                 encoder_depth='resnet34', encoder_block='NonBottleneck1D',
                 channels_decoder=None, pretrained_on_imagenet=False,
                 pretrained_dir='./trained_models/imagenet', activation='relu',
                 encoder_decoder_fusion='add', context_module='ppm', nr_decoder_blocks=None,
                 fuse_depth_in_rgb_encoder='add', upsampling='learned-3x3-zeropad', temp=1,
                 block_rule=None):
        super().__init__()
        channels_decoder = [128, 128, 128] if channels_decoder is None else list(channels_decoder)
        nr_decoder_blocks = [3, 3, 3] if nr_decoder_blocks is None else list(nr_decoder_blocks)
        if activation.lower() != 'relu':
            raise NotImplementedError('Only relu is implemented as activation on the HIP path. '
                                      'Got {}'.format(activation))
        if upsampling != 'learned-3x3-zeropad':
            raise NotImplementedError('Only learned-3x3-zeropad upsampling is implemented. Got {}'.format(upsampling))
        if encoder_decoder_fusion != 'add':
            raise NotImplementedError('Only encoder_decoder_fusion="add" is implemented')
        if fuse_depth_in_rgb_encoder not in ('add', 'SE-add'):
            raise NotImplementedError('fuse_depth_in_rgb_encoder must be "add" or "SE-add"')
        self.fuse_depth_in_rgb_encoder = fuse_depth_in_rgb_encoder
        self.block_rule = block_rule if block_rule else [1, 1, 1, 1]
        self.height, self.width = height, width

        self.encoder_rgb = ResNetEncoder(encoder_rgb, encoder_block, input_channels=3)
        self.encoder_depth = ResNetEncoder(encoder_depth, encoder_block, input_channels=1)
        if pretrained_on_imagenet:
            # resnet.py:395-509, from local files (raises FileNotFoundError when they are absent — never a silent
            # random initialisation)
            from ..src.pretrained import load_imagenet_encoder
            load_imagenet_encoder(self.encoder_rgb, encoder_rgb, encoder_block, 3, pretrained_dir)
            load_imagenet_encoder(self.encoder_depth, encoder_depth, encoder_block, 1, pretrained_dir)
        enc = self.encoder_rgb
        self.channels_decoder_in = enc.down_32_channels_out

        if fuse_depth_in_rgb_encoder == 'SE-add':
            for j, ch in enumerate((64, enc.down_4_channels_out, enc.down_8_channels_out,
                                    enc.down_16_channels_out, enc.down_32_channels_out)):
                setattr(self, f'se_layer{j}', SqueezeAndExciteFusionAdd(ch))

        for j, (cin, cout) in enumerate(((enc.down_4_channels_out, channels_decoder[2]),
                                         (enc.down_8_channels_out, channels_decoder[1]),
                                         (enc.down_16_channels_out, channels_decoder[0])), start=1):
            setattr(self, f'skip_layer{j}', nn.Sequential(*([ConvBNAct(cin, cout, 1)] if cin != cout else [])))

        self.context_module, ch_ctx = get_context_module(context_module, self.channels_decoder_in,
                                                         channels_decoder[0])
        self.decoder = Decoder(ch_ctx, channels_decoder, nr_decoder_blocks, num_classes)

        self.temp = temp
        self.gate_layer = GlobalGate(branch_num=5)
        self.baseline = False
        self.ini_stage = False
        self.hard_gate = False
        self.save_weight_info = False
        self.weight_list = torch.Tensor()
        r34 = encoder_rgb == 'resnet34'
        # plain attributes, not buffers (…globalgate.py:217-223) — kept out of the state_dict
        if r34:
            self.flop = torch.tensor(R34_FLOP)
        self.depth_enc_flop = torch.tensor(R34_DEPTH_ENC_FLOP if r34 else OTHER_DEPTH_ENC_FLOP)
        self.total_flop = torch.tensor(R34_TOTAL_FLOP if r34 else OTHER_TOTAL_FLOP)
        self._tab_cache = {}
        # K16 (new capability): in inference with one-hot gate weights, run each depth-encoder stage
        # only on the samples whose branch takes it.  Exact up to fp32 rounding; never used when
        # gradients are recorded or BN is in training mode.
        self.compact = True
        # Opt-in APPROXIMATE compaction while training with one-hot gates (BASELINE configs[3]): BatchNorm batch
        # statistics of depth stage j are taken over the samples that run it, and the straight-through gate
        # gradient of a stage comes from those samples only (the reference's dense forward would also need the
        # depth features of the skipped samples).  RGB path, decoder and all losses are unchanged.
        self.compact_train = False
        self.ini_branches = None          # optional fixed branch per sample for ini_stage (else CPU RNG)
        # benchmark / test knob: with hard gates, a FIXED branch per sample replaces the arg-max decision while the
        # gate network is still evaluated and trained (ops.gate_head force_branch)
        self.branch_override = None
        self._force_cache = None
        self._ini_cache = None
        # Run the depth encoder's stages on a second HIP stream so its kernels' ramp-up / store-burst /
        # tail phases overlap with the RGB encoder's (both encoders are independent between fusion
        # points).  Autograd replays each backward node on its forward stream, so the backward gets
        # the same concurrency.  Off by default until measured.
        self.dual_stream = False
        self.last_stage_batch = None      # depth-stage batch sizes of the last compacted forward

    # ---- caller protocol (train.py:141,190-197,284,351; eval.py:64-68) -------------------------
    def freeze(self):
        for name, param in self.named_parameters():
            if 'gate' not in name:
                param.requires_grad = False

    def start_weight(self):
        self.save_weight_info = True
        self.weight_list = torch.Tensor()

    def end_weight(self, print_each=False, print_flop=False):
        self.save_weight_info = False
        if print_each:
            print(self.weight_list)
        if print_flop and self.weight_list.numel():
            counts = np.array([(self.weight_list[:, i] == 1).sum().item() for i in range(5)], dtype=float)
            frac = torch.from_numpy(counts / max(counts.sum(), 1.0)).float()
            flop1 = (self.depth_enc_flop.cpu() * frac).sum()
            flop2 = (self.total_flop.cpu() * frac).sum()
            print(f'Depth Encoder Flop {flop1:.4f}G | Total Flop {flop2:.4f}G')
        self.weight_list = torch.Tensor()

    def _flop_table(self, device):
        key = str(device)
        if key not in self._tab_cache:
            self._tab_cache[key] = self.depth_enc_flop.detach().to(device=device, dtype=torch.float32).contiguous()
        return self._tab_cache[key]

    def _force_tensor(self, branches, device):
        key = (tuple(branches), str(device))
        if self._force_cache is None or self._force_cache[0] != key:
            self._force_cache = (key, torch.tensor(branches, dtype=torch.int32, device=device))
        return self._force_cache[1]

    def _se(self, j):
        return getattr(self, f'se_layer{j}').params8() if self.fuse_depth_in_rgb_encoder == 'SE-add' else None

    # ---- forward ------------------------------------------------------------------------------
    def _stem(self, rgb, depth):
        er, ed = self.encoder_rgb, self.encoder_depth
        if self.training:
            ops.begin_step()
        if ops.stem_bn_fuse_supported((rgb.shape[2] + 1) // 2, (rgb.shape[3] + 1) // 2, er.bn1, ed.bn1):
            # training: stem BatchNorm + ReLU are applied on load by the fusion / pooling kernels (never written)
            c_r, c_d = er.conv1, ed.conv1
            # (bn_stats: the batch statistics of bn1 come out of the stem convolution's epilogue where the kernel can;
            #  stem_bn_defer + this order — depth stem first — make the backward run BN backward rgb -> weight gradient rgb
            #  (asynchronous, the longer one) -> BN backward depth -> weight gradient depth at the end of the step)
            d = ops.stem_bn_defer(ops.conv2d(depth, c_d.weight, c_d.bias, c_d.stride, c_d.padding, bn_stats=True), ed.bn1)
            r = ops.stem_bn_defer(ops.conv2d(rgb, c_r.weight, c_r.bias, c_r.stride, c_r.padding, bn_stats=True), er.bn1)
            r, d = ops.stem_bn_fuse_pool(r, er.bn1, d, ed.bn1, self._se(0))
        else:
            r, d = self._stem_unfused(rgb, depth)
        return r, d

    def forward(self, rgb, depth, test=False, return_weight=False):
        st = self.forward_front(rgb, depth)
        if self.save_weight_info:
            self.weight_list = torch.cat((self.weight_list, st['weight'].detach().cpu()))
        return self.forward_back(st, self.stage_counts(st), test, return_weight)


    def _stem_unfused(self, rgb, depth):
        er, ed = self.encoder_rgb, self.encoder_depth
        r = er.forward_first_conv(rgb)
        d = ed.forward_first_conv(depth)
        if ops.se_fuse_pool_supported(r):
            # stem fusion (always on) + both max-pools as one pass; the full-resolution fused map is never written
            r, d = ops.se_fuse_pool(r, d, self._se(0))
        else:
            d, d_pool = ops.fan_out(d, 2)                           # depth stem output: stem fusion + its own max-pool
            fuse = ops.se_fuse_blend(r, d, self._se(0))
            r = ops.max_pool_3x3_s2(fuse)
            d = ops.max_pool_3x3_s2(d_pool)
        return r, d

    def forward_front(self, rgb, depth):
        """Stem, gate and the DEVICE side of the compaction decision — everything up to the one point where the host may have to
        read 16 bytes (stage_counts).  No host synchronisation in here: engine.InferStep captures it as the first of two graphs."""
        tab = self._flop_table(rgb.device)
        r, d = self._stem(rgb, depth)
        bs = r.shape[0]
        host_branch = None                                           # branch per sample when the host already knows it
        if self.baseline:                                            # …globalgate.py:264-266
            onehot = torch.zeros(bs, 5, device=rgb.device)
            onehot[:, 4] = 1
            weight, wcum, loss = ops.gate_from_weight(onehot, tab)
            host_branch = [4] * bs
        elif self.ini_stage:                                         # …globalgate.py:267-270 (CPU RNG)
            if self.ini_branches is None:
                idx = torch.randint(0, 5, (bs,))
                onehot = torch.zeros(bs, 5)
                onehot[torch.arange(bs), idx] = 1
                onehot = onehot.to(rgb.device)
            else:
                # injected branches (tests, bench): the one-hot rows are made resident once per distribution, so that a
                # forward with them holds no host -> device copy (engine.InferStep captures it)
                idx = torch.as_tensor(self.ini_branches)[:bs]
                key = (tuple(int(v) for v in idx.tolist()), str(rgb.device))
                if self._ini_cache is None or self._ini_cache[0] != key:
                    onehot = torch.zeros(bs, 5)
                    onehot[torch.arange(bs), idx] = 1
                    self._ini_cache = (key, onehot.to(rgb.device))
                onehot = self._ini_cache[1]
            weight, wcum, loss = ops.gate_from_weight(onehot, tab)
            host_branch = [int(v) for v in idx.tolist()]
        else:
            force = None
            if self.branch_override is not None and self.hard_gate:
                host_branch = [int(v) for v in list(self.branch_override)[:bs]]
                force = self._force_tensor(host_branch, rgb.device)
            (r, r_gate), (d, d_gate) = ops.fan_out(r, 2), ops.fan_out(d, 2)    # gate convs + first encoder stage
            # (the gate's kernels on a stream of their own beside encoder stage 1 — they are first read by the fusion at the END of
            #  stage 1 — were measured in rounds 3 and 5: +-0.1 ms then, 73.4 against 62.95 ms per step now, 4 alternating pairs (the
            #  cause: a FIFTH busy stream, see ops.WGRAD_STREAMS; on an existing stream — the last weight-gradient one — 64.05 against 63.77).  Not kept.)
            pooled = self.gate_layer.features(r_gate, d_gate)
            weight, wcum, loss = ops.gate_head(pooled, self.gate_layer.fc.weight, tab, self.temp, self.hard_gate, force)
        one_hot = self.baseline or self.ini_stage or self.hard_gate
        infer = not self.training and not torch.is_grad_enabled()
        compacted = one_hot and ((self.compact and infer) or (self.compact_train and not infer))
        st = {'r': r, 'd': d, 'weight': weight, 'wcum': wcum, 'loss': loss, 'host_branch': host_branch, 'infer': infer,
              'compacted': compacted, 'order': None, 'inv': None, 'counts_dev': None}
        if compacted:
            # K16: sort the batch by branch (descending, stable) ON THE DEVICE; the samples that still need depth
            # stage j are then the prefix of length counts[j-1] — each stage runs on a prefix view.  The host needs
            # only the 4 counts: known already for baseline / ini_stage / an injected distribution, otherwise ONE
            # 16-byte device->host read per forward (stage_counts).
            _, st['order'], st['inv'], st['counts_dev'] = ops.gate_decide(weight)
        return st

    def stage_counts(self, st):
        """Samples that run depth stage 1..4 of a compacted forward (None otherwise): from the host's own knowledge of the branches
        (baseline / ini_stage / an injected distribution) or by ONE 16-byte device->host read."""
        if not st['compacted']:
            return None
        hb = st['host_branch']
        if hb is not None:
            return [sum(1 for bch in hb if bch >= j) for j in (1, 2, 3, 4)]
        return [int(v) for v in st['counts_dev'].tolist()]

    def forward_back(self, st, counts, test=False, return_weight=False):
        """The four encoder stages with their fusions, context module and decoder, for a front state and its stage counts."""
        er, ed = self.encoder_rgb, self.encoder_depth
        r, d, weight, wcum, loss = st['r'], st['d'], st['weight'], st['wcum'], st['loss']
        host_branch, infer, compacted, order, inv = st['host_branch'], st['infer'], st['compacted'], st['order'], st['inv']
        self.last_stage_batch = None
        skips = []
        unpermute = None
        if compacted:
            presorted = host_branch is not None and all(x >= y for x, y in zip(host_branch, host_branch[1:]))
            wc = None
            if not presorted:
                r, d = ops.batch_permute(r, order, inv), ops.batch_permute(d, order, inv)
                unpermute = (inv, order)
            if not infer and wcum.requires_grad:
                wc = wcum if presorted else ops.batch_permute(wcum, order, inv)   # straight-through gate gradient
            self.last_stage_batch = list(counts)
            for j in (1, 2, 3, 4):
                c = counts[j - 1]
                r_in = r if j == 1 else fuse
                if c > 0:
                    d_in = d if d.shape[0] == c else d[:c]
                    if self.dual_stream:
                        r, d = encoder_stage_pair(self, j, r_in, d_in)
                    else:
                        r = getattr(er, f'forward_layer{j}')(r_in)
                        d = getattr(ed, f'forward_layer{j}')(d_in)
                    fuse = ops.se_fuse_blend(r, d, self._se(j), wc, j - 1, inplace=True)
                else:
                    r = getattr(er, f'forward_layer{j}')(r_in)
                    fuse, d = r, None                # every sample skips depth from here on
                if j < 4:
                    sk = getattr(self, f'skip_layer{j}')
                    skips.append(sk[0](fuse) if len(sk) else fuse)
        else:
            wcs = ops.fan_out(wcum, 4)               # one cumulative-weight column per stage
            for j in (1, 2, 3, 4):
                if self.dual_stream:
                    r, d = encoder_stage_pair(self, j, r if j == 1 else fuse, d)
                else:
                    r = getattr(er, f'forward_layer{j}')(r if j == 1 else fuse)
                    d = getattr(ed, f'forward_layer{j}')(d)
                d_f = d
                if j < 4:
                    d_f, d = ops.fan_out(d, 2)       # stage-j depth features: fusion + next depth stage
                # stage j<4: w*rgb + (1-w)*fused with w = sum_{k<j} weight[:,k];  stage 4: w = 1-weight[:,4]
                fuse = ops.se_fuse_blend(r, d_f, self._se(j), wcs[j - 1], j - 1)
                if j < 4:
                    fuse, f_skip = ops.fan_out(fuse, 2)   # fused map: next RGB stage + decoder skip connection
                    sk = getattr(self, f'skip_layer{j}')
                    skips.append(sk[0](f_skip) if len(sk) else f_skip)
        out = self.context_module(fuse)
        out = self.decoder([out, skips[2], skips[1], skips[0]], unpermute=unpermute)

        if test:
            return (out, weight) if return_weight else out
        return out, loss
