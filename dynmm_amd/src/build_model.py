"""build_model(args, n_classes) -> (model, device): the factory train.py / eval.py call
(FusionDynMM/src/build_model.py:18-218): `--dynamic --global-gate` (SkipGateESANet), `--dynamic` (SkipESANet, per-stage
Gumbel gates) and the static RGB-D ESANet (no --dynamic, --modality rgbd).  ImageNet / SceneNet weights are read from local
files (src/pretrained.py; a missing file is an error, never a silent random init).  The single-modality networks
(`--modality rgb|depth`, ESANetOneModality) are outside the hot path and raise NotImplementedError."""
import warnings

import torch
from torch import nn

from ..nn.esanet import ESANet
from ..nn.net import SkipGateESANet
from ..nn.net_skip import SkipESANet
from .pretrained import load_scenenet


def _decoder_shape(args):
    if 'decreasing' in args.decoder_channels_mode:
        warnings.warn('Argument --channels_decoder is ignored when --decoder_chanels_mode decreasing is set.')
        channels = [512, 256, 128]
    else:
        channels = [args.channels_decoder] * 3
    nb = args.nr_decoder_blocks
    if isinstance(nb, int):
        nb = [nb] * 3
    elif len(nb) == 1:
        nb = list(nb) * 3
    assert len(nb) == 3
    return channels, list(nb)


def build_model(args, n_classes):
    pretrained = bool(args.pretrained_on_imagenet) and not args.last_ckpt and args.pretrained_scenenet == ''
    channels_decoder, nr_decoder_blocks = _decoder_shape(args)
    if args.encoder_depth in (None, 'None'):
        args.encoder_depth = args.encoder
    common = dict(height=args.height, width=args.width, num_classes=n_classes,
                  pretrained_on_imagenet=pretrained, pretrained_dir=args.pretrained_dir,
                  encoder_rgb=args.encoder, encoder_depth=args.encoder_depth, encoder_block=args.encoder_block,
                  activation=args.activation, encoder_decoder_fusion=args.encoder_decoder_fusion,
                  context_module=args.context_module, nr_decoder_blocks=nr_decoder_blocks,
                  channels_decoder=channels_decoder, fuse_depth_in_rgb_encoder=args.fuse_depth_in_rgb_encoder,
                  upsampling=args.upsampling)
    if args.dynamic:
        block_rule = [int(ch) for ch in args.block_rule]
        assert len(block_rule) == 4
        model = (SkipGateESANet if args.global_gate else SkipESANet)(temp=args.temp, block_rule=block_rule, **common)
    elif getattr(args, 'modality', 'rgbd') == 'rgbd':
        model = ESANet(**common)                                          # build_model.py:93-113
    else:
        raise NotImplementedError(f'--modality {args.modality}: the single-modality ESANetOneModality '
                                  '(src/build_model.py:115-141) is not part of the HIP hot path')

    device = torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')
    print('Device:', device)
    model.to(device)

    if getattr(args, 'he_init', False):
        from ..nn.blocks import ResNetEncoder
        # (already initialised: the ImageNet-pre-trained encoders, build_model.py:152-156)
        mods = [m for child in model.children() if not (pretrained and isinstance(child, ResNetEncoder))
                for m in child.modules()]
        for i, m in enumerate(mods):
            if isinstance(m, nn.Conv2d):
                followed_by_sigmoid = isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and \
                    i + 1 < len(mods) and isinstance(mods[i + 1], nn.Identity) and m.bias is not None \
                    and m.out_channels > m.in_channels
                if m.out_channels == n_classes or followed_by_sigmoid or m.groups == m.in_channels:
                    continue
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        print('Applied He init.')

    if getattr(args, 'pretrained_scenenet', '') != '':                   # build_model.py:181-205
        load_scenenet(model, args.pretrained_scenenet, args.context_module)

    if getattr(args, 'finetune', None) is not None:
        ckpt = torch.load(args.finetune, map_location=device)
        model.load_state_dict(ckpt['state_dict'], strict=False)
        print(f'Loaded weights for finetuning: {args.finetune}')
    return model, device
