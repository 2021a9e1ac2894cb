"""Weight loading of the reference's build / train protocol, from LOCAL files (there is no network on the box):

  load_imagenet_encoder   FusionDynMM/src/models/resnet.py:395-466 (torchvision ResNet weights: BasicBlock / Bottleneck
                          trunks) and :469-509 (`load_pretrained_with_different_encoder_block`: the authors' ImageNet
                          pre-training of the NonBottleneck1D trunks, `<pretrained_dir>/r34_NBt1D.pth`, keys re-mapped);
  load_scenenet           src/build_model.py:181-205 (--pretrained_scenenet);
  load_ckpt               src/utils.py:145-175 + train.py:131-135 (--last_ckpt: model, optimizer state, epoch, best mIoU).

Everything fails LOUDLY when a file is missing — a run that asked for pre-trained weights never continues from a random
initialisation."""
import os
from collections import OrderedDict

import torch

# file names of the torchvision checkpoints the reference downloads (resnet.py:24-28, `model_dir='./'`)
TORCHVISION_FILES = {'resnet18': 'resnet18-5c106cde.pth', 'resnet34': 'resnet34-333f7ec4.pth',
                     'resnet50': 'resnet50-19c8e357.pth'}
NBT1D_NAMES = {'resnet18': 'r18', 'resnet34': 'r34'}


def _load(path):
    return torch.load(path, map_location='cpu')


def _find(name, pretrained_dir):
    for d in ('.', pretrained_dir, os.environ.get('TORCH_HOME', ''), os.path.join(os.path.expanduser('~'), '.cache', 'torch', 'hub', 'checkpoints')):
        if d and os.path.isfile(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def load_imagenet_encoder(encoder, resnet_name, encoder_block, input_channels, pretrained_dir='./trained_models/imagenet'):
    """In place.  `encoder`: dynmm_amd.nn.blocks.ResNetEncoder (keys == the reference ResNet's)."""
    if resnet_name == 'resnet50' or encoder_block == 'BasicBlock':
        fname = TORCHVISION_FILES[resnet_name]
        path = _find(fname, pretrained_dir)
        if path is None:
            raise FileNotFoundError(
                f'ImageNet weights requested (pretrained_on_imagenet) but {fname} is not present in ./ or {pretrained_dir} '
                '(the reference downloads it from download.pytorch.org; this machine has no network). Copy the file '
                'there or pass --no_imagenet_pretraining.')
        weights = _load(path)
        if input_channels == 1:                       # resnet.py:403-406: sum the first convolution over RGB
            weights['conv1.weight'] = torch.sum(weights['conv1.weight'], dim=1, keepdim=True)
        weights.pop('fc.weight', None)
        weights.pop('fc.bias', None)
        encoder.load_state_dict(weights, strict=True)
        print(f'Loaded {resnet_name} pretrained on ImageNet')
        return path
    # NonBottleneck1D trunks: the authors' own ImageNet checkpoints (resnet.py:469-509)
    short = NBT1D_NAMES[resnet_name]
    path = os.path.join(pretrained_dir, f'{short}_NBt1D.pth')
    if not os.path.exists(path):
        logs = os.path.join(pretrained_dir, 'logs.csv')
        if not os.path.exists(logs):
            raise FileNotFoundError(
                f'ImageNet weights requested for {resnet_name} / NonBottleneck1D but neither {path} nor {logs} exists '
                '(see the reference README: trained_models/imagenet). Pass --no_imagenet_pretraining to train from scratch.')
        import pandas as pd
        tab = pd.read_csv(logs)
        idx = tab['acc_val_top-1'].idxmax()
        path = os.path.join(pretrained_dir, 'ckpt_epoch_{}.pth'.format(tab.epoch[idx]))
        print(f"Choosing checkpoint {path} with top1 acc {tab['acc_val_top-1'][idx]}")
    ckpt = _load(path)
    weights = OrderedDict()
    for key, val in ckpt['state_dict'].items():       # 'encoder.<k>' -> '<k>'; the classifier head has no 'encoder' in its name
        if 'encoder' in key:
            weights[key.split('encoder.')[-1]] = val
    if input_channels == 1:
        weights['conv1.weight'] = torch.sum(weights['conv1.weight'], dim=1, keepdim=True)
    encoder.load_state_dict(weights, strict=False)
    print(f'Loaded {short} with encoder block {encoder_block} pretrained on ImageNet')
    print(path)
    return path


def load_scenenet(model, path, context_module='ppm'):
    """src/build_model.py:181-205: every weight of a SceneNet-pre-trained network except the (side) outputs and the two
    final learned up-samplings (their class count differs)."""
    if not os.path.isfile(path):
        raise FileNotFoundError(f'--pretrained_scenenet {path}: no such file')
    weights = _load(path)['state_dict']
    ignore = [k for k in weights if 'out' in k or 'decoder.upsample1' in k or 'decoder.upsample2' in k]
    if context_module not in ('ppm', 'appm'):
        ignore.extend(k for k in weights if 'context_module.features' in k)
    for k in ignore:
        weights.pop(k, None)
    sd = model.state_dict()
    sd.update(weights)
    model.load_state_dict(sd)
    print(f'Loaded pretrained SceneNet weights: {path}')


def load_ckpt(model, optimizer, model_file, device=None):
    """src/utils.py:145-175.  `optimizer`: anything with load_state_dict (engine's fused optimizers, torch.optim) or None.
    Returns (epoch, best_miou, best_miou_epoch)."""
    if not os.path.isfile(model_file):
        raise FileNotFoundError("=> no checkpoint found at '{}'".format(model_file))
    print("=> loading checkpoint '{}'".format(model_file))
    ckpt = _load(model_file)
    model.load_state_dict(ckpt['state_dict'])
    if optimizer is not None:
        optimizer.load_state_dict(ckpt['optimizer'])
    print("=> loaded checkpoint '{}' (epoch {})".format(model_file, ckpt['epoch']))
    return ckpt['epoch'], ckpt.get('best_miou', 0), ckpt.get('best_miou_epoch', 0)
