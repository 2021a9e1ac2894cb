"""Counterpart of FusionDynMM/src/models/resnet.py (ResNet-18/34 trunks used by the hot path)."""
from ...nn.blocks import BasicBlock, NonBottleneck1D, ResNetEncoder as ResNet  # noqa: F401


def _make(name, block='BasicBlock', pretrained_on_imagenet=False, pretrained_dir=None,
          input_channels=3, activation=None):
    if pretrained_on_imagenet:
        raise NotImplementedError('offline build: load weights with load_state_dict instead')
    return ResNet(name, block if isinstance(block, str) else block.__name__, input_channels)


def ResNet18(**kw):
    return _make('resnet18', **kw)


def ResNet34(**kw):
    return _make('resnet34', **kw)


def ResNet50(**kw):
    raise NotImplementedError('ResNet50/Bottleneck is outside the HIP hot path (north_star fixes ResNet-34)')
