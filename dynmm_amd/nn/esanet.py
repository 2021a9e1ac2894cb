"""ESANet — the STATIC RGB-D network (FusionDynMM/src/models/model.py:19-241): depth features are fused into the RGB
encoder at the stem and after every stage, no gate.  It is what `build_model` returns without `--dynamic`
(src/build_model.py:93-113) and what the README's baseline mIoU was produced with.

On the HIP path it is SkipGateESANet's forward with the gate pinned to "fuse everywhere" (the blend of
model_skip_mod_globalgate.py:282-310 with weight = e_4 is exactly model.py:196-238), minus the gate parameters: same
kernels, the reference's own state_dict (892 entries for ResNet-34 / NonBottleneck1D / SE-add, checked against the
reference in tests/test_esanet.py), the reference's forward contract `forward(rgb, depth) -> out` where `out` is the
decoder's 4-tuple in training mode (model.py:306-308)."""
from .net import SkipGateESANet


class ESANet(SkipGateESANet):
    def __init__(self, height=480, width=640, num_classes=37, encoder_rgb='resnet18', encoder_depth='resnet18',
                 encoder_block='BasicBlock', channels_decoder=None, pretrained_on_imagenet=True,
                 pretrained_dir='./trained_models/imagenet', activation='relu', encoder_decoder_fusion='add',
                 context_module='ppm', nr_decoder_blocks=None, fuse_depth_in_rgb_encoder='SE-add',
                 upsampling='bilinear'):
        # the reference's defaults (model.py:20-35), including the ones the HIP path refuses loudly (bilinear up-sampling)
        super().__init__(height=height, width=width, num_classes=num_classes, encoder_rgb=encoder_rgb,
                         encoder_depth=encoder_depth, encoder_block=encoder_block, channels_decoder=channels_decoder,
                         pretrained_on_imagenet=pretrained_on_imagenet, pretrained_dir=pretrained_dir,
                         activation=activation, encoder_decoder_fusion=encoder_decoder_fusion,
                         context_module=context_module,
                         nr_decoder_blocks=[1, 1, 1] if nr_decoder_blocks is None else nr_decoder_blocks,
                         fuse_depth_in_rgb_encoder=fuse_depth_in_rgb_encoder, upsampling=upsampling)
        del self.gate_layer                      # no gate parameters: the reference ESANet has none
        self.baseline = True

    def forward(self, rgb, depth, test=True, return_weight=False):
        # model.py:189-241: every stage fuses; SkipGateESANet with baseline = True computes the same blend with w = e_4.
        # The reference's ESANet.forward takes (rgb, depth) only; the two extra arguments are what train.py's `validate`
        # / eval.py pass to the dynamic models (train.py:438, eval.py:106) and are accepted so that engine.evaluate and
        # those callers work on either model: there is no gate, hence no FLOP loss to return and `weight` is e_4.
        self.baseline = True
        if return_weight:
            return super().forward(rgb, depth, test=True, return_weight=True)
        return super().forward(rgb, depth, test=True)

    def forward_front(self, rgb, depth):
        self.baseline = True                 # (engine.InferStep enters here, not through forward)
        return super().forward_front(rgb, depth)

    def freeze(self):
        raise NotImplementedError('ESANet has no gate to keep trainable (model.freeze() belongs to the --dynamic models)')
