"""ESANet decoder (model.py:244-410): 3 x [ConvBNAct 3x3, n x NonBottleneck1D, side output (train),
learned 2x upsample + skip add], 3x3 classifier, two learned 2x upsamples."""
import torch
import torch.nn as nn

from .. import ops
from .blocks import ConvBNAct, NonBottleneck1D, chain_ok


class Upsample(nn.Module):
    """'learned-3x3-zeropad': nearest x2 + depthwise 3x3 initialised to mimic bilinear (model.py:385-395)."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1, groups=channels)
        k = torch.tensor([1., 2., 1.]) / 4
        with torch.no_grad():
            self.conv.weight.copy_((k[:, None] * k[None, :]).expand(channels, 1, 3, 3))
            self.conv.bias.zero_()

    def forward(self, x, skip=None):
        return ops.upsample2x_dw3x3(x, self.conv.weight, self.conv.bias, skip)


class DecoderModule(nn.Module):
    def __init__(self, channels_in, channels_dec, nr_decoder_blocks, num_classes):
        super().__init__()
        self.conv3x3 = ConvBNAct(channels_in, channels_dec, 3)
        self.decoder_blocks = nn.Sequential(*[NonBottleneck1D(channels_dec, channels_dec)
                                              for _ in range(nr_decoder_blocks)])
        self.upsample = Upsample(channels_dec)
        self.side_output = nn.Conv2d(channels_dec, num_classes, 1)

    def forward(self, x, skip):
        y = self.conv3x3(x)
        blocks = list(self.decoder_blocks)
        for i, blk in enumerate(blocks):
            # (the chain contract of nn/blocks.py ResNetEncoder._stage: block i is the ONLY consumer of block i - 1's output)
            y = blk(y, chain=i > 0 and chain_ok(blocks[i - 1], blk))
        side = None
        if self.training:
            s = self.side_output
            y, ys = ops.fan_out(y, 2)            # y feeds the side output AND the up-sampling
            side = ops.conv2d(ys, s.weight, s.bias, 1, 0)
        return self.upsample(y, skip), side


class Decoder(nn.Module):
    def __init__(self, channels_in, channels_decoder, nr_decoder_blocks, num_classes):
        super().__init__()
        cd = channels_decoder
        self.decoder_module_1 = DecoderModule(channels_in, cd[0], nr_decoder_blocks[0], num_classes)
        self.decoder_module_2 = DecoderModule(cd[0], cd[1], nr_decoder_blocks[1], num_classes)
        self.decoder_module_3 = DecoderModule(cd[1], cd[2], nr_decoder_blocks[2], num_classes)
        self.conv_out = nn.Conv2d(cd[2], num_classes, 3, padding=1)
        self.upsample1 = Upsample(num_classes)
        self.upsample2 = Upsample(num_classes)
        # engine.TrainStep sets this: in training the full-resolution logits feed the loss only, so the last
        # up-sampling is fused with the cross entropy (ops.DeferredLogits, csrc/tail.hip) and never materialised
        self.defer_tail = False

    def forward(self, enc_outs, unpermute=None):
        """unpermute = (index, inverse): the batch arrives in a permuted (branch-sorted, K16) order; the natural
        order is restored on the 40-channel map in front of the two final up-samplings (3 MB/img instead of
        49 MB/img; everything after it is per-sample) and on the small side outputs."""
        out, s16, s8, s4 = enc_outs
        out, o32 = self.decoder_module_1(out, s16)
        out, o16 = self.decoder_module_2(out, s8)
        out, o8 = self.decoder_module_3(out, s4)
        c = self.conv_out
        out = ops.conv2d(out, c.weight, c.bias, 1, 1)
        if unpermute is not None:
            out = ops.batch_permute(out, *unpermute)
        if self.training and self.defer_tail and torch.is_grad_enabled():
            out = ops.DeferredLogits(self.upsample1(out), self.upsample2.conv)
        else:
            out = self.upsample2(self.upsample1(out))
        if self.training:
            if unpermute is not None:
                o8, o16, o32 = (ops.batch_permute(o, *unpermute) for o in (o8, o16, o32))
            return out, o8, o16, o32
        return out
