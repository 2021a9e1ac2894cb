"""CLI flags of the fusion-level path — same names, types and defaults as
FusionDynMM/src/args.py:9-197 (eval.py:41-46 adds its own), kept as a table."""
import argparse

# (flags, kwargs)
_COMMON = [
    (('--results_dir',), dict(default='./results')),
    (('--last_ckpt',), dict(default='', type=str)),
    (('--pretrained_dir',), dict(default='./trained_models/imagenet')),
    (('--pretrained_scenenet',), dict(default='')),
    (('--no_imagenet_pretraining',), dict(dest='pretrained_on_imagenet', default=True, action='store_false')),
    (('--finetune',), dict(default=None, type=str)),
    (('--batch_size',), dict(type=int, default=8)),
    (('--batch_size_valid',), dict(type=int, default=None)),
    (('--height',), dict(type=int, default=480)),
    (('--width',), dict(type=int, default=640)),
    (('--epochs',), dict(type=int, default=500)),
    (('--lr', '--learning-rate'), dict(type=float, default=0.01)),
    (('--weight_decay', '--wd'), dict(type=float, default=1e-4)),
    (('--momentum',), dict(type=float, default=0.9)),
    (('--optimizer',), dict(type=str, default='SGD', choices=['SGD', 'Adam'])),
    (('--class_weighting',), dict(type=str, default='median_frequency',
                                  choices=['median_frequency', 'logarithmic', 'None'])),
    (('--c_for_logarithmic_weighting',), dict(type=float, default=1.02)),
    (('--he_init',), dict(dest='he_init', default=False, action='store_true')),
    (('--valid_full_res',), dict(default=False, action='store_true')),
    # DynMM (args.py:87-100); README spells some with underscores (README.md:80-88) — accept both
    (('--dynamic',), dict(action='store_true')),
    (('--global-gate', '--global_gate'), dict(action='store_true', dest='global_gate')),
    (('--block-rule', '--block_rule'), dict(type=str, default='1111', dest='block_rule')),
    (('--temp',), dict(type=float, default=1)),
    (('--end-temp', '--end_temp'), dict(type=float, default=0.001, dest='end_temp')),
    (('--loss-ratio', '--loss_ratio'), dict(type=float, default=0.0, dest='loss_ratio')),
    (('--flop-budget', '--flop_budget'), dict(type=float, default=0.0, dest='flop_budget')),
    (('--epoch-ini', '--epoch_ini'), dict(type=int, default=0, dest='epoch_ini')),
    (('--epoch-hard', '--epoch_hard'), dict(type=int, default=500, dest='epoch_hard')),
    (('--eval-every', '--eval_every'), dict(type=int, default=2, dest='eval_every')),
    (('--save-every', '--save_every'), dict(type=int, default=100, dest='save_every')),
    (('--baseline',), dict(action='store_true')),
    (('--freeze',), dict(action='store_true')),
    (('--soft-eval', '--soft_eval'), dict(action='store_true', dest='soft_eval')),
    # model
    (('--activation',), dict(type=str, default='relu', choices=['relu', 'swish', 'hswish'])),
    (('--encoder',), dict(type=str, default='resnet50', choices=['resnet18', 'resnet34', 'resnet50'])),
    (('--encoder_block',), dict(type=str, default='BasicBlock', choices=['BasicBlock', 'NonBottleneck1D'])),
    (('--nr_decoder_blocks',), dict(type=int, default=[3], nargs='+')),
    (('--encoder_depth',), dict(type=str, default=None, choices=['resnet18', 'resnet34', 'resnet50', 'None'])),
    (('--modality',), dict(type=str, default='rgbd', choices=['rgbd', 'rgb', 'depth'])),
    (('--encoder_decoder_fusion',), dict(type=str, default='add', choices=['add', 'None'])),
    (('--context_module',), dict(type=str, default='ppm',
                                 choices=['ppm', 'None', 'ppm-1-2-4-8', 'appm', 'appm-1-2-4-8'])),
    (('--channels_decoder',), dict(type=int, default=128)),
    (('--decoder_channels_mode',), dict(default='decreasing', choices=['constant', 'decreasing'])),
    (('--fuse_depth_in_rgb_encoder',), dict(default='SE-add', choices=['SE-add', 'add', 'None'])),
    (('--upsampling',), dict(default='learned-3x3-zeropad',
                             choices=['nearest', 'bilinear', 'learned-3x3', 'learned-3x3-zeropad'])),
    # data
    (('--dataset',), dict(default='nyuv2', choices=['sunrgbd', 'nyuv2', 'cityscapes', 'cityscapes-with-depth', 'scenenetrgbd', 'synthetic'])),
    (('--dataset_dir',), dict(default=None)),
    (('--raw_depth',), dict(action='store_true', default=False)),
    (('--aug_scale_min',), dict(default=1.0, type=float)),
    (('--aug_scale_max',), dict(default=1.4, type=float)),
    (('-j', '--workers'), dict(default=32, type=int)),
    (('--debug',), dict(default=False, action='store_true')),
]


class ArgumentParserRGBDSegmentation(argparse.ArgumentParser):
    def set_common_args(self):
        for flags, kw in _COMMON:
            self.add_argument(*flags, **kw)

    def set_eval_args(self):
        """eval.py:41-46"""
        self.add_argument('--ckpt_path', type=str, required=False, default=None)
        self.add_argument('--hard', action='store_true')
        self.add_argument('--mode', type=int, default=-1)
        self.add_argument('--num-runs', '--num_runs', type=int, default=1, dest='num_runs')
        self.add_argument('--noise', type=float, default=0.0)
        self.add_argument('--ini', action='store_true')
