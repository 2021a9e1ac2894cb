"""python -m dynmm_amd.eval --dynamic --global-gate [--baseline] --hard --ckpt_path CKPT ...

Counterpart of FusionDynMM/eval.py:35-151: load a (reference-format) checkpoint strictly, set the gate
flags, optional Gaussian-noise robustness runs (modes 0/1/2), mIoU*100 per run."""
import random
import sys

import numpy as np
import torch

from . import engine
from .data import SyntheticRGBD
from .src.args import ArgumentParserRGBDSegmentation
from .src.build_model import build_model


def set_seed(seed):
    """src/utils.py set_seed as eval.py uses it before every noise-robustness run."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def run(args, model, valid_loader, device=None, use_graph=True):
    """eval.py:63-151 for a model the caller built and ANY loader the caller brings — an iterable of dicts with `image` [N,3,H,W],
    `depth` [N,1,H,W] (normalised as src/preprocessing.py does) and `label_orig` [N,H0,W0] (0 = void), on the host or the device:
    the entry a user with NYUv2 on disk calls (`python -m dynmm_amd.eval` itself ships only the synthetic loader; the data set's
    preparation is host-side code outside the hot path).  Sets the gate flags from `args` (hard / ini / baseline), runs
    `args.num_runs` passes with the reference's per-run seeds and Gaussian-noise modes 0 / 1 / 2, prints and returns mIoU*100 per
    run.  `use_graph`: the forward is replayed as hipGraphs (engine.InferStep; eager where that does not apply)."""
    device = next(model.parameters()).device if device is None else device
    model.eval()
    if hasattr(model, 'start_weight'):
        model.start_weight()
    model.hard_gate, model.ini_stage, model.baseline = args.hard, args.ini, args.baseline
    step = engine.InferStep(model, policy='auto') if use_graph else None      # (replay or eager launches: whichever this host runs faster)
    results = []
    for r in range(args.num_runs):
        set_seed(r)                                              # eval.py: per-run seed -> reproducible noise runs

        def batches():
            for s in valid_loader:
                image, depth = s['image'].to(device), s['depth'].to(device)
                u = random.random()                              # eval.py:91-102
                if args.mode in (0, 2) and u < 0.33:
                    image = image + args.noise * image.abs().mean() * torch.randn_like(image)
                elif (args.mode == 1 and u < 0.33) or (args.mode == 2 and u < 0.66):
                    depth = depth + args.noise * depth.abs().mean() * torch.randn_like(depth)
                yield image, depth, s['label_orig'].to(device)
        miou, _ = engine.evaluate(model, batches(), hard=args.hard, infer_step=step)
        print(f'Run {r}, mIoU: {miou:0.2f}')
        results.append(miou)
    if hasattr(model, 'end_weight'):
        model.end_weight(print_flop=args.hard)
    print(results)
    return results


def main(argv=None):
    p = ArgumentParserRGBDSegmentation(description='Efficient RGBD Indoor Semantic Segmentation (Evaluation, MI355X)')
    p.set_common_args()
    p.set_eval_args()
    p.add_argument('--synthetic_samples', type=int, default=8)
    p.add_argument('--no_hip_graph', action='store_true', help='eager launches instead of hipGraph replays of the forward')
    args = p.parse_args(argv)
    args.pretrained_on_imagenet = False
    model, device = build_model(args, n_classes=40)
    if args.ckpt_path:
        ckpt = torch.load(args.ckpt_path, map_location=device)
        model.load_state_dict(ckpt['state_dict'])               # strict, eval.py:60-61
        print(f'Loaded checkpoint from {args.ckpt_path}')
    data = SyntheticRGBD(args.synthetic_samples, args.batch_size_valid or args.batch_size, args.height, args.width,
                         seed=77, device=device)
    return run(args, model, data, device, use_graph=not args.no_hip_graph)


if __name__ == '__main__':
    main(sys.argv[1:])
