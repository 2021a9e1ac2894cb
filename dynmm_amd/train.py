"""python -m dynmm_amd.train --dynamic --global-gate --encoder resnet34 --encoder_block NonBottleneck1D \
       --decoder_channels_mode constant --dataset synthetic [...]

Counterpart of FusionDynMM/train.py on the HIP path: same flags (dynmm_amd/src/args.py), same epoch
protocol (ini_stage / hard_gate / temperature per epoch, OneCycle stepped per epoch, lr scaled by
batch/8, total loss rule, NaN guard, periodic validation, checkpoints with the reference's keys).
One process per GPU under torch.distributed.run = data parallel (new; the reference is single-GPU)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import dp, engine, schedules
from .data import SyntheticRGBD
from .src.args import ArgumentParserRGBDSegmentation
from .src.build_model import build_model
from .src.pretrained import load_ckpt


def parse_args(argv=None):
    p = ArgumentParserRGBDSegmentation(description='Efficient RGBD Indoor Semantic Segmentation (Training, MI355X)')
    p.set_common_args()
    p.add_argument('--synthetic_samples', type=int, default=64)
    p.add_argument('--hip_graph', action='store_true', help='replay each step as one hipGraph')
    args = p.parse_args(argv)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.lr = schedules.scaled_lr(args.lr, args.batch_size * world)      # train.py:46-49 on the GLOBAL batch
    return args


def save_ckpt(ckpt_dir, model, opt, epoch, best_miou=None, best_miou_epoch=None):
    """Same keys as src/utils.py:118-127 (+ the best-mIoU bookkeeping load_ckpt looks for, :160-170)."""
    path = os.path.join(ckpt_dir, f'ckpt_epoch_{epoch}.pth')
    state = {'epoch': epoch, 'state_dict': model.state_dict(), 'optimizer': opt.state_dict()}
    if best_miou is not None:
        state.update(best_miou=best_miou, best_miou_epoch=best_miou_epoch)
    torch.save(state, path)
    return path


def train_main(argv=None):
    args = parse_args(argv)
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        dist.init_process_group(os.environ.get('DYNMM_DIST_BACKEND', 'nccl'))
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
    if args.dataset != 'synthetic' and args.dataset_dir is None:
        raise NotImplementedError('only --dataset synthetic ships with the HIP path; pass your own loader to '
                                  'dynmm_amd.train.run(...) for real data (NYUv2 preparation is host-side code '
                                  'outside the hot path)')
    ckpt_dir = os.path.join(args.results_dir, args.dataset, time.strftime('checkpoints_%d_%m_%Y-%H_%M_%S'))
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
        with open(os.path.join(ckpt_dir, 'args.json'), 'w') as f:
            json.dump(vars(args), f, sort_keys=True, indent=4)
    model, device = build_model(args, n_classes=40)
    lo, hi = dp.shard_batch(args.synthetic_samples, rank, world)
    train = SyntheticRGBD(hi - lo, args.batch_size, args.height, args.width, seed=1000 * rank, device=device, nyu_like=False)
    valid = SyntheticRGBD(max(args.batch_size, 8), args.batch_size_valid or args.batch_size, args.height, args.width,
                          seed=777, device=device)
    return run(args, model, train, valid, ckpt_dir, rank, world)


CAMERA = 'kv1'        # NYUv2's only camera (src/datasets/nyuv2/nyuv2.py: CAMERAS); train.py:217 reports mIoU_test_kv1


def class_weights(args, train_loader, world=1):
    """--class_weighting over the WHOLE training set (train.py:100-103): each rank holds a shard of it, so the two label
    histograms are summed over ranks before the weights are formed — every replica then weighs its CE identically."""
    if args.class_weighting == 'None':
        return np.ones(40)
    if world > 1 and hasattr(train_loader, 'class_counts'):
        per, with_ = train_loader.class_counts()
        both = torch.from_numpy(np.stack([per, with_]))                 # float64 pixel counts: exact in a sum over ranks
        if dist.get_backend() != 'gloo':
            both = both.cuda()
        dist.all_reduce(both)
        per, with_ = both.cpu().numpy()
        return train_loader.weights_from_counts(per, with_, args.class_weighting, c=args.c_for_logarithmic_weighting)
    return train_loader.compute_class_weights(args.class_weighting, c=args.c_for_logarithmic_weighting)


def run(args, model, train_loader, valid_loader, ckpt_dir, rank=0, world=1):
    dp.broadcast_parameters(model)
    cw = class_weights(args, train_loader, world)
    if args.freeze and args.dynamic:                      # train.py:139-141
        print('Freeze everything but the soft gates')
        model.freeze()
    step = engine.TrainStep(model, cw, lr=args.lr, momentum=args.momentum, weight_decay=args.weight_decay,
                            loss_ratio=args.loss_ratio, flop_budget=args.flop_budget, use_graph=args.hip_graph,
                            optimizer=args.optimizer)
    print('Using {} as optimizer'.format(args.optimizer))
    temp = schedules.ExpDecayTemp(args.temp, args.end_temp, args.epoch_hard)
    model.baseline = args.baseline
    best_miou, best_epoch, best_state, logs = 0.0, 0, None, []
    start_epoch = 0
    if args.last_ckpt:                                    # train.py:131-135: model + optimizer state + epoch counter
        last, best_miou, best_epoch = load_ckpt(model, step.opt, args.last_ckpt)
        start_epoch = last + 1
        step.flatp.refresh()                              # load_state_dict copied INTO the flat views; nothing to re-home
    for epoch in range(start_epoch, args.epochs):
        assert args.epoch_ini <= args.epoch_hard
        model.ini_stage = epoch < args.epoch_ini
        model.hard_gate = epoch >= args.epoch_hard
        model.temp = temp.get_t(epoch)
        lr = schedules.one_cycle_lr(epoch, args.epochs, args.lr)
        step.opt.set_lr(lr)
        # OneCycleLR(cycle_momentum=True) also rewrites the momentum (SGD) / beta1 (Adam) every epoch
        step.opt.set_momentum(schedules.one_cycle_momentum(epoch, args.epochs))
        model.train()
        t0, tot, flop, nb = time.time(), [], [], 0
        for i, sample in enumerate(train_loader):
            targets = [sample['label']] + [sample['label_down'][r] for r in (8, 16, 32)]
            out = step(sample['image'], sample['depth'], targets)
            tot.append(out['total'])
            flop.append(out['loss_flop'])
            nb += 1
            if args.debug:
                break
        # NaN guard: every step's loss is inspected on the device by the optimizer kernel (a non-finite loss
        # skips the update and latches the step index); the host reads the latch once per epoch instead of
        # synchronising on the loss every step (train.py:328-335).
        step.opt.check_finite()
        total = torch.cat([t.reshape(1) for t in tot]).mean().item()
        if np.isnan(total):
            raise ValueError('Loss is None')
        row = {'epoch': epoch, 'lr_0': lr, 'loss_train_total': total,
               'loss_flop': torch.stack([f.reshape(()) for f in flop]).mean().item(),
               'time_training': time.time() - t0, 'temp': model.temp}
        if epoch == start_epoch or epoch % args.eval_every == 0:        # train.py:207
            # validate (train.py:368-551): per-camera mIoU + weighted / unweighted validation loss; under data parallel
            # the validation batches are sharded over the ranks and the confusion matrix is all-reduced once
            miou, row = engine.validate(model, {CAMERA: valid_loader}, cw, logs=row, split='test',
                                        soft_eval=args.soft_eval, dynamic=args.dynamic)
            row.pop('confusion_matrices', None)
            row['mIoU_test'] = miou[CAMERA]
            if miou[CAMERA] > best_miou:
                best_miou, best_epoch = miou[CAMERA], epoch
                if rank == 0:
                    # train.py:235 deep-copies the model here; a CPU snapshot of its state_dict is what gets saved
                    # (rank 0 only: it is the rank that writes checkpoints)
                    best_state = {k: v.detach().to('cpu', copy=True) for k, v in model.state_dict().items()}
        if rank == 0:
            print(f"Epoch {epoch} | Train loss {row['loss_train_total']:.4f} | Flop loss {row['loss_flop']:.4f} "
                  f"Temperature {model.temp} | lr {lr}" + (f" | Test loss {row['loss_test']:.4f} | Test mIoU {row['mIoU_test']:.2f}" if 'mIoU_test' in row else ''))
            if epoch >= 10 and epoch % args.save_every == args.save_every - 1:
                save_ckpt(ckpt_dir, model, step.opt, epoch, best_miou, best_epoch)
        logs.append(row)
    if rank == 0:
        # train.py:250: the BEST model's weights under the best epoch's name
        if best_state is not None:
            path = os.path.join(ckpt_dir, f'ckpt_epoch_{best_epoch}.pth')
            torch.save({'epoch': best_epoch, 'state_dict': best_state, 'optimizer': step.opt.state_dict()}, path)
        else:
            # no new best in this run (e.g. a resume whose earlier best still stands): the best checkpoint on disk keeps
            # its name and content; the final weights go under the last epoch's name instead of overwriting it
            save_ckpt(ckpt_dir, model, step.opt, max(start_epoch, args.epochs - 1), best_miou, best_epoch)
        with open(os.path.join(ckpt_dir, 'finished.txt'), 'w') as f:
            f.write(f'best miou: {best_miou}\nbest miou epoch: {best_epoch}\n')
    if world > 1:
        dist.destroy_process_group()
    return logs


if __name__ == '__main__':
    train_main(sys.argv[1:])
