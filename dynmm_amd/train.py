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


def run(args, model, train_loader, valid_loader, ckpt_dir, rank=0, world=1):
    dp.broadcast_parameters(model)
    cw = train_loader.compute_class_weights(args.class_weighting, c=args.c_for_logarithmic_weighting) if args.class_weighting != 'None' else np.ones(40)
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
        if epoch == 0 or epoch % args.eval_every == 0:
            batches = ((s['image'], s['depth'], s['label_orig']) for s in valid_loader)
            miou, _ = engine.evaluate(model, batches, hard=not args.soft_eval)
            row['mIoU_test'] = miou
            if miou > best_miou:
                best_miou, best_epoch = miou, epoch
                # train.py:235 deep-copies the model here; a CPU snapshot of its state_dict is what gets saved
                best_state = {k: v.detach().to('cpu', copy=True) for k, v in model.state_dict().items()}
        if rank == 0:
            print(f"Epoch {epoch} | Train loss {row['loss_train_total']:.4f} | Flop loss {row['loss_flop']:.4f} "
                  f"Temperature {model.temp} | lr {lr}" + (f" | mIoU {row['mIoU_test']:.2f}" if 'mIoU_test' in row else ''))
            if epoch >= 10 and epoch % args.save_every == args.save_every - 1:
                save_ckpt(ckpt_dir, model, step.opt, epoch, best_miou, best_epoch)
        logs.append(row)
    if rank == 0:
        # train.py:250: the BEST model's weights under the best epoch's name
        path = os.path.join(ckpt_dir, f'ckpt_epoch_{best_epoch}.pth')
        state = best_state if best_state is not None else {k: v.detach().cpu() for k, v in model.state_dict().items()}
        torch.save({'epoch': best_epoch, 'state_dict': state, 'optimizer': step.opt.state_dict()}, path)
        with open(os.path.join(ckpt_dir, 'finished.txt'), 'w') as f:
            f.write(f'best miou: {best_miou}\nbest miou epoch: {best_epoch}\n')
    if world > 1:
        dist.destroy_process_group()
    return logs


if __name__ == '__main__':
    train_main(sys.argv[1:])
