#!/usr/bin/env python3
"""ImageNet training script for the GA models on the MI355X-native engine -- the timm-style surface of the
reference's GA/train.py (argument names, loss, step, logging line), with the hot path on libgaext HIP kernels.

Differences from the reference, all deliberate: bf16 instead of fp16+GradScaler (`--amp`), a
`--synthetic` loader (the only one shipped: metric runs use synthetic 3x224x224), gradients reduced once per
optimizer step, and `--device cpu` is refused: the product path has no CPU implementation (the CPU baseline is the
oracle timed by bench.py).

  python train.py --synthetic --model ga_convnext_tiny_768 -b 256 --epochs 1 --steps-per-epoch 50 --GA_lam -0.8
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py --synthetic ...
"""
import argparse
import logging
import os
import sys
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

_logger = logging.getLogger('train')

parser = argparse.ArgumentParser(description='GA ImageNet training (MI355X-native)')
parser.add_argument('data_dir', nargs='?', default='', help='dataset root (unused with --synthetic)')
parser.add_argument('--model', default='ga_convnext_tiny_768')
parser.add_argument('--num-classes', type=int, default=None)
parser.add_argument('-b', '--batch-size', type=int, default=128)
parser.add_argument('--epochs', type=int, default=300)
parser.add_argument('--steps-per-epoch', type=int, default=100, help='synthetic epoch length (batches)')
parser.add_argument('--opt', default='sgd')
parser.add_argument('--opt-eps', type=float, default=None)
parser.add_argument('--opt-betas', type=float, nargs='+', default=None)
parser.add_argument('--momentum', type=float, default=0.9)
parser.add_argument('--weight-decay', type=float, default=2e-5)
parser.add_argument('--lr', type=float, default=0.05)
parser.add_argument('--sched', default='cosine')
parser.add_argument('--warmup-lr', type=float, default=1e-4)
parser.add_argument('--min-lr', type=float, default=1e-6)
parser.add_argument('--warmup-epochs', type=int, default=3)
parser.add_argument('--smoothing', type=float, default=0.1)
parser.add_argument('--bce-loss', action='store_true')
parser.add_argument('--bce-target-thresh', type=float, default=None, help='threshold for binarizing softened BCE targets')
# mixup / cutmix: the defaults are the reference's (GA/train.py:215-228: both ON); 0 / 0 turns them off
parser.add_argument('--mixup', type=float, default=0.2, help='mixup alpha, mixup enabled if > 0')
parser.add_argument('--cutmix', type=float, default=1.0, help='cutmix alpha, cutmix enabled if > 0')
parser.add_argument('--cutmix-minmax', type=float, nargs='+', default=None)
parser.add_argument('--mixup-prob', type=float, default=1.0)
parser.add_argument('--mixup-switch-prob', type=float, default=0.5)
parser.add_argument('--mixup-mode', default='batch', help='only "batch" is built')
parser.add_argument('--mixup-off-epoch', type=int, default=0)
parser.add_argument('--drop-path', type=float, default=None)
parser.add_argument('--grad-accumulation', type=int, default=1)
parser.add_argument('--GA_lam', type=float, default=0)
parser.add_argument('--dec-lam', type=float, default=None, help='MAP: weight of the group-decorrelation KL term '
                    '(MAP/train_with_script.py:39, multi_group_loss); alias of --GA_lam for the map_* models')
parser.add_argument('--no-ddp-bb', action='store_true', help='no per-forward broadcast of the BatchNorm buffers from rank 0 '
                    '(GA/train.py:283,514)')
parser.add_argument('--sync-bn', action='store_true', help='BatchNorm statistics over the global batch (GA/train.py:247,449-455): the '
                    'per-channel sums go through the native RCCL communicator; implies --comm native')
parser.add_argument('--comm', default='torch', choices=['torch', 'native', 'native-bf16'], help='transport of the gradient buckets: '
                    'torch.distributed, or libgaext\'s own RCCL communicator (ga_allreduce_bucket) with an fp32 / bf16 wire')
parser.add_argument('--dist-bn', default='reduce', help='"reduce" | "broadcast" | "": distribute the BatchNorm running '
                    'statistics between ranks after every epoch (timm distribute_bn, GA/train.py:665-674)')
parser.add_argument('--model-ema', action='store_true', help='track an EMA of the weights (timm ModelEmaV2)')
parser.add_argument('--model-ema-decay', type=float, default=0.9998)
parser.add_argument('--clip-grad', type=float, default=None, help='clip gradients (GA/train.py --clip-grad)')
parser.add_argument('--clip-mode', default='norm', help='"norm", "value" or "agc"')
parser.add_argument('--amp', action='store_true', help='bf16 math mode (default)')
parser.add_argument('--fp32', action='store_true', help='fp32 parity math mode')
parser.add_argument('--channels-last', action='store_true', help='accepted for CLI compatibility (activations are always NHWC)')
parser.add_argument('--synthetic', action='store_true')
parser.add_argument('--device', default='cuda')
parser.add_argument('--seed', type=int, default=42)
parser.add_argument('--log-interval', type=int, default=50)
parser.add_argument('--output', default='')
parser.add_argument('--local_rank', default=0, type=int)


class SyntheticLoader:
    """fixed-shape random batches generated on the device (seed = args.seed + rank, like timm random_seed)"""

    def __init__(self, batch, steps, num_classes, seed, device, img=224):
        self.batch, self.steps, self.nc, self.img = batch, steps, num_classes, img
        self.g = torch.Generator(device=device).manual_seed(seed)
        self.device = device

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            x = torch.randn(self.batch, 3, self.img, self.img, device=self.device, generator=self.g)
            y = torch.randint(0, self.nc, (self.batch,), device=self.device, generator=self.g)
            yield x, y


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def train_one_epoch(epoch, step_fn, loader, args, world, rank, model_ema=None):
    batch_time_m, losses_m = AverageMeter(), AverageMeter()
    end = time.time()
    last_idx = len(loader) - 1
    loss = None
    for batch_idx, (x, y) in enumerate(loader):
        loss = step_fn(x, y)
        if model_ema is not None and step_fn.micro % step_fn.accum == 0:   # after every optimizer step (GA/train.py:774)
            model_ema.update()
        if batch_idx % args.log_interval == 0 or batch_idx == last_idx:
            torch.cuda.synchronize()
            lv = loss.detach().clone()
            if world > 1:
                dist.all_reduce(lv)
                lv /= world
            losses_m.update(float(lv), x.size(0))
            batch_time_m.update((time.time() - end) / (args.log_interval if batch_idx else 1))
            if rank == 0:
                _logger.info('Train: {} [{:>4d}/{} ({:>3.0f}%)]  Loss: {:#.4g} ({:#.3g})  Time: {:.3f}s, {:>7.2f}/s  LR: {:.3e}'.format(
                    epoch, batch_idx, len(loader), 100. * batch_idx / max(last_idx, 1), losses_m.val, losses_m.avg,
                    batch_time_m.val, x.size(0) * world / batch_time_m.val, step_fn.opt.param_groups[0]['lr']))
            end = time.time()
    return OrderedDict([('loss', losses_m.avg)])


def validate(model, loader, args, world):
    import imagenet_models_amd as A
    model.eval()
    top1_m, top5_m = AverageMeter(), AverageMeter()
    with torch.no_grad():
        for x, y in loader:
            outs = model(x)
            _, idx = A.heads_topk(outs, 5)                       # output = sum_k out_k.float() (train.py:848-851)
            acc1, acc5 = A.accuracy_from_topk(idx, y, (1, 5))
            if world > 1:
                t = torch.stack([acc1, acc5])
                dist.all_reduce(t)
                acc1, acc5 = t / world
            top1_m.update(float(acc1), x.size(0))
            top5_m.update(float(acc5), x.size(0))
    model.train()
    return OrderedDict([('top1', top1_m.avg), ('top5', top5_m.avg)])


def main():
    logging.basicConfig(level=logging.INFO, format='%(message)s')
    args = parser.parse_args()
    if args.device != 'cuda':
        raise SystemExit('train.py: --device cpu is not available: the product path runs on the libgaext HIP kernels only '
                         '(the CPU baseline is the oracle timed by `bench.py`, kind "port")')
    if not args.synthetic:
        raise SystemExit('train.py: only --synthetic data is shipped (no dataset / network in this environment)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(args.local_rank)))
    backend = os.environ.get('GA_DIST_BACKEND', 'nccl')          # gloo: rehearsal with ranks sharing a device
    ndev = max(1, torch.cuda.device_count())
    if local >= ndev:
        if world > 1 and backend == 'nccl':
            raise SystemExit(f'train.py: LOCAL_RANK {local} but only {ndev} GPU(s) are visible: RCCL needs one device per rank '
                             '(GA_DIST_BACKEND=gloo rehearses several ranks on one device)')
        local %= ndev
    torch.cuda.set_device(local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', init_method='env://', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(backend, init_method='env://')
    import numpy as np
    import imagenet_models_amd as A
    torch.manual_seed(args.seed + rank)
    np.random.seed(args.seed + rank)          # timm random_seed(): mixup draws lam / boxes from numpy's global generator
    model = A.create_model(args.model, pretrained=False, num_classes=args.num_classes, drop_path_rate=args.drop_path,
                           math_mode='fp32' if args.fp32 else 'bf16').cuda()
    if world > 1:
        dist.broadcast(model.flat_state()['params'], 0)
        dist.broadcast(model.flat_state()['buffers'], 0)
    comm = None
    if world > 1 and (args.sync_bn or args.comm != 'torch'):
        if backend != 'nccl':
            raise SystemExit('train.py: --comm native / --sync-bn need one GPU per rank (the RCCL communicator of libgaext)')
        comm = A.NativeComm(wire='bf16' if args.comm == 'native-bf16' else 'fp32')
        if args.sync_bn:
            model.convert_sync_batchnorm(comm)
    if rank == 0:
        _logger.info('Model %s created, param count: %d', args.model, sum(p.numel() for p in model.parameters()))
    opt = A.create_optimizer_v2(model, opt=args.opt, lr=args.lr, weight_decay=args.weight_decay, momentum=args.momentum,
                                eps=args.opt_eps, betas=tuple(args.opt_betas) if args.opt_betas else None)
    sched = A.CosineLRScheduler(opt, t_initial=args.epochs, lr_min=args.min_lr, warmup_t=args.warmup_epochs,
                                warmup_lr_init=args.warmup_lr) if args.sched == 'cosine' else None
    lam = args.dec_lam if args.dec_lam is not None else args.GA_lam
    # mixup / cutmix (GA/train.py:544-557): smoothing then lives in the dense target and the loss is SoftTargetCrossEntropy /
    # BinaryCrossEntropy on it (:616-621)
    mixup_fn = None
    if args.mixup > 0 or args.cutmix > 0. or args.cutmix_minmax is not None:
        mixup_fn = A.Mixup(mixup_alpha=args.mixup, cutmix_alpha=args.cutmix, cutmix_minmax=args.cutmix_minmax, prob=args.mixup_prob,
                           switch_prob=args.mixup_switch_prob, mode=args.mixup_mode, label_smoothing=args.smoothing,
                           num_classes=model.num_classes)
    step_fn = A.TrainStep(model, opt, args.batch_size, lam=lam, loss='bce' if args.bce_loss else 'ce',
                          smoothing=args.smoothing, grad_accumulation=args.grad_accumulation,
                          clip_grad=args.clip_grad, clip_mode=args.clip_mode, broadcast_buffers=not args.no_ddp_bb,
                          mixup_fn=mixup_fn, bce_target_thresh=args.bce_target_thresh, comm=comm)
    model_ema = A.ModelEma(model, args.model_ema_decay) if args.model_ema else None
    img = getattr(model, 'cfg', {}).get('img_size', 224)
    loader = SyntheticLoader(args.batch_size, args.steps_per_epoch, model.num_classes, args.seed + rank, 'cuda', img)
    eval_loader = SyntheticLoader(args.batch_size, max(1, args.steps_per_epoch // 10), model.num_classes, 7 + rank, 'cuda', img)
    model.train()
    for epoch in range(args.epochs):
        if sched is not None:
            sched.step(epoch)
        if mixup_fn is not None and args.mixup_off_epoch and epoch >= args.mixup_off_epoch:
            mixup_fn.mixup_enabled = False      # GA/train.py:705-709
        train_metrics = train_one_epoch(epoch, step_fn, loader, args, world, rank, model_ema)
        if world > 1 and args.dist_bn in ('broadcast', 'reduce'):
            A.distribute_bn(model, world, args.dist_bn == 'reduce')
        eval_metrics = validate(model, eval_loader, args, world)
        if rank == 0:
            _logger.info('*** epoch %d: train loss %.4f  top1 %.3f  top5 %.3f', epoch, train_metrics['loss'],
                         eval_metrics['top1'], eval_metrics['top5'])
            if args.output:
                os.makedirs(args.output, exist_ok=True)
                A.save_checkpoint(model, opt, epoch, os.path.join(args.output, f'checkpoint-{epoch}.pth.tar'),
                                  metric=eval_metrics['top1'], arch=args.model, model_ema=model_ema)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
