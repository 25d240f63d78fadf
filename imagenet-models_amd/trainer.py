"""One optimizer step of the GA training loop (GA/train.py:732-769) on the HIP engine, without autograd:
forward plan -> fused GA loss (writes dlogits) -> backward plan in segments, each finished segment's slices of the
flat gradient buffer all-reduced over RCCL (one process per GPU, torch.distributed backend "nccl") in <= 32 MB
buckets while the remaining backward runs -> fused flat optimizer.

Replaces NativeDDP (GA/train.py:514): its 25 MB reverse-registration-order bucket reducer, its per-forward broadcast of
the BatchNorm buffers from rank 0 (`broadcast_buffers`, off with --no-ddp-bb :283) and, with `distribute_bn`, timm's
epoch-end running-statistics reduction (:665-674).  The flat gradient layout follows the parameter registration order,
so the parameters of one trunk stage / of the heads are a few contiguous slices; `FlatModel.grad_groups()` says which
name prefixes are final at which mark of the backward plan."""
import torch
import torch.distributed as dist

_KINDS = {'ce': 0, 'bce': 1}

BUCKET_ELEMS = 8 << 20     # 32 MB of fp32 per all-reduce: enough to run at link rate over xGMI, small enough to pipeline


def make_buckets(st, groups, bucket_elems=BUCKET_ELEMS):
    """[(mark, start, end)] -- an exact partition of the flat gradient buffer [0, total) into contiguous slices of at most
    `bucket_elems` elements, listed in backward-completion order: the slices of groups[0] (final at its mark) first, ...,
    everything no group claims under mark 'end'."""
    sl = st['slices']
    owner = {}
    order = [m for m, _ in groups] + ['end']
    for n in sl:
        owner[n] = 'end'
        for mark, prefixes in groups:
            if n.startswith(tuple(prefixes)):
                owner[n] = mark
                break
    # contiguous runs of parameters with the same owner, in flat-buffer order
    runs = []
    for n, (off, k) in sorted(sl.items(), key=lambda kv: kv[1][0]):
        if runs and runs[-1][0] == owner[n] and runs[-1][2] == off:
            runs[-1][2] = off + k
        else:
            runs.append([owner[n], off, off + k])
    buckets = []
    for mark in order:
        for own, a, b in sorted((r for r in runs if r[0] == mark), key=lambda r: -r[1]):   # highest offsets first
            nchunk = max(1, -(-(b - a) // bucket_elems))
            step = -(-(b - a) // nchunk)
            cuts = list(range(a, b, step)) + [b]
            for lo, hi in reversed(list(zip(cuts, cuts[1:]))):
                buckets.append((mark, lo, hi))
    return buckets


def distribute_bn(model, world, reduce=False, group=None):
    """timm.utils.distribute_bn (GA/train.py:665-674): average (reduce=True) or broadcast from rank 0 the BatchNorm
    running statistics of every rank -- ONE collective over the model's flat buffer of float buffers"""
    if world <= 1:
        return
    fb = model.flat_state()['buffers']
    if fb.numel() == 0:
        return
    if reduce:
        dist.all_reduce(fb, group=group)
        fb /= float(world)
    else:
        dist.broadcast(fb, 0, group=group)


class TrainStep:
    def __init__(self, model, optimizer, batch, lam=0.0, loss='ce', smoothing=0.0, grad_accumulation=1,
                 process_group=None, clip_grad=None, clip_mode='norm', broadcast_buffers=True, bucket_elems=BUCKET_ELEMS,
                 mixup_fn=None, bce_target_thresh=None, comm=None, force_buckets=False, nan_guard=False,
                 overlap_optimizer=False):
        """comm: an imagenet_models_amd.NativeComm -- the gradient buckets (and the BatchNorm-buffer broadcast) go through the
        library's own RCCL entry points (ga_allreduce_bucket on a side stream) instead of torch.distributed.
        force_buckets: take the segmented-backward + bucketed-reduction path even with one rank (tests: the real collective
        path on a one-GPU box).
        nan_guard: the device-side counterpart of MAP/train.py:887-891 (all_gather of the loss + isnan + exit): the loss value
        rides in the LAST gradient bucket's reduction as one extra element; `last_loss_sum` (a device tensor, the sum of the
        ranks' scaled losses) can be inspected by the caller at its logging interval -- no host synchronisation per step."""
        self.model, self.opt = model, optimizer
        self.eng = model.engine(batch, True)
        self.lam, self.kind, self.smoothing = lam, _KINDS[loss], smoothing
        # mixup / cutmix (imagenet_models_amd.Mixup, GA/train.py:727-728): applied to every batch, the loss then runs on the
        # dense target it returns (SoftTargetCrossEntropy / BinaryCrossEntropy, train.py:616-621)
        self.mixup_fn = mixup_fn
        self.bce_threshold = -1.0 if bce_target_thresh is None else float(bce_target_thresh)
        self.accum = grad_accumulation
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.micro = 0
        st = model.flat_state()
        self.gen = st['gen']
        self.flat_g = st['grads']
        self.flat_buffers = st['buffers']
        self.broadcast_buffers = broadcast_buffers and self.flat_buffers.numel() > 0     # NativeDDP default (GA/train.py:514)
        self.comm = comm
        if comm is not None and comm.world != self.world:
            raise ValueError(f'comm has {comm.world} ranks, the process group {self.world}')
        self.bucketed = self.world > 1 or force_buckets
        self.nan_guard = nan_guard
        # overlap_optimizer: the optimizer update of a bucket (and the zeroing of its gradients) follows its reduction on a side
        # stream while backward goes on -- possible whenever nothing needs ALL gradients first (norm clipping, LAMB's norms, the
        # non-finite guard).  Off by default: on one GPU the 0.26 ms AdamW pass hidden this way costs more in segmented issue
        # and HBM contention than it saves (ga_convnext_tiny_768, B = 256: 24.9 ms with, 24.7 without, same box)
        self.overlap_opt = (overlap_optimizer and getattr(optimizer, 'supports_ranges', False) and clip_grad is None
                            and not nan_guard and bool(model.grad_groups()))
        self.buckets = make_buckets(st, model.grad_groups(), bucket_elems) if (self.bucketed or self.overlap_opt) else []
        self.opt_stream = torch.cuda.Stream() if (self.overlap_opt and comm is None) else None
        self._opt_done = torch.cuda.Event() if self.overlap_opt else None
        self.last_loss_sum = torch.zeros(1, device=self.flat_g.device) if nan_guard else None
        # gradient clipping (timm dispatch_clip_grad through NativeScaler, GA/train.py:312-333): global L2 norm or
        # value clamp over the flat gradient buffer, after the all-reduce, before the optimizer
        if clip_mode not in ('norm', 'value', 'agc'):
            raise ValueError(f"clip_mode {clip_mode!r}: 'norm', 'value' or 'agc'")
        self.clip_plan = None
        if clip_grad is not None:
            from . import ops
            self.gnorm_sq = torch.zeros(1, device=self.flat_g.device)
            p = ops.Plan(name='clip')
            n = self.flat_g.numel()
            if clip_mode == 'agc':
                # timm adaptive_clip_grad over model_parameters(model, exclude_head=True) = parameters()[:-2] (train.py:755):
                # one unit per row of a >= 2-d parameter, per tensor otherwise
                units = []
                named = list(model.named_parameters())[:-2]
                offs = st['slices']
                for name, prm in named:
                    o = offs[name][0]
                    if prm.dim() > 1:
                        row = prm[0].numel()
                        units.extend((o + r * row, row) for r in range(prm.shape[0]))
                    else:
                        units.append((o, prm.numel()))
                self.agc_units = torch.tensor(units, dtype=torch.int64, device=self.flat_g.device)
                p.agc_clip(st['params'], self.flat_g, self.agc_units, len(units), clip_grad)
            elif clip_mode == 'norm':
                p.zero(self.gnorm_sq)
                p.sumsq_f32(self.flat_g, n, self.gnorm_sq)
                p.clip_grad_f32(self.flat_g, n, self.gnorm_sq, clip_grad, 0)
            else:
                p.clip_grad_f32(self.flat_g, n, self.gnorm_sq, clip_grad, 1)
            self.clip_plan = p

    def _backward_with_updates(self, bwd):
        """backward in bucket segments; behind each segment, on a side stream: [all-reduce of the bucket,] optimizer update of
        the bucket, zeroing of its gradients.  The main stream rejoins before the step returns."""
        opt = self.opt
        opt.step_begin()                                  # hyper-parameters reach the device on the main stream, ahead of every event
        main = torch.cuda.current_stream()
        side = self.comm.stream if self.comm is not None else self.opt_stream
        reduce = self.bucketed and (self.comm is not None or self.world > 1 or dist.is_initialized())
        pos = 0
        for mark, a, b in self.buckets:
            stop = len(bwd.calls) if mark == 'end' else bwd.marks[mark]
            if stop > pos:
                bwd.run_range(pos, stop)
                pos = stop
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            if reduce and self.comm is not None:
                self.comm.allreduce(self.flat_g[a:b])
            elif reduce:
                work = dist.all_reduce(self.flat_g[a:b], group=self.pg, async_op=True)
                with torch.cuda.stream(side):
                    work.wait()
            opt.step_range(a, b, side.cuda_stream)
        if pos < len(bwd.calls):
            bwd.run_range(pos, len(bwd.calls))
        self._opt_done.record(side)
        main.wait_event(self._opt_done)
        opt.step_end()

    def __call__(self, x, target):
        eng = self.eng
        self.model.check_flat_generation(self.gen, 'TrainStep')
        last_micro = (self.micro + 1) % self.accum == 0
        if self.world > 1 and self.broadcast_buffers:                # rank 0's BatchNorm statistics before every forward
            if self.comm is not None:
                main = torch.cuda.current_stream()
                self.comm.after(main)
                self.comm.broadcast(self.flat_buffers, 0)
                self.comm.join(main)
            else:
                dist.broadcast(self.flat_buffers, 0, group=self.pg)
        # the reference divides the loss by grad_accumulation (train.py:750); DDP averages over ranks
        scale = 1.0 / (self.accum * self.world)
        if self.mixup_fn is not None:
            # a uint8 batch (PrefetchLoader layout) is normalised on the device FIRST: Mixup blends normalised pixels, and a
            # float tensor coming back from it would make the engine skip the normalisation
            x, target = self.mixup_fn(eng._normalize_u8(x), target)
        loss = eng.forward_loss(x, target, self.lam, self.kind, self.smoothing, scale, self.bce_threshold)
        bwd = eng.bwd
        if self.overlap_opt and last_micro:
            self._backward_with_updates(bwd)
            self.micro += 1
            return loss
        if self.bucketed and last_micro:
            works, pos = [], 0
            main = torch.cuda.current_stream()
            for mark, a, b in self.buckets:
                stop = len(bwd.calls) if mark == 'end' else bwd.marks[mark]
                if stop > pos:
                    bwd.run_range(pos, stop)
                    pos = stop
                if self.comm is not None:            # RCCL through the C ABI, on the comm stream, behind this segment
                    self.comm.after(main)
                    self.comm.allreduce(self.flat_g[a:b])
                elif self.world > 1 or dist.is_initialized():
                    works.append(dist.all_reduce(self.flat_g[a:b], group=self.pg, async_op=True))
            if self.nan_guard:                        # one float beside the last bucket: sum over ranks of the scaled loss
                self.last_loss_sum.copy_(loss.reshape(1), non_blocking=True)
                if self.comm is not None:
                    self.comm.after(main)
                    self.comm.allreduce(self.last_loss_sum)
                elif self.world > 1 or dist.is_initialized():
                    works.append(dist.all_reduce(self.last_loss_sum, group=self.pg, async_op=True))
            if self.comm is not None:
                self.comm.join(main)
            for w in works:
                w.wait()
        else:
            bwd.run()
        self.micro += 1
        if last_micro:
            if self.clip_plan is not None:
                self.clip_plan.run()
            self.opt.step()
            self.opt.zero_grad()
        return loss
