"""One optimizer step of the GA training loop (GA/train.py:732-769) on the HIP engine, without autograd:
forward plan -> fused GA loss (writes dlogits) -> backward plan in segments, each finished segment's slice of the
flat gradient buffer all-reduced over RCCL (one process per GPU) while the remaining backward runs -> fused flat
optimizer.  Replaces NativeDDP's 25 MB bucket reducer (GA/train.py:514): the flat gradient layout follows the
parameter registration order, so "stages.4 + heads", "stages.3", ... are contiguous slices."""
import torch
import torch.distributed as dist

_KINDS = {'ce': 0, 'bce': 1}


class TrainStep:
    def __init__(self, model, optimizer, batch, lam=0.0, loss='ce', smoothing=0.0, grad_accumulation=1,
                 process_group=None, clip_grad=None, clip_mode='norm'):
        self.model, self.opt = model, optimizer
        self.eng = model.engine(batch, True)
        self.lam, self.kind, self.smoothing = lam, _KINDS[loss], smoothing
        self.accum = grad_accumulation
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.micro = 0
        st = model.flat_state()
        self.flat_g = st['grads']
        self.buckets = self._make_buckets(st) if self.world > 1 else []
        # gradient clipping (timm dispatch_clip_grad through NativeScaler, GA/train.py:312-333): global L2 norm or
        # value clamp over the flat gradient buffer, after the all-reduce, before the optimizer
        if clip_mode not in ('norm', 'value'):
            raise NotImplementedError(f"clip_mode {clip_mode!r}: only 'norm' and 'value' are built ('agc' is not)")
        self.clip_plan = None
        if clip_grad is not None:
            from . import ops
            self.gnorm_sq = torch.zeros(1, device=self.flat_g.device)
            p = ops.Plan(name='clip')
            n = self.flat_g.numel()
            if clip_mode == 'norm':
                p.zero(self.gnorm_sq)
                p.sumsq_f32(self.flat_g, n, self.gnorm_sq)
                p.clip_grad_f32(self.flat_g, n, self.gnorm_sq, clip_grad, 0)
            else:
                p.clip_grad_f32(self.flat_g, n, self.gnorm_sq, clip_grad, 1)
            self.clip_plan = p

    def _make_buckets(self, st):
        """[(plan mark, start, end)] -- contiguous slices of the flat gradient buffer in backward-completion order"""
        sl = st['slices']
        names_decay = [n for n in sl if sl[n][0] < st['n_decay']]

        def first_off(prefixes):
            offs = [sl[n][0] for n in names_decay if n.startswith(prefixes)]
            return min(offs) if offs else None

        cuts = []  # (mark, start offset of the slice that becomes final at this mark)
        head_start = first_off(('stages.4.',))
        cuts.append(('heads', head_start))
        for i in (3, 2, 1):
            cuts.append((f'stage{i}', first_off((f'stages.{i}.',))))
        buckets, end = [], st['n_decay']
        for mark, start in cuts:
            buckets.append((mark, start, end))
            end = start
        # stem + stages.0 (decay) and every no-decay parameter (biases, norms, gammas) go last
        buckets.append(('end', 0, end))
        buckets.append(('end', st['n_decay'], st['total']))
        return buckets

    def __call__(self, x, target):
        eng = self.eng
        last_micro = (self.micro + 1) % self.accum == 0
        # the reference divides the loss by grad_accumulation (train.py:750); DDP averages over ranks
        scale = 1.0 / (self.accum * self.world)
        loss = eng.forward_loss(x, target, self.lam, self.kind, self.smoothing, scale)
        bwd = eng.bwd
        if self.world > 1 and last_micro:
            works, pos = [], 0
            for mark, a, b in self.buckets:
                stop = len(bwd.calls) if mark == 'end' else bwd.marks[mark]
                if stop > pos:
                    bwd.run_range(pos, stop)
                    pos = stop
                works.append(dist.all_reduce(self.flat_g[a:b], group=self.pg, async_op=True))
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
