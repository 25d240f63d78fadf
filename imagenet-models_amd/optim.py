"""Fused flat optimizers (train.py:466,769): one kernel per weight-decay segment over the model's flat fp32
parameter / gradient buffers.  torch.optim.SGD(nesterov) / torch.optim.AdamW arithmetic; timm's create_optimizer_v2
weight-decay rule (no decay for ndim<=1 and *.bias) is baked into the flat layout ([decay | no-decay])."""
import math

import torch

from .ops import Plan, zero_


class _FlatOptimizer:
    def __init__(self, model, lr, weight_decay):
        st = model.flat_state()
        self.model = model
        self.p, self.g = st['params'], st['grads']
        self.n_decay, self.total = st['n_decay'], st['total']
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, initial_lr=lr)]
        self.hp = torch.zeros(8, device=self.p.device)
        self.gen = st['gen']
        self.steps = 0
        self._hp_host = None
        self._range_plans = {}

    @property
    def lr(self):
        return self.param_groups[0]['lr']

    def zero_grad(self, set_to_none=False):
        zero_(self.g)

    def _segments(self):
        segs = []
        if self.n_decay > 0:
            segs.append((0, self.n_decay, 1.0))
        if self.total > self.n_decay:
            segs.append((self.n_decay, self.total - self.n_decay, 0.0))
        return segs

    def _push_hp(self, vals):
        if vals != self._hp_host:
            self.hp.copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=True)
            self._hp_host = vals

    def _mark_dirty(self):
        for e in self.model._engines.values():
            e.weights_dirty = True

    # ---- update by slices of the flat buffers (TrainStep: a gradient bucket is updated, and zeroed, as soon as backward has
    # completed it, on a side stream, while the rest of backward runs).  step_begin / step_range* / step_end == step.
    supports_ranges = False

    def _range_launch(self, plan, off, n, mult):  # pragma: no cover
        raise NotImplementedError

    def _begin(self):  # pragma: no cover
        raise NotImplementedError

    def range_plan(self, lo, hi):
        key = (lo, hi)
        pl = self._range_plans.get(key)
        if pl is None:
            pl = Plan(name=f'{type(self).__name__.lower()}[{lo}:{hi}]')
            for off, n, mult in self._segments():
                a, b = max(lo, off), min(hi, off + n)
                if b > a:
                    self._range_launch(pl, a, b - a, mult)
            pl.zero(self.g[lo:hi])
            self._range_plans[key] = pl
        return pl

    def step_begin(self):
        self.model.check_flat_generation(self.gen, type(self).__name__)
        self._begin()

    def step_range(self, lo, hi, stream=None):
        """update parameters [lo, hi) from their (final) gradients and zero the gradients, on `stream`"""
        self.range_plan(lo, hi).run(stream)

    def step_end(self):
        self._mark_dirty()


class FusedSGD(_FlatOptimizer):
    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=0.0, nesterov=True):
        super().__init__(model, lr, weight_decay)
        self.momentum, self.nesterov = momentum, nesterov
        self.buf = torch.zeros_like(self.p)
        self.plan = Plan(name='sgd')
        for off, n, mult in self._segments():
            self.plan.sgd_step(self.p[off:], self.g[off:], self.buf[off:], self.hp, n, nesterov, mult)

    supports_ranges = True

    def _range_launch(self, plan, off, n, mult):
        plan.sgd_step(self.p[off:], self.g[off:], self.buf[off:], self.hp, n, self.nesterov, mult)

    def _begin(self):
        g = self.param_groups[0]
        self._push_hp([g['lr'], g['weight_decay'], self.momentum, 0.0, 0.0, 0.0, 0.0, 1.0 if self.steps == 0 else 0.0])
        self.steps += 1

    def step(self):
        self.model.check_flat_generation(self.gen, type(self).__name__)
        self._begin()
        self.plan.run()
        self._mark_dirty()

    def state_dict(self):
        return dict(kind='sgd', steps=self.steps, buf=self.buf.clone(), param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd):
        self.steps = sd['steps']
        self.buf.copy_(sd['buf'])
        self.param_groups[0].update(sd['param_groups'][0])


class FusedAdamW(_FlatOptimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(model, lr, weight_decay)
        self.betas, self.eps = betas, eps
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.plan = Plan(name='adamw')
        for off, n, mult in self._segments():
            self.plan.adamw_step(self.p[off:], self.g[off:], self.m[off:], self.v[off:], self.hp, n, mult)

    supports_ranges = True

    def _range_launch(self, plan, off, n, mult):
        plan.adamw_step(self.p[off:], self.g[off:], self.m[off:], self.v[off:], self.hp, n, mult)

    def _begin(self):
        g = self.param_groups[0]
        t = self.steps + 1
        b1, b2 = self.betas
        self._push_hp([g['lr'], g['weight_decay'], b1, b2, self.eps, 1 - b1 ** t, 1 - b2 ** t, 0.0])
        self.steps = t

    def step(self):
        self.model.check_flat_generation(self.gen, type(self).__name__)
        self._begin()
        self.plan.run()
        self._mark_dirty()

    def state_dict(self):
        return dict(kind='adamw', steps=self.steps, m=self.m.clone(), v=self.v.clone(),
                    param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd):
        self.steps = sd['steps']
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.param_groups[0].update(sd['param_groups'][0])


class FusedLamb(_FlatOptimizer):
    """timm.optim.Lamb (bias correction, grad averaging, max_grad_norm 1.0, trust ratio where weight decay applies) -- the
    optimizer of the published GA recipes (GA/README.md:26).  timm is un-vendored: semantics restated, parity unpinned."""
    CHUNK = 16384

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, max_grad_norm=1.0,
                 grad_averaging=True):
        super().__init__(model, lr, weight_decay)
        self.betas, self.eps, self.max_grad_norm, self.grad_averaging = betas, eps, max_grad_norm, grad_averaging
        self.m = torch.zeros_like(self.p)
        self.v = torch.zeros_like(self.p)
        self.u = torch.empty_like(self.p)
        st = model.flat_state()
        rows = []
        for tid, (name, (off, numel)) in enumerate(st['slices'].items()):
            decay = 1 if off < self.n_decay else 0
            for c0 in range(0, numel, self.CHUNK):
                rows.append((off + c0, min(self.CHUNK, numel - c0), tid, decay))
        self.ntensors = len(st['slices'])
        self.chunks = torch.tensor(rows, dtype=torch.int32, device=self.p.device)
        self.norms = torch.zeros(2 * self.ntensors, device=self.p.device)
        self.gsumsq = torch.zeros(1, device=self.p.device)
        self.hp = torch.zeros(9, device=self.p.device)
        pl = self.plan = Plan(name='lamb')
        pl.zero(self.gsumsq)
        pl.zero(self.norms)
        pl.sumsq_f32(self.g, self.total, self.gsumsq)
        pl.lamb_stage1(self.p, self.g, self.m, self.v, self.u, self.hp, self.gsumsq, self.chunks, len(rows), self.norms)
        pl.lamb_stage2(self.p, self.u, self.hp, self.chunks, len(rows), self.norms)

    def step(self):
        self.model.check_flat_generation(self.gen, type(self).__name__)
        g = self.param_groups[0]
        t = self.steps + 1
        b1, b2 = self.betas
        self._push_hp([g['lr'], g['weight_decay'], b1, b2, self.eps, 1 - b1 ** t, 1 - b2 ** t,
                       (1 - b1) if self.grad_averaging else 1.0, self.max_grad_norm])
        self.plan.run()
        self.steps = t
        self._mark_dirty()

    def state_dict(self):
        return dict(kind='lamb', steps=self.steps, m=self.m.clone(), v=self.v.clone(),
                    param_groups=[dict(g) for g in self.param_groups])

    def load_state_dict(self, sd):
        self.steps = sd['steps']
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.param_groups[0].update(sd['param_groups'][0])


def create_optimizer_v2(model, opt='sgd', lr=None, weight_decay=0., momentum=0.9, eps=None, betas=None, **_):
    """timm.optim.create_optimizer_v2 surface: sgd / momentum / nesterov / adamw (north star) and lamb (the published recipes)."""
    opt = opt.lower()
    if opt in ('sgd', 'nesterov'):
        return FusedSGD(model, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True)
    if opt == 'momentum':
        return FusedSGD(model, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=False)
    if opt == 'adamw':
        return FusedAdamW(model, lr=lr, betas=betas or (0.9, 0.999), eps=eps or 1e-8, weight_decay=weight_decay)
    if opt == 'lamb':
        return FusedLamb(model, lr=lr, betas=betas or (0.9, 0.999), eps=eps or 1e-6, weight_decay=weight_decay)
    raise ValueError(f'optimizer {opt!r} is not implemented on the HIP path yet (available: sgd, momentum, nesterov, adamw, lamb)')


class CosineLRScheduler:
    """timm cosine schedule stepped per epoch: linear warm-up from warmup_lr over warmup_epochs, then
    lr = min_lr + 0.5 (base - min_lr)(1 + cos(pi t / T))."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0):
        self.opt, self.T, self.lr_min, self.warmup_t, self.warmup_lr = optimizer, t_initial, lr_min, warmup_t, warmup_lr_init
        self.base = [g['initial_lr'] for g in optimizer.param_groups]
        if warmup_t:
            for g in optimizer.param_groups:
                g['lr'] = warmup_lr_init

    def _lr(self, t, base):
        if t < self.warmup_t:
            return self.warmup_lr + t * (base - self.warmup_lr) / self.warmup_t
        return self.lr_min + 0.5 * (base - self.lr_min) * (1 + math.cos(math.pi * min(t, self.T) / self.T))

    def step(self, epoch, metric=None):
        for g, base in zip(self.opt.param_groups, self.base):
            g['lr'] = self._lr(epoch, base)

    def step_update(self, num_updates, metric=None):
        pass
