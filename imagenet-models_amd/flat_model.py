"""FlatModel: the nn.Module side of every model family of this package.

The nn.Modules of a model only HOLD parameters and buffers under the reference's `state_dict` names; every FLOP runs
in the engine's launch plans over libgaext.  FlatModel gives them the storage layout the engines, the fused optimizers
and the gradient all-reduce share:

  * ONE flat fp32 parameter buffer `[decay | no-decay]` in registration order (timm's weight-decay rule -- no decay for
    `ndim <= 1` and `*.bias`, GA/train.py:466 -> create_optimizer_v2 -- baked into the layout) and ONE flat fp32 gradient
    buffer; every `param.data` / `param.grad` is a view into them;
  * a cache of engines keyed by (batch, train|eval, math mode);
  * `forward(x)` -> list of per-head fp32 logits, through one autograd node (engine.GAFunction).

Subclasses set `engine_cls` (a callable (model, batch, training, mode) -> engine) and build their holders in __init__.
"""
import torch
import torch.nn as nn


class Holder(nn.Module):
    """parameter container only: never called"""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter container only: the model runs through its engine (libgaext kernels)')


class FlatModel(nn.Module):
    def __init__(self):
        super().__init__()
        self.math_mode = None   # None -> bf16 (throughput); 'fp32' -> parity math mode
        self._engines = {}
        self._flat = None
        self._flat_gen = 0      # bumped whenever the flat buffers are re-created: holders of raw pointers check it

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def no_weight_decay_param(name, p):
        return p.ndim <= 1 or name.endswith('.bias')

    def _flatten(self):
        params = list(self.named_parameters())
        dev = params[0][1].device
        decay = [(n, p) for n, p in params if not self.no_weight_decay_param(n, p)]
        nodecay = [(n, p) for n, p in params if self.no_weight_decay_param(n, p)]
        order = decay + nodecay
        total = sum(p.numel() for _, p in order)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        grads = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        slices = {}
        with torch.no_grad():
            for n, p in order:
                k = p.numel()
                flat[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + k].view(p.shape)
                p.grad = grads[off:off + k].view(p.shape)
                slices[n] = (off, k)
                off += k
        # BatchNorm running statistics in ONE flat fp32 buffer too (DDP's per-forward buffer broadcast and timm's
        # distribute_bn, GA/train.py:514,665-674, then are one collective each); num_batches_tracked in one int64 buffer
        fbufs = [(n, b) for n, b in self.named_buffers() if b.dtype == torch.float32]
        ibufs = [(n, b) for n, b in self.named_buffers() if n.endswith('num_batches_tracked')]   # not e.g. GramToken.bp_index
        fb = torch.empty(sum(b.numel() for _, b in fbufs), dtype=torch.float32, device=dev)
        ib = torch.empty(sum(b.numel() for _, b in ibufs), dtype=torch.int64, device=dev)
        with torch.no_grad():
            for flatbuf, lst in ((fb, fbufs), (ib, ibufs)):
                off = 0
                for n, b in lst:
                    k = b.numel()
                    flatbuf[off:off + k].copy_(b.detach().reshape(-1))
                    b.data = flatbuf[off:off + k].view(b.shape)
                    off += k
        self._flat_gen += 1
        self._flat = dict(params=flat, grads=grads, n_decay=sum(p.numel() for _, p in decay), total=total,
                          slices=slices, gen=self._flat_gen, buffers=fb, ibuffers=ib)
        self._nbt_pending = 0
        self._engines = {}

    def _is_flat_on(self, first):
        f = self._flat
        if f is None or not first.is_cuda or first.dtype != torch.float32 or f['params'].device != first.device:
            return False
        lo, hi = f['params'].data_ptr(), f['params'].data_ptr() + 4 * f['total']
        return all(lo <= p.data_ptr() < hi for p in self.parameters())

    def _apply(self, fn, recurse=True):
        """model.cuda() / .to(...) / .float(): (re)build the flat buffers when the parameters moved.  A call that leaves
        every parameter where it is (a second .cuda(), .to(memory_format=...) -- normal timm flow after the optimizer
        exists, GA/train.py:444-446) keeps the buffers: optimizers, TrainStep and ModelEma hold raw pointers into them."""
        before = [p.data_ptr() for p in self.parameters()]
        out = super()._apply(fn, recurse)
        first = next(self.parameters())
        if first.is_cuda:
            unchanged = self._flat is not None and before == [p.data_ptr() for p in self.parameters()]
            if not (unchanged and self._is_flat_on(first)):
                self._flatten()
        else:
            self._flat = None
            self._engines = {}
            self._flat_gen += 1
        return out

    def flat_state(self):
        if self._flat is None:
            raise RuntimeError(f'{type(self).__name__} must be moved to the GPU (model.cuda()) before use: the product '
                               'path has no CPU implementation')
        return self._flat

    def check_flat_generation(self, gen, who):
        if self._flat is None or self._flat['gen'] != gen:
            raise RuntimeError(f'{who}: the model\'s flat parameter buffers were re-created (model.cuda()/.to() moved the '
                               'parameters after this object was built); rebuild it')

    def zero_grad(self, set_to_none=False):
        """Gradients live in one flat fp32 buffer that the wgrad kernels accumulate into: zero it in place (hipMemsetAsync
        on the current stream through the C ABI)."""
        if self._flat is not None:
            from .ops import zero_
            zero_(self._flat['grads'])
        else:
            super().zero_grad(set_to_none=set_to_none)

    # nn.BatchNorm2d.num_batches_tracked: counted on the host per training forward and written into the int64 buffers
    # when somebody looks (state_dict / checkpoint), so the step itself issues no launch for it
    def count_training_forward(self):
        self._nbt_pending = getattr(self, '_nbt_pending', 0) + 1

    def flush_counters(self):
        n = getattr(self, '_nbt_pending', 0)
        if n and self._flat is not None and self._flat['ibuffers'].numel():
            self._flat['ibuffers'] += n
        self._nbt_pending = 0

    def state_dict(self, *args, **kwargs):
        self.flush_counters()
        return super().state_dict(*args, **kwargs)

    def convert_sync_batchnorm(self, comm):
        """--sync-bn (GA/train.py:449-455, torch.nn.SyncBatchNorm.convert_sync_batchnorm): every train-mode BatchNorm of the engines
        built FROM NOW ON takes its batch statistics over all ranks -- the column sums the producing GEMM's epilogue left are
        all-reduced through `comm` (an imagenet_models_amd.NativeComm) before the finalisation, the two sums of the backward
        pass likewise.  comm = None switches it off again.  Cached engines are dropped."""
        self.sync_bn_comm = comm
        self._engines = {}
        return self

    def grad_groups(self):
        """[(backward-plan mark, parameter-name prefixes whose gradients are final at that mark)] in backward-completion
        order; parameters matched by no group are final at the end of backward.  Used to cut the flat gradient buffer into
        all-reduce buckets that start while the rest of backward still runs (trainer.make_buckets)."""
        return []

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if self._flat is not None:
            own = self.state_dict()
            missing = [k for k in own if k not in state_dict]
            unexpected = [k for k in state_dict if k not in own]
            if strict and (missing or unexpected):
                raise RuntimeError(f'load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}')
            bad = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in state_dict.items()
                   if k in own and tuple(v.shape) != tuple(own[k].shape)]
            if bad:
                raise RuntimeError(f'load_state_dict: size mismatch for {bad[:5]} (name, checkpoint shape, model shape)')
            with torch.no_grad():
                for k, v in state_dict.items():
                    if k in own:
                        own[k].copy_(v.to(own[k].device))
            for e in self._engines.values():
                e.weights_dirty = True
            return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    # ------------------------------------------------------------------------------------------
    def make_engine(self, batch, training, mode):  # pragma: no cover
        raise NotImplementedError

    def engine(self, batch, training):
        mode = self.math_mode or 'bf16'
        key = (batch, bool(training), mode)
        if key not in self._engines:
            self._engines[key] = self.make_engine(batch, bool(training), mode)
        return self._engines[key]

    def forward(self, x):
        """(B,3,H,W) float -> list of per-head logits (B,num_classes), fp32"""
        if not x.is_cuda:
            raise RuntimeError(f'{type(self).__name__}.forward needs a CUDA/HIP tensor: there is no CPU fallback')
        from .engine import GAFunction
        eng = self.engine(x.shape[0], self.training)
        if self.training and torch.is_grad_enabled():
            logits = GAFunction.apply(eng, x, eng.anchor)
        else:
            logits = eng.forward(x)
        outs = list(logits.unbind(0))
        for o in outs:
            o._ga_stack = logits   # lets ga_loss / heads_topk use the stacked (K,B,NC) tensor without a copy
        return outs

    def set_math_mode(self, mode):
        assert mode in (None, 'bf16', 'fp32')
        self.math_mode = mode
        return self
