"""GA training loss and validation metric on the libgaext kernels.

ga_loss restates GA/train.py:735-745:  sum_k L(out_k, y) + GA_lam * sum_k KL_mean(log_softmax(out_k) || log_softmax(mean_j out_j.detach()))
accuracy / summed-head top-k restate GA/train.py:848-860 (timm `accuracy`).
"""
import torch

from . import ops
from .ops import GA_F32, Plan

_KINDS = {'ce': 0, 'bce': 1}


class _GALossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, lam, kind, smoothing):
        K, B, NC = logits.shape
        loss = torch.zeros(1, device=logits.device)
        dl = torch.empty_like(logits)
        Plan(eager=True).loss_fwd_bwd(logits, target, loss, dl, K, B, NC, float(lam), kind, float(smoothing), 1.0, GA_F32)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None


def stack_heads(outputs):
    """the engine returns the K heads as unbind(0) views of one (K,B,NC) tensor: recover it without a copy"""
    base = getattr(outputs[0], '_ga_stack', None)
    if base is not None and base.dim() == 3 and base.shape[0] == len(outputs) and all(
            getattr(o, '_ga_stack', None) is base for o in outputs):
        return base
    return torch.stack(list(outputs))


def ga_loss(outputs, target, lam=0.0, kind='ce', smoothing=0.0):
    """outputs: list of K (B,NC) fp32 logits; target: int64 (B,). Returns the scalar GA loss (differentiable)."""
    logits = stack_heads(outputs)
    if not logits.is_cuda:
        raise RuntimeError('ga_loss runs on the HIP kernels only (no CPU fallback)')
    return _GALossFn.apply(logits.float().contiguous(), target.contiguous(), lam, _KINDS[kind], smoothing)


def heads_topk(outputs, k=5):
    """validate(): output = sum_k out_k.float(); returns (summed logits (B,NC), top-k indices (B,k) int64)"""
    logits = stack_heads(outputs).float().contiguous()
    K, B, NC = logits.shape
    s = torch.empty(B, NC, device=logits.device)
    idx = torch.empty(B, min(k, NC), dtype=torch.int64, device=logits.device)
    Plan(eager=True).heads_topk(logits, K, B, NC, min(k, NC), s, idx)
    return s, idx


def accuracy_from_topk(idx, target, topk=(1, 5)):
    """timm accuracy(): correct[:k].sum() * 100 / B from the (B,maxk) index matrix"""
    correct = idx.eq(target.view(-1, 1))
    return [correct[:, :min(k, idx.shape[1])].any(dim=1).float().sum() * 100.0 / target.shape[0] for k in topk]


class _MAPLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, lam, kind, smoothing, G):
        n, B, NC = logits.shape
        loss = torch.zeros(1, device=logits.device)
        dl = torch.empty_like(logits)
        has_avg = n == 2 * G
        Plan(eager=True).map_loss_fwd_bwd(logits[:G], logits[G:] if has_avg else None, target, loss, dl[:G], dl[G:] if has_avg else None,
                                          G, B, NC, float(lam), kind, float(smoothing), 1.0, GA_F32)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


def map_loss(outputs, target, dec_lam=0.0, kind='ce', smoothing=0.0):
    """MAP/train.py:792-839 (`multi_group_loss`, distill_tokens == 0) on the fused HIP kernel.  outputs: the model's train-mode
    list of [org_out, avg_out] pairs (or a plain list of logits: then it is the GA loss with lam = dec_lam)."""
    if isinstance(outputs[0], (list, tuple)):
        G = len(outputs)
        base = getattr(outputs[0][0], '_ga_stack', None)
        if base is not None and base.dim() == 3 and base.shape[0] == 2 * G and all(
                getattr(o[0], '_ga_stack', None) is base and getattr(o[1], '_ga_stack', None) is base for o in outputs):
            logits = base
        else:
            logits = torch.stack([o[0] for o in outputs] + [o[1] for o in outputs])
    else:
        G = len(outputs)
        logits = stack_heads(outputs)
    if not logits.is_cuda:
        raise RuntimeError('map_loss runs on the HIP kernels only (no CPU fallback)')
    return _MAPLossFn.apply(logits.float().contiguous(), target.contiguous(), dec_lam, _KINDS[kind], smoothing, G)


def heads_mean_topk(outputs, k=5):
    """MAP validate (MAP/train.py:1000-1006): output = MEAN of the group logits; returns (mean logits, top-k indices)"""
    s, idx = heads_topk(outputs, k)
    return s / len(outputs), idx
