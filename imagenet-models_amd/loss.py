"""GA training loss and validation metric on the libgaext kernels.

ga_loss restates GA/train.py:735-745:  sum_k L(out_k, y) + GA_lam * sum_k KL_mean(log_softmax(out_k) || log_softmax(mean_j out_j.detach()))
accuracy / summed-head top-k restate GA/train.py:848-860 (timm `accuracy`).
"""
import torch

from . import ops
from .ops import GA_F32, Plan

_KINDS = {'ce': 0, 'bce': 1}


def _split_target(target, B, NC):
    """class indices (B,) int64 -> (target, None);  dense (B, NC) floating point (mixup / cutmix, SoftTargetCrossEntropy /
    BinaryCrossEntropy of GA/train.py:616-621) -> (None, fp32 dense)"""
    if target.dim() == 2 and target.dtype.is_floating_point:
        if tuple(target.shape) != (B, NC):
            raise ValueError(f'dense target {tuple(target.shape)} does not match the logits ({B}, {NC})')
        return None, target.float().contiguous()
    if target.dim() != 1 or target.dtype != torch.int64:
        raise TypeError('target must be int64 class indices (B,) or a floating-point dense target (B, num_classes)')
    return target.contiguous(), None


class _GALossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, lam, kind, smoothing, thr):
        K, B, NC = logits.shape
        loss = torch.zeros(1, device=logits.device)
        dl = torch.empty_like(logits)
        idx, dense = _split_target(target, B, NC)
        Plan(eager=True).loss_dense_fwd_bwd(logits, None, idx, dense, loss, dl, None, K, B, NC, float(lam), kind, float(smoothing),
                                            float(thr), 1.0, GA_F32)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


def stack_heads(outputs):
    """the engine returns the K heads as unbind(0) views of one (K,B,NC) tensor: recover it without a copy"""
    base = getattr(outputs[0], '_ga_stack', None)
    if base is not None and base.dim() == 3 and base.shape[0] == len(outputs) and all(
            getattr(o, '_ga_stack', None) is base for o in outputs):
        return base
    return torch.stack(list(outputs))


def ga_loss(outputs, target, lam=0.0, kind='ce', smoothing=0.0, bce_target_thresh=None):
    """outputs: list of K (B,NC) fp32 logits; target: int64 (B,) class indices, or a dense (B,NC) target as mixup / cutmix
    produce (then `smoothing` is already inside the target, GA/train.py:617).  Returns the scalar GA loss (differentiable)."""
    logits = stack_heads(outputs)
    if not logits.is_cuda:
        raise RuntimeError('ga_loss runs on the HIP kernels only (no CPU fallback)')
    thr = -1.0 if bce_target_thresh is None else float(bce_target_thresh)
    return _GALossFn.apply(logits.float().contiguous(), target, lam, _KINDS[kind], smoothing, thr)


def heads_topk(outputs, k=5):
    """validate(): output = sum_k out_k.float(); returns (summed logits (B,NC), top-k indices (B,k) int64).  A single (B,NC)
    tensor (the plain-head models) counts as one head, as the reference's validate does for non-list outputs"""
    if isinstance(outputs, torch.Tensor):
        outputs = [outputs]
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
    def forward(ctx, logits, target, lam, kind, smoothing, G, thr):
        n, B, NC = logits.shape
        loss = torch.zeros(1, device=logits.device)
        dl = torch.empty_like(logits)
        has_avg = n == 2 * G
        idx, dense = _split_target(target, B, NC)
        Plan(eager=True).loss_dense_fwd_bwd(logits[:G], logits[G:] if has_avg else None, idx, dense, loss, dl[:G],
                                            dl[G:] if has_avg else None, G, B, NC, float(lam), kind, float(smoothing), float(thr), 1.0,
                                            GA_F32)
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None, None


def map_loss(outputs, target, dec_lam=0.0, kind='ce', smoothing=0.0, bce_target_thresh=None):
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
    thr = -1.0 if bce_target_thresh is None else float(bce_target_thresh)
    return _MAPLossFn.apply(logits.float().contiguous(), target, dec_lam, _KINDS[kind], smoothing, G, thr)


def heads_mean_topk(outputs, k=5):
    """MAP validate (MAP/train.py:1000-1006): output = MEAN of the group logits; returns (mean logits, top-k indices)"""
    s, idx = heads_topk(outputs, k)
    return s / len(outputs), idx
