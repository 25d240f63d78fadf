"""TEST INFRASTRUCTURE (CPU oracle): restatement of timm.data.mixup.Mixup in its 'batch' mode and of
timm.utils.agc.adaptive_clip_grad, the two timm pieces GA/train.py reaches on its training path
(/root/reference/GA/train.py:544-557 Mixup(**mixup_args), :727-728 mixup_fn(input, target); :752-756 dispatch_clip_grad).

timm is a third-party dependency that is NOT vendored in /root/reference and not installed here (GA/README.md:13 pins
`timm>=0.4.5`); the algorithm below is timm's published one (timm/data/mixup.py: one_hot, mixup_target, rand_bbox,
cutmix_bbox_and_lam, Mixup._params_per_batch / _mix_batch; timm/utils/agc.py: unitwise_norm, adaptive_clip_grad) restated
from its documented behaviour.  The reference holds no fixtures for it: parity of this file is UNPINNED; the tests pin the
HIP kernels against THIS restatement and against closed-form properties (lam = 1 identity, box area, target row sums)."""
import numpy as np
import torch


def one_hot(x, num_classes, on_value=1.0, off_value=0.0):
    x = x.long().view(-1, 1)
    return torch.full((x.size(0), num_classes), off_value, dtype=torch.float32).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0):
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    y1 = one_hot(target, num_classes, on, off)
    y2 = one_hot(target.flip(0), num_classes, on, off)
    return y1 * lam + y2 * (1.0 - lam)


def rand_bbox(img_shape, lam, rng, margin=0.0):
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = rng.randint(0 + margin_y, img_h - margin_y)
    cx = rng.randint(0 + margin_x, img_w - margin_x)
    yl = int(np.clip(cy - cut_h // 2, 0, img_h))
    yh = int(np.clip(cy + cut_h // 2, 0, img_h))
    xl = int(np.clip(cx - cut_w // 2, 0, img_w))
    xh = int(np.clip(cx + cut_w // 2, 0, img_w))
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax, rng):
    img_h, img_w = img_shape[-2:]
    cut_h = rng.randint(int(img_h * minmax[0]), int(img_h * minmax[1]))
    cut_w = rng.randint(int(img_w * minmax[0]), int(img_w * minmax[1]))
    yl = rng.randint(0, img_h - cut_h)
    xl = rng.randint(0, img_w - cut_w)
    return yl, yl + cut_h, xl, xl + cut_w


def cutmix_bbox_and_lam(img_shape, lam, rng, ratio_minmax=None, correct_lam=True):
    box = rand_bbox_minmax(img_shape, ratio_minmax, rng) if ratio_minmax is not None else rand_bbox(img_shape, lam, rng)
    if correct_lam or ratio_minmax is not None:
        yl, yh, xl, xh = box
        lam = 1.0 - (yh - yl) * (xh - xl) / float(img_shape[-2] * img_shape[-1])
    return box, lam


class Mixup:
    """mode='batch' of timm's Mixup; rng = a numpy RandomState (timm draws from the global numpy generator in this order)"""

    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000, rng=None):
        assert mode == 'batch'
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        if cutmix_minmax is not None:
            assert len(cutmix_minmax) == 2
            self.cutmix_alpha = 1.0
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes, self.correct_lam = label_smoothing, num_classes, correct_lam
        self.mixup_enabled = True
        self.rng = rng or np.random.RandomState(0)

    def params_per_batch(self):
        lam, use_cutmix = 1.0, False
        if self.mixup_enabled and self.rng.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = self.rng.rand() < self.switch_prob
                lam_mix = self.rng.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    self.rng.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.0:
                lam_mix = self.rng.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam_mix = self.rng.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                raise AssertionError('one of mixup_alpha > 0, cutmix_alpha > 0, cutmix_minmax not None must be true')
            lam = float(lam_mix)
        return lam, use_cutmix

    def __call__(self, x, target):
        assert len(x) % 2 == 0, 'Batch size should be even when using this'
        lam, use_cutmix = self.params_per_batch()
        x = x.clone()
        if lam != 1.0:
            if use_cutmix:
                (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam, self.rng, self.cutmix_minmax, self.correct_lam)
                x[:, :, yl:yh, xl:xh] = x.flip(0)[:, :, yl:yh, xl:xh]
            else:
                x_flipped = x.flip(0).mul_(1.0 - lam)
                x.mul_(lam).add_(x_flipped)
        return x, mixup_target(target, self.num_classes, lam, self.label_smoothing)


def unitwise_norm(x):
    if x.ndim <= 1:
        return x.norm(2.0)
    return x.norm(2.0, dim=tuple(range(1, x.ndim)), keepdim=True)


def adaptive_clip_grad(params, grads, clip_factor=0.01, eps=1e-3):
    """returns the clipped gradients (timm clips p.grad in place)"""
    out = []
    for p, g in zip(params, grads):
        max_norm = unitwise_norm(p).clamp_(min=eps).mul_(clip_factor)
        grad_norm = unitwise_norm(g)
        clipped = g * (max_norm / grad_norm.clamp(min=1e-6))
        out.append(torch.where(grad_norm < max_norm, g, clipped))
    return out
