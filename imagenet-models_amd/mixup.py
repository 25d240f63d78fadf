"""mixup / cutmix on the device for the training step (timm.data.Mixup, mode 'batch', as /root/reference/GA/train.py:544-557
builds it and :727-728 applies it to every batch when --mixup / --cutmix are set -- the reference's defaults 0.2 / 1.0 have it
on).  lam and the cut box are drawn on the host from numpy's generator exactly as timm draws them; the blend / box copy and
the dense smoothed target are HIP kernels (ga_mixup_batch, ga_mixup_target).  The dense (B, num_classes) target goes to
ga_loss / map_loss / TrainStep, which then evaluate SoftTargetCrossEntropy (or BinaryCrossEntropy) on it (train.py:616-621).

Modes 'pair' and 'elem' of timm's Mixup are not built (GA/README's recipes use the default 'batch')."""
import numpy as np
import torch

from . import ops


class Mixup:
    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode='batch',
                 correct_lam=True, label_smoothing=0.1, num_classes=1000, rng=None):
        if mode != 'batch':
            raise NotImplementedError(f"Mixup mode {mode!r}: only 'batch' is built")
        self.mixup_alpha, self.cutmix_alpha, self.cutmix_minmax = mixup_alpha, cutmix_alpha, cutmix_minmax
        if cutmix_minmax is not None:
            assert len(cutmix_minmax) == 2
            self.cutmix_alpha = 1.0          # force cutmix alpha == 1.0 when minmax active to keep logic simple & safe
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes, self.correct_lam = label_smoothing, num_classes, correct_lam
        self.mixup_enabled = True            # set False to turn it off (--mixup-off-epoch, train.py:705-709)
        self.rng = rng if rng is not None else np.random      # timm draws from the global numpy generator
        self.last = None                     # (lam, use_cutmix, box) of the last call, for logging / tests

    def _params_per_batch(self):
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

    def _box(self, H, W, lam):
        if self.cutmix_minmax is not None:
            lo, hi = self.cutmix_minmax
            cut_h = self.rng.randint(int(H * lo), int(H * hi))
            cut_w = self.rng.randint(int(W * lo), int(W * hi))
            yl = self.rng.randint(0, H - cut_h)
            xl = self.rng.randint(0, W - cut_w)
            box = (yl, yl + cut_h, xl, xl + cut_w)
        else:
            ratio = np.sqrt(1 - lam)
            cut_h, cut_w = int(H * ratio), int(W * ratio)
            cy, cx = self.rng.randint(0, H), self.rng.randint(0, W)
            box = (int(np.clip(cy - cut_h // 2, 0, H)), int(np.clip(cy + cut_h // 2, 0, H)),
                   int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cx + cut_w // 2, 0, W)))
        if self.correct_lam or self.cutmix_minmax is not None:
            lam = 1.0 - (box[1] - box[0]) * (box[3] - box[2]) / float(H * W)
        return box, lam

    def __call__(self, x, target):
        """x: (B, C, H, W) fp32 on the device, target: (B,) int64 -> (mixed x (a new tensor), dense target (B, num_classes))"""
        if not x.is_cuda:
            raise RuntimeError('Mixup runs on the HIP kernels only (no CPU fallback)')
        if not x.is_floating_point():
            raise TypeError(f'Mixup expects a normalised floating-point batch, got {x.dtype}: normalise uint8 input first '
                            '(TrainStep does; engine._normalize_u8)')
        B, _, H, W = x.shape
        assert B % 2 == 0, 'Batch size should be even when using this'
        lam, use_cutmix = self._params_per_batch()
        box = (0, 0, 0, 0)
        x = x.float().contiguous()
        p = ops.Plan(eager=True)
        if lam != 1.0:
            if use_cutmix:
                box, lam = self._box(H, W, lam)
            out = torch.empty_like(x)
            p.mixup_batch(x, out, lam, use_cutmix, box)
        else:
            out = x
        dense = torch.empty(B, self.num_classes, device=x.device, dtype=torch.float32)
        p.mixup_target(target.contiguous(), dense, self.num_classes, lam, self.label_smoothing)
        self.last = (lam, use_cutmix, box)
        return out, dense
