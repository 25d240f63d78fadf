"""GA-ConvNeXt on the MI355X-native engine: same registry names, constructor arguments, `state_dict` keys/shapes
and list-of-head-logits output as the reference (/root/reference/GA/ga_convnext.py:320-613), but every FLOP runs
in the hand-written HIP kernels of libgaext (engine.GAEngine).  The nn.Modules below only HOLD parameters and
buffers under the reference's names; they have no forward of their own.
"""
import torch
import torch.nn as nn

from .registry import register_model

__all__ = ['GA_ConvNeXt']


def se_rd_channels(c, rd_ratio=0.25, divisor=8):
    # timm make_divisible(c * rd_ratio, 8, round_limit=0.) as used by create_attn('se', width, rd_ratio=1/4)
    return max(divisor, int(c * rd_ratio + divisor / 2) // divisor * divisor)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('parameter container only: the model runs through engine.GAEngine (libgaext kernels)')


class BlockParams(_Holder):
    """ConvNeXtBlock parameters (ga_convnext.py:86-96)."""

    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Holder()
        self.mlp.fc1 = nn.Linear(dim, 4 * dim)
        self.mlp.fc2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))


class StageParams(_Holder):
    """ConvNeXtStage parameters (ga_convnext.py:116-137)."""

    def __init__(self, in_chs, out_chs, stride, depth, ls_init_value):
        super().__init__()
        if in_chs != out_chs or stride > 1:
            self.downsample = nn.Sequential(nn.LayerNorm(in_chs, eps=1e-6),
                                            nn.Conv2d(in_chs, out_chs, kernel_size=stride, stride=stride))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[BlockParams(out_chs, ls_init_value) for _ in range(depth)])


class SEParams(_Holder):
    def __init__(self, channels):
        super().__init__()
        rd = se_rd_channels(channels)
        self.fc1 = nn.Conv2d(channels, rd, kernel_size=1)
        self.fc2 = nn.Conv2d(rd, channels, kernel_size=1)


class BottleneckParams(_Holder):
    """Bottleneck parameters (ga_convnext.py:254-289)."""

    def __init__(self, inplanes, planes, outplanes):
        super().__init__()
        self.downsample = nn.Sequential(nn.Conv2d(inplanes, outplanes, kernel_size=1), nn.BatchNorm2d(outplanes))
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.se = SEParams(planes)
        self.conv3 = nn.Conv2d(planes, outplanes, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(outplanes)


class ClassAttnParams(_Holder):
    """LayerScaleBlockClassAttn parameters (ga_convnext.py:228-242)."""

    def __init__(self, dim, dim_embed, mlp_groups, init_values=1e-4):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Holder()
        self.attn.q = nn.Linear(dim, dim_embed, bias=False)
        self.attn.k = nn.Linear(dim, dim_embed, bias=False)
        self.attn.v = nn.Linear(dim, dim_embed, bias=False)
        self.attn.proj = nn.Linear(dim_embed, dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Holder()
        self.mlp.fc1 = nn.Conv2d(dim, 4 * dim, kernel_size=1, groups=mlp_groups)
        self.mlp.fc2 = nn.Conv2d(4 * dim, dim, kernel_size=1, groups=mlp_groups)
        self.gamma_1 = nn.Parameter(init_values * torch.ones(dim))
        self.gamma_2 = nn.Parameter(init_values * torch.ones(dim))


def _init_weights(module):
    # ga_convnext.py:508-519 (timm trunc_normal_: absolute bounds [-2, 2])
    if isinstance(module, (nn.Conv2d, nn.Linear)):
        nn.init.trunc_normal_(module.weight, std=.02, a=-2., b=2.)
        if module.bias is not None:
            nn.init.constant_(module.bias, 0)


class GA_ConvNeXt(nn.Module):
    def __init__(self, in_chans=3, num_classes=1000, output_stride=32, patch_size=4,
                 depths=(3, 3, 9, 3, 1), dims=(96, 192, 384, 768, 768), ls_init_value=1e-6, conv_mlp=False,
                 head_init_scale=1., norm_layer=None, drop_rate=0., drop_path_rate=0.,
                 branches=5, gram_embedding_gropus=8, dim_embed=128, stage3_naggre=2, gram_dim=192, gram_layer=True,
                 math_mode=None):
        super().__init__()
        assert output_stride == 32 and patch_size == 4 and in_chans == 3 and len(dims) == 5
        assert not conv_mlp and norm_layer is None and gram_layer, 'only the configuration the reference registers'
        self.num_classes = num_classes
        self.drop_rate = drop_rate
        self.drop_path_rate = drop_path_rate
        self.cfg = dict(depths=tuple(depths), dims=tuple(dims), branches=branches, gram_groups=gram_embedding_gropus,
                        dim_embed=dim_embed, naggre=stage3_naggre, gram_dim=gram_dim, num_heads=8, mlp_groups=4,
                        num_classes=num_classes, patch_size=patch_size, in_chans=in_chans,
                        drop_path_rate=drop_path_rate)
        d = dims
        self.stem = nn.Sequential(nn.Conv2d(in_chans, d[0], kernel_size=patch_size, stride=patch_size),
                                  nn.LayerNorm(d[0], eps=1e-6))
        stages, prev = [], d[0]
        for i in range(4):
            stages.append(StageParams(prev, d[i], 2 if i > 0 else 1, depths[i], ls_init_value))
            prev = d[i]
        cin = sum(d[:-1]) + d[2] * stage3_naggre
        stages.append(BottleneckParams(cin, d[4] // 4, d[4]))
        self.stages = nn.Sequential(*stages)
        self.num_features = d[4]
        self.gram_contraction = nn.ModuleList()
        self.gram_layer = nn.ModuleList()
        self.gram_embedding = nn.ModuleList()
        self.ga = nn.ModuleList()
        self.fc = nn.ModuleList()
        ntri = (gram_dim + 1) * gram_dim // 2
        assert ntri % gram_embedding_gropus == 0 and d[4] % gram_embedding_gropus == 0
        for _ in range(branches):
            self.gram_contraction.append(nn.Sequential(nn.Conv2d(d[4], gram_dim, kernel_size=1), nn.BatchNorm2d(gram_dim)))
            self.gram_layer.append(StageParams(gram_dim, gram_dim, 1, 1, ls_init_value))
            self.gram_embedding.append(nn.Sequential(
                nn.Conv2d(ntri, d[4], kernel_size=1, groups=gram_embedding_gropus), nn.BatchNorm2d(d[4])))
            self.ga.append(ClassAttnParams(d[4], dim_embed, mlp_groups=4))
            self.fc.append(nn.Linear(d[4], num_classes))
        self.apply(_init_weights)
        self.math_mode = math_mode  # None -> bf16 (throughput); 'fp32' -> parity math mode
        self._engines = {}
        self._flat = None

    # ------------------------------------------------------------------------------------------
    # flat parameter / gradient storage (one fp32 buffer each: [decay | no-decay], timm's rule)
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
        self._flat = dict(params=flat, grads=grads, n_decay=sum(p.numel() for _, p in decay), total=total,
                          slices=slices)
        self._engines = {}

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        first = next(self.parameters())
        if first.is_cuda:
            self._flatten()
        else:
            self._flat = None
            self._engines = {}
        return out

    def flat_state(self):
        if self._flat is None:
            raise RuntimeError('GA_ConvNeXt must be moved to the GPU (model.cuda()) before use: the product path has '
                               'no CPU implementation')
        return self._flat

    def zero_grad(self, set_to_none=False):
        """Gradients live in one flat fp32 buffer that the wgrad kernels accumulate into: zero it in place."""
        if self._flat is not None:
            self._flat['grads'].zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if self._flat is not None:
            # keep the flat views: copy values instead of re-assigning tensors
            own = self.state_dict()
            missing = [k for k in own if k not in state_dict]
            unexpected = [k for k in state_dict if k not in own]
            if strict and (missing or unexpected):
                raise RuntimeError(f'load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}')
            with torch.no_grad():
                for k, v in state_dict.items():
                    if k in own:
                        own[k].copy_(v.to(own[k].device))
            for e in self._engines.values():
                e.weights_dirty = True
            return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)
        return super().load_state_dict(state_dict, strict=strict, assign=assign)

    # ------------------------------------------------------------------------------------------
    def engine(self, batch, training):
        from .engine import GAEngine
        mode = self.math_mode or 'bf16'
        key = (batch, bool(training), mode)
        if key not in self._engines:
            self._engines[key] = GAEngine(self, batch, bool(training), mode)
        return self._engines[key]

    def forward(self, x):
        """(B,3,224,224) float -> list of `branches` per-head logits (B,num_classes), fp32 (ga_convnext.py:487-505)."""
        if not x.is_cuda:
            raise RuntimeError('GA_ConvNeXt.forward needs a CUDA/HIP tensor: there is no CPU fallback')
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


def _create(variant, pretrained=False, **kwargs):
    # timm build_model_with_cfg: pops cfg kwargs; GA configs carry no weight URL (ga_convnext.py:37-41)
    for k in ('pretrained_cfg', 'pretrained_cfg_overlay', 'features_only', 'default_cfg'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: the reference publishes no pretrained weights (url is empty)')
    return GA_ConvNeXt(**kwargs)


@register_model
def ga_convnext_tiny_688(pretrained=False, **kwargs):
    return _create('ga_convnext_tiny', pretrained, depths=[3, 3, 9, 3, 1], dims=[96, 192, 384, 688, 688],
                   gram_embedding_gropus=8, dim_embed=168, stage3_naggre=2, gram_dim=192, **kwargs)


@register_model
def ga_convnext_tiny_768(pretrained=False, **kwargs):
    return _create('ga_convnext_tiny', pretrained, depths=[3, 3, 9, 3, 1], dims=[96, 192, 384, 768, 768],
                   gram_embedding_gropus=8, dim_embed=192, stage3_naggre=2, gram_dim=192, **kwargs)


@register_model
def ga_convnext_small_688(pretrained=False, **kwargs):
    return _create('ga_convnext_small', pretrained, depths=[3, 3, 27, 3, 1], dims=[96, 192, 384, 688, 688],
                   gram_embedding_gropus=8, dim_embed=168, stage3_naggre=4, gram_dim=192, **kwargs)


@register_model
def ga_convnext_small_768(pretrained=False, **kwargs):
    return _create('ga_convnext_small', pretrained, depths=[3, 3, 27, 3, 1], dims=[96, 192, 384, 768, 768],
                   gram_embedding_gropus=8, dim_embed=192, stage3_naggre=4, gram_dim=192, **kwargs)


@register_model
def ga_convnext_base_976(pretrained=False, **kwargs):
    return _create('ga_convnext_base', pretrained, depths=[3, 3, 27, 3, 1], dims=[128, 256, 512, 976, 976],
                   gram_embedding_gropus=8, dim_embed=240, stage3_naggre=4, gram_dim=192, **kwargs)


@register_model
def ga_convnext_base_1024(pretrained=False, **kwargs):
    return _create('ga_convnext_base', pretrained, depths=[3, 3, 27, 3, 1], dims=[128, 256, 512, 1024, 1024],
                   gram_embedding_gropus=8, dim_embed=256, stage3_naggre=4, gram_dim=192, **kwargs)


# README aliases that do not resolve in the reference (GA/README.md:26,53) -- bound to the primary variants
@register_model
def ga_convnext_tiny(pretrained=False, **kwargs):
    return ga_convnext_tiny_768(pretrained, **kwargs)


@register_model
def ga_convnext_small(pretrained=False, **kwargs):
    return ga_convnext_small_768(pretrained, **kwargs)


@register_model
def ga_convnext_base(pretrained=False, **kwargs):
    return ga_convnext_base_1024(pretrained, **kwargs)
