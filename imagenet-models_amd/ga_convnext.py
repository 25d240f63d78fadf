"""GA-ConvNeXt on the MI355X-native engine: same registry names, constructor arguments, `state_dict` keys/shapes
and list-of-head-logits output as the reference (/root/reference/GA/ga_convnext.py:320-613), but every FLOP runs
in the hand-written HIP kernels of libgaext (engine.GAEngine).  The nn.Modules below only HOLD parameters and
buffers under the reference's names; they have no forward of their own.
"""
import torch
import torch.nn as nn

from .flat_model import FlatModel, Holder as _Holder
from .registry import register_model

__all__ = ['GA_ConvNeXt']


def se_rd_channels(c, rd_ratio=0.25, divisor=8):
    # timm make_divisible(c * rd_ratio, 8, round_limit=0.) as used by create_attn('se', width, rd_ratio=1/4)
    return max(divisor, int(c * rd_ratio + divisor / 2) // divisor * divisor)


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


class GA_ConvNeXt(FlatModel):
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

    def make_engine(self, batch, training, mode):
        from .engine import GAEngine
        return GAEngine(self, batch, training, mode)

    def grad_groups(self):
        return [('heads', ('stages.4.', 'gram_contraction.', 'gram_layer.', 'gram_embedding.', 'ga.', 'fc.')),
                ('stage3', ('stages.3.',)), ('stage2', ('stages.2.',)), ('stage1', ('stages.1.',))]


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
