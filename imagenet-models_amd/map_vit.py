"""MAP-ViT: a timm VisionTransformer trunk (parameter names and shapes of timm's `vit_*_patch16_*`: patch_embed.proj, cls_token,
pos_embed, blocks.N.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2}, norm) feeding the reference's MAPHead
(/root/reference/MAP/models/map.py:462-539), attached the way /root/reference/MAP/models/map_pit.py:133-144,185-201 attaches it to
its transformer.  BASELINE configs[4] names "MAP-ViT-B/16 @ 384"; the reference registers no such model, so the composition is
builder-defined (see oracle/map_vit_oracle.py for the exact definition) -- the block and the head are the reference's."""
import torch
import torch.nn as nn

from .flat_model import FlatModel, Holder
from .map_convnext import _MAPHead, _init_weights
from .registry import register_model

__all__ = ['MAP_ViT']


class _Attn(Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim)
        self.proj = nn.Linear(dim, dim)


class _Mlp(Holder):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class _VitBlock(Holder):
    def __init__(self, dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)


class _PatchEmbed(Holder):
    def __init__(self, in_chans, dim, ps):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=ps, stride=ps)


class MAP_ViT(FlatModel):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12, num_heads=12,
                 drop_path_rate=0., taps=None, last_dim=384, n_groups=4, n_tokens=2, gram_group=24, bp_dim=384, ca_dim=384,
                 ca_heads=12, head_drop=0.05, head_attn_drop=0.05, math_mode=None, **kwargs):
        """num_heads: the trunk's attention heads (head_dim 64 runs on MFMA); ca_heads: the MAP head's class-attention heads"""
        super().__init__()
        assert in_chans == 3 and img_size % patch_size == 0 and (img_size // patch_size) % 2 == 0 and embed_dim % num_heads == 0
        self.num_classes = num_classes
        self.drop_path_rate = drop_path_rate
        taps = tuple(taps) if taps else (depth // 3, 2 * depth // 3, depth)
        assert taps[-1] == depth and all(0 < t <= depth for t in taps)
        self.cfg = dict(family='map_vit', img_size=img_size, patch_size=patch_size, embed_dim=embed_dim, depth=depth, vit_heads=num_heads,
                        taps=taps, num_classes=num_classes, drop_path_rate=drop_path_rate, last_dim=last_dim, n_groups=n_groups,
                        n_tokens=n_tokens, gram_group=gram_group, bp_dim=bp_dim, bp_groups=1, gram_dim=last_dim, ca_dim=ca_dim,
                        num_heads=ca_heads, mlp_ratio=4, mlp_groups=2, multi_scale_level=0, head_drop=head_drop,
                        head_attn_drop=head_attn_drop)
        gw = img_size // patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, gw * gw + 1, embed_dim) * .02)
        self.patch_embed = _PatchEmbed(in_chans, embed_dim, patch_size)
        self.blocks = nn.ModuleList([_VitBlock(embed_dim) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = _MAPHead(self.cfg, [embed_dim] * (len(taps) + 1))
        self.apply(_init_weights)
        nn.init.trunc_normal_(self.cls_token, std=1e-6)
        self.math_mode = math_mode

    @staticmethod
    def no_weight_decay_param(name, p):
        return p.ndim <= 1 or name.endswith('.bias') or name in ('cls_token', 'pos_embed')      # timm ViT no_weight_decay()

    def make_engine(self, batch, training, mode):
        from .engine_vit import MAPViTEngine
        return MAPViTEngine(self, batch, training, mode)

    def grad_groups(self):
        d = self.cfg['depth']
        return [('heads', ('head.',)), ('stage2', ('norm.',) + tuple(f'blocks.{i}.' for i in range(d // 2, d)))]

    def forward(self, x, pre_logits=False):
        """eval: list of n_groups logits; train: list of [org_out, avg_out] (map.py:519-537)"""
        assert not pre_logits
        outs = super().forward(x)
        if not self.training:
            return outs
        K = self.cfg['n_groups']
        return [[outs[k], outs[K + k]] for k in range(K)]


def _create(variant, pretrained=False, **kwargs):
    kwargs.pop('pretrained_cfg', None)
    kwargs.pop('pretrained_cfg_overlay', None)
    if pretrained:
        raise RuntimeError(f'{variant}: no pretrained weights exist for this builder-defined composition')
    return MAP_ViT(**kwargs)


@register_model
def map_vit_base_patch16_384(pretrained=False, **kwargs):
    return _create('map_vit_base_patch16_384', pretrained, img_size=384, patch_size=16, embed_dim=768, depth=12, num_heads=12, **kwargs)


@register_model
def map_vit_base_patch16_224(pretrained=False, **kwargs):
    return _create('map_vit_base_patch16_224', pretrained, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, **kwargs)


@register_model
def map_vit_small_patch16_224(pretrained=False, **kwargs):
    return _create('map_vit_small_patch16_224', pretrained, img_size=224, patch_size=16, embed_dim=384, depth=12, num_heads=6, **kwargs)
