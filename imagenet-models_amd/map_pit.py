"""MAP-PiT: the pooling transformer of /root/reference/MAP/models/map_pit.py (PoolingTransformer :84-201 with pool_type='map';
registered variant map_pit_s :224-251).  Parameter names / shapes / registration order are the reference's state_dict:
pos_embed (1, C0, w, w), patch_embed.conv, transformers.S.blocks.J.{norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2} (timm's
ViT Block), pools.S.conv (depthwise 3x3 / stride 2, channel multiplier = C[S+1] / C[S]), head.* (MAPHead).  Compute lives in
engine_pit.MAPPiTEngine; this class only holds parameters."""
import math

import torch
import torch.nn as nn

from .flat_model import FlatModel, Holder
from .map_convnext import _MAPHead
from .map_vit import _VitBlock
from .registry import register_model

__all__ = ['MAP_PiT']


class _ConvEmbedding(Holder):
    def __init__(self, in_chans, dim, ps, stride):
        super().__init__()
        self.conv = nn.Conv2d(in_chans, dim, kernel_size=ps, stride=stride, padding=0, bias=True)


class _Transformer(Holder):
    def __init__(self, dim, depth):
        super().__init__()
        self.blocks = nn.ModuleList([_VitBlock(dim) for _ in range(depth)])


class _HeadPooling(Holder):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size=3, padding=1, stride=2, padding_mode='zeros', groups=cin)


class MAP_PiT(FlatModel):
    def __init__(self, image_size=224, patch_size=16, stride=8, base_dims=(48, 48, 48), depth=(2, 6, 4), heads=(3, 6, 12), mlp_ratio=4,
                 num_classes=1000, in_chans=3, attn_drop_rate=0., drop_rate=0., drop_path_rate=0., pool_type='map', last_dim=384,
                 n_groups=4, n_tokens=3, gram_group=24, self_distill_token=True, gram=True, multi_scale_level=2, head_drop=0.05,
                 head_attn_drop=0.05, math_mode=None, **kwargs):
        super().__init__()
        assert pool_type == 'map' and gram and self_distill_token and in_chans == 3 and mlp_ratio == 4, \
            'only the MAP configuration of the registered map_pit_s model'
        assert attn_drop_rate == 0. and drop_rate == 0., 'the reference recipes run PiT without token / attention dropout'
        base_dims, depth, heads = tuple(base_dims), tuple(depth), tuple(heads)
        assert len(base_dims) == len(depth) == len(heads) == 3 and multi_scale_level == 2
        dims = tuple(b * h for b, h in zip(base_dims, heads))
        assert all(dims[i + 1] % dims[i] == 0 for i in range(2)), 'conv_head_pooling is depthwise: C[s+1] must be a multiple of C[s]'
        width = math.floor((image_size - patch_size) / stride + 1)
        self.num_classes = num_classes
        self.drop_path_rate = drop_path_rate
        # the MAPHead arguments PoolingTransformer hard-codes (:137-143): ca_dim 192, 12 heads, mlp_ratio 4, mlp_groups 2, bp_groups 1
        self.cfg = dict(family='map_pit', img_size=image_size, patch_size=patch_size, stride=stride, base_dims=base_dims, depth=depth,
                        heads=heads, dims=dims, width=width, num_classes=num_classes, drop_path_rate=drop_path_rate, last_dim=last_dim,
                        n_groups=n_groups, n_tokens=n_tokens, gram_group=gram_group, bp_dim=last_dim, bp_groups=1, gram_dim=last_dim,
                        ca_dim=192, num_heads=12, mlp_ratio=4, mlp_groups=2, multi_scale_level=multi_scale_level, head_drop=head_drop,
                        head_attn_drop=head_attn_drop)
        self.pos_embed = nn.Parameter(torch.randn(1, dims[0], width, width))
        self.patch_embed = _ConvEmbedding(in_chans, dims[0], patch_size, stride)
        self.transformers = nn.ModuleList([_Transformer(dims[s], depth[s]) for s in range(3)])
        self.pools = nn.ModuleList([_HeadPooling(dims[s], dims[s + 1]) for s in range(2)])
        self.head = _MAPHead(self.cfg, [dims[0]] + list(dims))
        nn.init.trunc_normal_(self.pos_embed, std=.02)          # :150; Conv / Linear keep PyTorch's default init (:153-156)
        self.math_mode = math_mode

    @staticmethod
    def no_weight_decay_param(name, p):
        return p.ndim <= 1 or name.endswith('.bias') or name in ('pos_embed', 'cls_token')       # :158-160 + the usual 1-d rule

    def make_engine(self, batch, training, mode):
        from .engine_pit import MAPPiTEngine
        return MAPPiTEngine(self, batch, training, mode)

    def grad_groups(self):
        return [('heads', ('head.',)), ('stage3', ('transformers.2.', 'pools.1.')), ('stage2', ('transformers.1.', 'pools.0.'))]

    def forward(self, x, pre_logits=False):
        """eval: list of n_groups logits; train: list of [org_out, avg_out] (map.py:519-537)"""
        assert not pre_logits
        outs = super().forward(x)
        if not self.training:
            return outs
        K = self.cfg['n_groups']
        return [[outs[k], outs[K + k]] for k in range(K)]


@register_model
def map_pit_s(pretrained=False, **kwargs):
    kwargs.pop('pretrained_cfg', None)
    kwargs.pop('pretrained_cfg_overlay', None)
    if pretrained:
        raise RuntimeError('map_pit_s: pretrained weights are a network download (map_pit.py:244-248); load a state_dict instead')
    return MAP_PiT(image_size=224, patch_size=16, stride=8, base_dims=[48, 48, 48], depth=[2, 6, 4], heads=[3, 6, 12], mlp_ratio=4,
                   pool_type='map', last_dim=384, n_groups=2, n_tokens=4, gram_group=32, **kwargs)
