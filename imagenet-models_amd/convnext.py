"""Plain ConvNeXt: the `global_pool='avg'` branch of /root/reference/MAP/models/map_convnext.py (ConvNeXt :44-140; registered
baselines convnext_tiny :186-196 and convnext_small :214-224): the trunk of map_convnext, `norm` = LayerNorm(dims[-1], 1e-6)
on the global average pool, `head` = Linear.  One (B, num_classes) output, not a list.  Compute: engine_convnext."""
import torch.nn as nn

from .flat_model import FlatModel
from .map_convnext import _Block, _LN, _init_weights
from .registry import register_model

__all__ = ['ConvNeXt']


class ConvNeXt(FlatModel):
    def __init__(self, in_chans=3, num_classes=1000, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.,
                 layer_scale_init_value=1e-6, head_init_scale=1., global_pool='avg', math_mode=None, **kwargs):
        super().__init__()
        assert global_pool == 'avg' and in_chans == 3 and layer_scale_init_value > 0, \
            "this class is the global_pool='avg' ConvNeXt; the 'mmcap' configuration is MAP_ConvNeXt"
        depths, dims = tuple(depths), tuple(dims)
        self.num_classes = num_classes
        self.drop_path_rate = drop_path_rate
        self.cfg = dict(family='convnext', depths=depths, dims=dims, num_classes=num_classes, drop_path_rate=drop_path_rate, naggre=0)
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4), _LN(dims[0])))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(_LN(dims[i]), nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList([nn.Sequential(*[_Block(dims[i], layer_scale_init_value) for _ in range(depths[i])])
                                     for i in range(4)])
        self.norm = nn.LayerNorm(dims[-1], eps=1e-6)
        self.head = nn.Linear(dims[-1], num_classes)
        self.apply(_init_weights)
        self.head.weight.data.mul_(head_init_scale)
        self.head.bias.data.mul_(head_init_scale)
        self.math_mode = math_mode

    def make_engine(self, batch, training, mode):
        from .engine_convnext import ConvNeXtEngine
        return ConvNeXtEngine(self, batch, training, mode)

    def grad_groups(self):
        return [('heads', ('head.', 'norm.')), ('stage3', ('stages.3.', 'downsample_layers.3.')),
                ('stage2', ('stages.2.', 'downsample_layers.2.')), ('stage1', ('stages.1.', 'downsample_layers.1.'))]

    def forward(self, x, pre_logits=False):
        assert not pre_logits
        return super().forward(x)[0]


def _create(variant, pretrained=False, **kwargs):
    kwargs.pop('pretrained_cfg', None)
    kwargs.pop('pretrained_cfg_overlay', None)
    kwargs.pop('in_22k', None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a network fetch (map_convnext.py:191-194); load a local file with '
                           f'load_checkpoint instead')
    return ConvNeXt(**kwargs)


@register_model
def convnext_tiny(pretrained=False, **kwargs):
    return _create('convnext_tiny', pretrained, depths=[3, 3, 9, 3], dims=[96, 192, 384, 768], **kwargs)


@register_model
def convnext_small(pretrained=False, **kwargs):
    return _create('convnext_small', pretrained, depths=[3, 3, 27, 3], dims=[96, 192, 384, 768], **kwargs)
