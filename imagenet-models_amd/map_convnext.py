"""MAP-ConvNeXt on the MI355X-native engine: registry names, constructor arguments, `state_dict` keys / shapes and outputs
of the reference's MAP ConvNeXt (/root/reference/MAP/models/map_convnext.py:43-140 with global_pool='mmcap', its MAPHead
from /root/reference/MAP/models/map.py:462-539); every FLOP runs in the hand-written HIP kernels of libgaext
(engine_map.MAPEngine).  The nn.Modules below only HOLD parameters and buffers under the reference's names.

Outputs (map.py:519-537): eval -> list of n_groups (B, num_classes) logits (the `heads`); train -> list of
[org_out, avg_out] pairs (avg_out from the self-distillation token through `self_dt_heads`), which
`map_loss` (MAP/train.py:792-839) consumes.
"""
import torch
import torch.nn as nn

from .flat_model import FlatModel, Holder
from .registry import register_model

__all__ = ['MAP_ConvNeXt']


class _LN(Holder):
    """map_convnext.LayerNorm (:143-170): weight / bias only"""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Block(Holder):
    """map_convnext.Block (:16-25): registration order dwconv, norm, pwconv1, pwconv2; gamma is the module's own parameter"""

    def __init__(self, dim, ls_init=1e-6):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = _LN(dim)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(ls_init * torch.ones(dim))


class _CABlock(Holder):
    """map.CABlock with ClassAttention / GroupConvMlp (map.py:147-169, 69-98, 43-54); `interactive` adds the two head-mixing
    linears w1 / w2 (:96-98); in_dim != dim (`dim_mismatch`, :85-90, 165-167): class rows of width in_dim with their own q / k1 / v1
    and norm1_1, image rows k2 / v2 and norm1_2"""

    def __init__(self, dim, ca_dim, mlp_ratio, mlp_groups, num_heads=0, interactive=False, in_dim=None):
        super().__init__()
        in_dim = in_dim or dim
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Holder()
        self.attn.proj = nn.Linear(ca_dim, dim)
        if in_dim != dim:
            self.attn.q = nn.Linear(in_dim, ca_dim, bias=True)
            self.attn.k1 = nn.Linear(in_dim, ca_dim, bias=True)
            self.attn.v1 = nn.Linear(in_dim, ca_dim, bias=True)
            self.attn.k2 = nn.Linear(dim, ca_dim, bias=True)
            self.attn.v2 = nn.Linear(dim, ca_dim, bias=True)
        else:
            self.attn.q = nn.Linear(dim, ca_dim, bias=True)
            self.attn.k = nn.Linear(dim, ca_dim, bias=True)
            self.attn.v = nn.Linear(dim, ca_dim, bias=True)
        if interactive:
            self.attn.w1 = nn.Linear(num_heads, num_heads)
            self.attn.w2 = nn.Linear(num_heads, num_heads)
        hid = int(dim * mlp_ratio)
        self.mlp = Holder()
        self.mlp.fc1 = nn.Conv2d(dim, hid, kernel_size=1, groups=mlp_groups)
        self.mlp.fc2 = nn.Conv2d(hid, dim, kernel_size=1, groups=mlp_groups)
        if in_dim != dim:
            self.norm1_1 = nn.LayerNorm(in_dim, eps=1e-6)
            self.norm1_2 = nn.LayerNorm(dim, eps=1e-6)
        else:
            self.norm1 = nn.LayerNorm(dim, eps=1e-6)


class _GramToken(Holder):
    """map.GramToken (:187-208)"""

    def __init__(self, ch_dim, num_groups, num_tokens, bp_groups, bp_dim, out_dim):
        super().__init__()
        tri = torch.triu_indices(bp_dim, bp_dim)
        self.register_buffer('bp_index', tri[0] * bp_dim + tri[1])
        gram_dim = bp_dim * (bp_dim + 1) // 2
        self.ch_reduction = nn.Sequential(nn.Conv2d(ch_dim, bp_dim, 1, bias=False, groups=bp_groups), nn.BatchNorm2d(bp_dim))
        self.gram_blk = nn.Identity()
        self.bp_reduction = nn.Sequential(nn.Conv2d(gram_dim, out_dim * num_tokens, 1, bias=False, groups=num_groups),
                                          nn.BatchNorm2d(out_dim * num_tokens))


class _CAP(Holder):
    def __init__(self, cfg):
        super().__init__()
        self.attention = nn.Sequential(_CABlock(cfg['last_dim'], cfg['ca_dim'], cfg['mlp_ratio'], cfg['mlp_groups'], cfg['num_heads'],
                                                cfg.get('interactive', False), in_dim=cfg['gram_dim']))
        self.gram_token_extraction = _GramToken(cfg['last_dim'], cfg['gram_group'], cfg['n_tokens'], cfg['bp_groups'],
                                                cfg['bp_dim'], cfg['gram_dim'])


class _NormHead(Holder):
    def __init__(self, ch, num_classes):
        super().__init__()
        self.norm = nn.LayerNorm(ch)
        self.head = nn.Linear(ch, num_classes)


class _SplitNormHead(Holder):
    """map.SplitNormHead (:415-441): per token its own LayerNorm + Linear, outputs summed"""

    def __init__(self, ch, num_classes, nt):
        super().__init__()
        self.norm = nn.ModuleList([nn.LayerNorm(ch // nt) for _ in range(nt)])
        self.head = nn.ModuleList([nn.Linear(ch // nt, num_classes) for _ in range(nt)])


class _MAPHead(Holder):
    """map.MAPHead (:462-492): mmcap (MAP: the CAPs, then multi_scale), heads, self_dt_heads"""

    def __init__(self, cfg, channels):
        super().__init__()
        L, G, T = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens']
        self.mmcap = Holder()
        self.mmcap.mmcap = nn.ModuleList([_CAP(cfg) for _ in range(G)])
        ms = Holder()
        ms.concat_conv = nn.Sequential(nn.Conv2d(sum(channels), L, 1, bias=False), nn.BatchNorm2d(L))
        self.mmcap.multi_scale = ms
        hfn = cfg.get('head_fn', 'norm')
        if hfn == 'split':
            self.heads = nn.ModuleList([_SplitNormHead(L * T, cfg['num_classes'], T) for _ in range(G)])
        elif hfn == 'linear':        # head_fn = nn.Linear through the try / except of map.py:485-489
            self.heads = nn.ModuleList([nn.Linear(L * T, cfg['num_classes']) for _ in range(G)])
        else:
            self.heads = nn.ModuleList([_NormHead(L * T, cfg['num_classes']) for _ in range(G)])
        if cfg.get('self_distill_token', True):
            self.self_dt_heads = nn.ModuleList([_NormHead(L, cfg['num_classes']) for _ in range(G)])


def _init_weights(m):
    # map_convnext.py:118-122 (timm trunc_normal_: absolute bounds)
    if isinstance(m, (nn.Conv2d, nn.Linear)):
        nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class MAP_ConvNeXt(FlatModel):
    def __init__(self, in_chans=3, num_classes=1000, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.,
                 layer_scale_init_value=1e-6, head_init_scale=1., global_pool='mmcap', last_dim=384, n_groups=4, n_tokens=3,
                 gram_group=8, ch_reduce=1, bp_dim=192, bp_groups=1, gram_layer=None, gram_dim=None, ca_dim=128, num_heads=8,
                 gram=True, split_norm=False, self_distill_token=True, head_drop=0.05, head_attn_drop=0.05, math_mode=None,
                 head_fn=None, interactive=False, **kwargs):
        """head_drop / head_attn_drop: the dropout probabilities CABlock hard-codes (map.py:149: drop=0.05 -> proj / MLP
        dropout; MAPHead attn_drop=0.05, :464) -- exposed so that parity tests can switch the (irreproducible) masks off"""
        super().__init__()
        # split_norm (map_convnext.py:97-100) selects SplitNormHead; head_fn = 'linear' (nn.Linear heads), self_distill_token = False,
        # interactive = True and gram_dim != last_dim are the MAPHead options the other reference backbones use:
        # same head code
        assert global_pool == 'mmcap' and gram and in_chans == 3, 'only the MAP head (global_pool="mmcap") with Gram tokens'
        assert gram_layer is None and bp_groups == 1 and layer_scale_init_value > 0
        depths, dims = tuple(depths), tuple(dims)
        self.num_classes = num_classes
        self.drop_path_rate = drop_path_rate
        self.cfg = dict(family='map_convnext', depths=depths, dims=dims, num_classes=num_classes, drop_path_rate=drop_path_rate,
                        last_dim=last_dim, n_groups=n_groups, n_tokens=n_tokens, gram_group=gram_group, bp_dim=bp_dim,
                        bp_groups=bp_groups, gram_dim=gram_dim or last_dim, ca_dim=ca_dim, num_heads=num_heads, mlp_ratio=4,
                        mlp_groups=2, multi_scale_level=3, naggre=0, head_drop=head_drop, head_attn_drop=head_attn_drop,
                        self_distill_token=bool(self_distill_token), head_fn=head_fn or ('split' if split_norm else 'norm'),
                        interactive=bool(interactive))
        assert self.cfg['head_fn'] in ('norm', 'split', 'linear')
        assert self.cfg['gram_dim'] % 8 == 0      # gram_dim != last_dim: the dim_mismatch CABlock (map.py:85-90,165-177)
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4), _LN(dims[0])))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(_LN(dims[i]), nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList([nn.Sequential(*[_Block(dims[i], layer_scale_init_value) for _ in range(depths[i])])
                                     for i in range(4)])
        self.norm = nn.Identity()
        self.head = _MAPHead(self.cfg, [dims[0]] + list(dims))
        self.apply(_init_weights)
        self.math_mode = math_mode

    def make_engine(self, batch, training, mode):
        from .engine_map import MAPEngine
        return MAPEngine(self, batch, training, mode)

    def grad_groups(self):
        return [('heads', ('head.',)), ('stage3', ('stages.3.', 'downsample_layers.3.')),
                ('stage2', ('stages.2.', 'downsample_layers.2.')), ('stage1', ('stages.1.', 'downsample_layers.1.'))]

    def forward(self, x, pre_logits=False):
        """eval: list of n_groups logits; train: list of [org_out, avg_out] (map.py:519-537)"""
        assert not pre_logits, 'pre_logits (MAP/validate.py --logit-extract) is not on the hot path'
        outs = super().forward(x)     # train: the engine's logits buffer is [2 * n_groups][B][NC] = org heads, then avg heads
        if not self.training or not self.cfg['self_distill_token']:
            return outs          # (without the self-distillation token MAPHead returns plain logits in both modes, map.py:536-537)
        K = self.cfg['n_groups']
        return [[outs[k], outs[K + k]] for k in range(K)]


def _create(variant, pretrained=False, **kwargs):
    kwargs.pop('pretrained_cfg', None)
    kwargs.pop('pretrained_cfg_overlay', None)
    kwargs.pop('in_22k', None)
    if pretrained:
        raise RuntimeError(f'{variant}: pretrained weights need a network fetch (map_convnext.py:206-210); load a local file with '
                           'checkpoint_path= instead')
    return MAP_ConvNeXt(**kwargs)


@register_model
def map_convnext_tiny(pretrained=False, **kwargs):
    # map_convnext.py:198-211
    return _create('map_convnext_tiny', pretrained, depths=[3, 3, 9, 3], dims=[96, 192, 384, 768], global_pool='mmcap',
                   last_dim=384, n_groups=4, n_tokens=2, gram_group=24, bp_dim=384, ca_dim=384, num_heads=12, **kwargs)


@register_model
def map_convnext_small(pretrained=False, **kwargs):
    # map_convnext.py:226-239
    return _create('map_convnext_small', pretrained, depths=[3, 3, 27, 3], dims=[96, 192, 384, 768], global_pool='mmcap',
                   last_dim=384, n_groups=4, n_tokens=3, gram_group=16, bp_dim=384, ca_dim=384, num_heads=12, **kwargs)
