"""GA-CSWin on the MI355X-native engine: the constructor arguments, `state_dict` keys / shapes and list-of-head-logits
output of the reference's GA_CSWinTransformer (/root/reference/GA/ga_cswin.py:447-693); every FLOP runs in the
hand-written HIP kernels of libgaext (engine_cswin.CSWinEngine).  The nn.Modules below only HOLD parameters and
buffers under the reference's names.

The reference registers NO factory for this family (ga_cswin.py ends at `_conv_filter`): the two names
of its `default_cfgs` / README (`ga_CSWin_64_12211_tiny_224`, `ga_CSWin_64_24322_small_224`) are bound here to the
survey's candidate hyper-parameters (41.86 M parameters vs the README's 42.0 M) -- CONFIG UNPINNED; the arithmetic of
every class is pinned by tests/golden/cswin_*.npz.
"""
import torch
import torch.nn as nn

from .flat_model import FlatModel, Holder
from .ga_convnext import BottleneckParams, ClassAttnParams
from .registry import register_model

__all__ = ['GA_CSWinTransformer']


def branch_num(reso, split, last_stage=False):
    """ga_cswin.py:155-160"""
    return 1 if (last_stage or reso == split) else 2


class CSWinBlockParams(Holder):
    """CSWinBlock parameters in registration order (ga_cswin.py:152-189)"""

    def __init__(self, dim, reso, num_heads, split_size, qkv_bias=True, last_stage=False, mlp_groups=1, mlp_ratio=4.):
        super().__init__()
        self.dim, self.reso, self.num_heads, self.split_size = dim, reso, num_heads, split_size
        self.mlp_groups = mlp_groups
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.norm1 = nn.LayerNorm(dim)
        self.branch_num = branch_num(reso, split_size, last_stage)
        self.proj = nn.Linear(dim, dim)
        bd = dim // self.branch_num
        self.attns = nn.ModuleList()
        for _ in range(self.branch_num):
            a = Holder()
            a.get_v = nn.Conv2d(bd, bd, kernel_size=3, stride=1, padding=1, groups=bd)
            self.attns.append(a)
        hid = int(dim * mlp_ratio)
        self.mlp = Holder()
        if mlp_groups == 1:
            self.mlp.fc1 = nn.Linear(dim, hid)
            self.mlp.fc2 = nn.Linear(hid, dim)
        else:
            self.mlp.fc1 = nn.Conv2d(dim, hid, kernel_size=1, groups=mlp_groups)
            self.mlp.fc2 = nn.Conv2d(hid, dim, kernel_size=1, groups=mlp_groups)
        self.norm2 = nn.LayerNorm(dim)

    def stripes(self):
        """[(H_sp, W_sp)] per branch (ga_cswin.py:71-81)"""
        r, s = self.reso, self.split_size
        return [(r, r)] if self.branch_num == 1 else [(r, s), (s, r)]


class MergeParams(Holder):
    """Merge_Block (3x3 s2, ga_cswin.py:253-257) / Merge_Block_LCF (1x1, :236-240)"""

    def __init__(self, dim, dim_out, kernel):
        super().__init__()
        self.conv = nn.Conv2d(dim, dim_out, kernel, 2 if kernel == 3 else 1, 1 if kernel == 3 else 0)
        self.norm = nn.LayerNorm(dim_out)


def _init_weights(m):
    # ga_cswin.py:598-605: only Linear / LayerNorm / BatchNorm are touched; Conv2d keeps PyTorch's default init
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class GA_CSWinTransformer(FlatModel):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=64, depth=(2, 2, 6, 2),
                 split_size=(3, 5, 7), num_heads=12, mlp_ratio=4., mlp_ratio_stage4=4., mlp_ratio_stage5=4., qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=None, use_chk=False,
                 dims=(64, 128, 256, 512), stage3_naggre=4, ga_mlp_groups=2, ga_layer_mlp_groups=1, branches=5, gram_dim=192,
                 deep_stem=True, stage5='CSWin', stage5_mlp_groups=1, ga_layer=True, math_mode=None):
        super().__init__()
        assert img_size == 224 and in_chans == 3, 'the aggregation is tied to a 14 x 14 map (ga_cswin.py:546,666-669)'
        assert deep_stem and ga_layer and norm_layer is None and qk_scale is None, 'only the configuration the hot path uses'
        assert mlp_ratio == mlp_ratio_stage4 == mlp_ratio_stage5 == 4. and drop_rate == 0. and attn_drop_rate == 0.
        assert stage5 in ('CSWin', 'bottleneck')
        depth, split_size, num_heads, dims = tuple(depth), tuple(split_size), tuple(num_heads), tuple(dims)
        assert len(depth) == 4 and len(split_size) == 5 and len(num_heads) == 5 and len(dims) == 4
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        self.drop_path_rate = drop_path_rate
        cur = dims[3]
        self.cfg = dict(family='cswin', img_size=img_size, embed_dim=embed_dim, depth=depth, split_size=split_size,
                        heads=num_heads, dims=dims, naggre=stage3_naggre, branches=branches, gram_dim=gram_dim,
                        stage5=stage5, stage5_mlp_groups=stage5_mlp_groups, ga_layer_mlp_groups=ga_layer_mlp_groups,
                        qkv_bias=qkv_bias, num_classes=num_classes, drop_path_rate=drop_path_rate,
                        # the GA head in the vocabulary of the shared head code (engine.GAEngine._build_heads):
                        gram_groups=8, gram_heads=6, dim_embed=cur // 4, num_heads=8, mlp_groups=ga_mlp_groups)
        e = embed_dim
        # deep stem: Sequential indices of the reference (Rearrange / GELU positions hold no parameters), ga_cswin.py:463-477
        idt = nn.Identity
        self.stage1_conv_embed = nn.Sequential(
            nn.Conv2d(in_chans, e, 3, stride=2, padding=1, bias=False), idt(), nn.LayerNorm(e), idt(), idt(),
            nn.Conv2d(e, e, 3, stride=1, padding=1, bias=False), idt(), nn.LayerNorm(e), idt(), idt(),
            nn.Conv2d(e, dims[0], 3, stride=2, padding=1, bias=False), idt(), nn.LayerNorm(dims[0]))
        reso = [img_size // 4, img_size // 8, img_size // 16, img_size // 32]
        self.stage1 = nn.ModuleList([CSWinBlockParams(dims[0], reso[0], num_heads[0], split_size[0], qkv_bias)
                                     for _ in range(depth[0])])
        self.merge1 = MergeParams(dims[0], dims[1], 3)
        self.stage2 = nn.ModuleList([CSWinBlockParams(dims[1], reso[1], num_heads[1], split_size[1], qkv_bias)
                                     for _ in range(depth[1])])
        self.merge2 = MergeParams(dims[1], dims[2], 3)
        self.stage3 = nn.ModuleList([CSWinBlockParams(dims[2], reso[2], num_heads[2], split_size[2], qkv_bias)
                                     for _ in range(depth[2])])
        self.merge3 = MergeParams(dims[2], dims[3], 3)
        self.stage4 = nn.ModuleList([CSWinBlockParams(dims[3], reso[3], num_heads[3], split_size[-1], qkv_bias, last_stage=True)
                                     for _ in range(depth[3])])
        aggre = sum(dims) + dims[2] * stage3_naggre
        if stage5 == 'CSWin':
            self.stage5 = nn.Sequential(idt(), MergeParams(aggre, cur, 1),
                                        CSWinBlockParams(cur, reso[2], num_heads[4], split_size[4], qkv_bias,
                                                         mlp_groups=stage5_mlp_groups), idt())
        else:
            self.stage5 = BottleneckParams(aggre, cur // 4, cur)
        self.gram_contraction = nn.ModuleList()
        self.gram_layer = nn.ModuleList()
        self.gram_embedding = nn.ModuleList()
        self.gram_expansion = nn.ModuleList()   # registered empty by the reference (ga_cswin.py:551)
        self.ga = nn.ModuleList()
        self.fc = nn.ModuleList()
        ntri = (gram_dim + 1) * gram_dim // 2
        for _ in range(branches):
            self.gram_contraction.append(nn.Sequential(nn.Conv2d(cur, gram_dim, kernel_size=1, groups=8), nn.BatchNorm2d(gram_dim)))
            self.gram_layer.append(nn.Sequential(idt(), CSWinBlockParams(gram_dim, reso[2], 6, split_size[4], qkv_bias,
                                                                         mlp_groups=ga_layer_mlp_groups), idt()))
            self.gram_embedding.append(nn.Sequential(nn.Conv2d(ntri, cur, kernel_size=1, groups=8), nn.BatchNorm2d(cur)))
            self.ga.append(ClassAttnParams(cur, cur // 4, mlp_groups=ga_mlp_groups))
            self.fc.append(nn.Linear(cur, num_classes))
        self.apply(_init_weights)
        self.math_mode = math_mode

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}   # ga_cswin.py:607-609 (neither exists in this model)

    def make_engine(self, batch, training, mode):
        from .engine_cswin import CSWinEngine
        return CSWinEngine(self, batch, training, mode)

    def grad_groups(self):
        return [('heads', ('stage5.', 'gram_contraction.', 'gram_layer.', 'gram_embedding.', 'ga.', 'fc.')),
                ('stage3', ('stage4.', 'merge3.')), ('stage2', ('stage3.', 'merge2.')), ('stage1', ('stage2.', 'merge1.'))]


def _create(variant, pretrained=False, **kwargs):
    for k in ('pretrained_cfg', 'pretrained_cfg_overlay', 'features_only', 'default_cfg'):
        kwargs.pop(k, None)
    if pretrained:
        raise RuntimeError(f'{variant}: the reference publishes no pretrained weights (url is empty, ga_cswin.py:34-37)')
    return GA_CSWinTransformer(**kwargs)


# CONFIG UNPINNED: original CSWin-T / -S trunk hyper-parameters + the GA head defaults of the class
@register_model
def ga_CSWin_64_12211_tiny_224(pretrained=False, **kwargs):
    return _create('ga_CSWin_64_12211_tiny_224', pretrained, embed_dim=64, depth=[1, 2, 21, 1], split_size=[1, 2, 7, 7, 7],
                   num_heads=[2, 4, 8, 16, 16], dims=[64, 128, 256, 512], stage3_naggre=4, stage5_mlp_groups=4, **kwargs)


@register_model
def ga_CSWin_64_24322_small_224(pretrained=False, **kwargs):
    return _create('ga_CSWin_64_24322_small_224', pretrained, embed_dim=64, depth=[2, 4, 32, 2], split_size=[1, 2, 7, 7, 7],
                   num_heads=[2, 4, 8, 16, 16], dims=[64, 128, 256, 512], stage3_naggre=4, stage5_mlp_groups=4, **kwargs)
