"""ORACLE (test infrastructure only -- never imported by the product package).

CPU restatement, in plain PyTorch ops over a flat ``state_dict``, of the reference's GA-CSWin hot path:

* model ................. /root/reference/GA/ga_cswin.py:59-693 (LePEAttention, CSWinBlock, Merge_Block(_LCF),
                          ClassAttn(expansion 4), GroupConvMlp, LayerScaleBlockClassAttn, GA_CSWinTransformer)
* loss / metric / step .. shared with GA-ConvNeXt: oracle/ga_convnext_oracle.py (GA/train.py:735-745,848-860)

Backward is torch autograd over this restated forward.  The restatement is *pinned* by tests/golden/cswin_*.npz,
which oracle/gen_golden_cswin.py produced in the build container by importing the real reference classes (against
oracle/timm_stub; einops is installed) -- see tests/test_oracle_golden.py.

The reference registers NO GA-CSWin factory: the *configuration* of "GA-CSWin-Tiny" used by the
benchmarks (`TINY`, below) is the survey's candidate -- labelled "config unpinned"; the arithmetic of every class is
pinned by the fixtures.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .ga_convnext_oracle import (_bn, _dp, accuracy, bottleneck, channel_shuffle, ga_loss, gen_input, grad_errors,  # noqa: F401
                                 gram_index, is_buffer, se_rd_channels, topk_indices, validate_output)

# SURVEY.md F3 candidate for `ga_CSWin_64_12211_tiny_224` (41.86 M parameters vs README 42.0 M): config unpinned
TINY = dict(embed_dim=64, depth=(1, 2, 21, 1), split_size=(1, 2, 7, 7, 7), num_heads=(2, 4, 8, 16, 16),
            dims=(64, 128, 256, 512), naggre=4, stage5_mlp_groups=4)
# same family, the README's `ga_CSWin_64_24322_small_224` (orig. CSWin-S depths); config unpinned
SMALL = dict(embed_dim=64, depth=(2, 4, 32, 2), split_size=(1, 2, 7, 7, 7), num_heads=(2, 4, 8, 16, 16),
             dims=(64, 128, 256, 512), naggre=4, stage5_mlp_groups=4)
VARIANTS = {'ga_CSWin_64_12211_tiny_224': TINY, 'ga_CSWin_64_24322_small_224': SMALL}


def make_cfg(name=None, **over):
    """constructor defaults of GA_CSWinTransformer (ga_cswin.py:450-453)"""
    cfg = dict(img_size=224, in_chans=3, num_classes=1000, embed_dim=64, depth=(2, 2, 6, 2), split_size=(3, 5, 7),
               num_heads=12, mlp_ratio=4.0, qkv_bias=True, dims=(64, 128, 256, 512), naggre=4, ga_mlp_groups=2,
               ga_layer_mlp_groups=1, branches=5, gram_dim=192, stage5='CSWin', stage5_mlp_groups=1,
               gram_heads=6, ga_heads=8, ga_expansion=4, gram_groups=8, drop_path_rate=0.0)
    if name is not None:
        cfg.update(VARIANTS[name])
    cfg.update(over)
    for k in ('depth', 'split_size', 'num_heads', 'dims'):
        cfg[k] = tuple(cfg[k])
    return cfg


def tap_after(nblocks, naggre):
    """ga_cswin.py:659 -- 1-based block counts of stage3 after which a tap is taken (at most `naggre`)."""
    step = nblocks // (naggre + 1)
    taps = []
    for b in range(1, nblocks + 1):
        if b % step == 0 and len(taps) < naggre:
            taps.append(b)
    return taps


def drop_path_rates(cfg):
    """ga_cswin.py:486 -- linspace over sum(depth) of the 4 stages; stage5 and every gram layer take dpr[-1] (:538,571)"""
    return [x.item() for x in torch.linspace(0, cfg['drop_path_rate'], int(np.sum(cfg['depth'])))]


# --------------------------------------------------------------------------------------
# state_dict layout (names + shapes identical to the reference nn.Module's state_dict)
# --------------------------------------------------------------------------------------
def _ln_shapes(pre, c, o):
    o[pre + 'weight'] = (c,)
    o[pre + 'bias'] = (c,)


def _bn_shapes(pre, c, o):
    o[pre + 'weight'] = (c,)
    o[pre + 'bias'] = (c,)
    o[pre + 'running_mean'] = (c,)
    o[pre + 'running_var'] = (c,)
    o[pre + 'num_batches_tracked'] = ()


def branch_num(reso, split, last_stage=False):
    """ga_cswin.py:155-160"""
    return 1 if (last_stage or reso == split) else 2


def _cswin_block_shapes(pre, dim, nbranch, qkv_bias, mlp_groups, mlp_ratio, o):
    """registration order of CSWinBlock.__init__ (ga_cswin.py:152-189)"""
    o[pre + 'qkv.weight'] = (3 * dim, dim)
    if qkv_bias:
        o[pre + 'qkv.bias'] = (3 * dim,)
    _ln_shapes(pre + 'norm1.', dim, o)
    o[pre + 'proj.weight'] = (dim, dim)
    o[pre + 'proj.bias'] = (dim,)
    bd = dim if nbranch == 1 else dim // 2
    for i in range(nbranch):
        o[pre + f'attns.{i}.get_v.weight'] = (bd, 1, 3, 3)
        o[pre + f'attns.{i}.get_v.bias'] = (bd,)
    hid = int(dim * mlp_ratio)
    if mlp_groups == 1:
        o[pre + 'mlp.fc1.weight'] = (hid, dim)
        o[pre + 'mlp.fc1.bias'] = (hid,)
        o[pre + 'mlp.fc2.weight'] = (dim, hid)
        o[pre + 'mlp.fc2.bias'] = (dim,)
    else:
        o[pre + 'mlp.fc1.weight'] = (hid, dim // mlp_groups, 1, 1)
        o[pre + 'mlp.fc1.bias'] = (hid,)
        o[pre + 'mlp.fc2.weight'] = (dim, hid // mlp_groups, 1, 1)
        o[pre + 'mlp.fc2.bias'] = (dim,)
    _ln_shapes(pre + 'norm2.', dim, o)


def stage_layout(cfg):
    """[(prefix, dim, reso, split, heads, last_stage)] of the four trunk stages (ga_cswin.py:487-526)"""
    img, d, sp, nh = cfg['img_size'], cfg['dims'], cfg['split_size'], cfg['num_heads']
    return [('stage1.', d[0], img // 4, sp[0], nh[0], False), ('stage2.', d[1], img // 8, sp[1], nh[1], False),
            ('stage3.', d[2], img // 16, sp[2], nh[2], False), ('stage4.', d[3], img // 32, sp[-1], nh[3], True)]


def state_shapes(cfg):
    d, e, dep = cfg['dims'], cfg['embed_dim'], cfg['depth']
    o = OrderedDict()
    # deep stem (ga_cswin.py:463-477)
    o['stage1_conv_embed.0.weight'] = (e, cfg['in_chans'], 3, 3)
    _ln_shapes('stage1_conv_embed.2.', e, o)
    o['stage1_conv_embed.5.weight'] = (e, e, 3, 3)
    _ln_shapes('stage1_conv_embed.7.', e, o)
    o['stage1_conv_embed.10.weight'] = (d[0], e, 3, 3)
    _ln_shapes('stage1_conv_embed.12.', d[0], o)
    st = stage_layout(cfg)
    for si, (pre, dim, reso, split, heads, last) in enumerate(st):
        if si > 0:
            mp = f'merge{si}.'
            o[mp + 'conv.weight'] = (dim, d[si - 1], 3, 3)
            o[mp + 'conv.bias'] = (dim,)
            _ln_shapes(mp + 'norm.', dim, o)
            # the reference registers merge_i BEFORE stage_{i+1} (ga_cswin.py:495-526)
        for j in range(dep[si]):
            _cswin_block_shapes(f'{pre}{j}.', dim, branch_num(reso, split, last), cfg['qkv_bias'], 1, cfg['mlp_ratio'], o)
    cur = d[3]
    aggre = sum(d) + d[2] * cfg['naggre']
    r14 = cfg['img_size'] // 16
    if cfg['stage5'] == 'CSWin':
        o['stage5.1.conv.weight'] = (cur, aggre, 1, 1)
        o['stage5.1.conv.bias'] = (cur,)
        _ln_shapes('stage5.1.norm.', cur, o)
        _cswin_block_shapes('stage5.2.', cur, branch_num(r14, cfg['split_size'][4]), cfg['qkv_bias'],
                            cfg['stage5_mlp_groups'], cfg['mlp_ratio'], o)
    else:  # Bottleneck (ga_cswin.py:541): registration order downsample, conv1, bn1, conv2, bn2, se, conv3, bn3
        w = cur // 4
        o['stage5.downsample.0.weight'] = (cur, aggre, 1, 1)
        o['stage5.downsample.0.bias'] = (cur,)
        _bn_shapes('stage5.downsample.1.', cur, o)
        o['stage5.conv1.weight'] = (w, aggre, 1, 1)
        _bn_shapes('stage5.bn1.', w, o)
        o['stage5.conv2.weight'] = (w, w, 3, 3)
        _bn_shapes('stage5.bn2.', w, o)
        rd = se_rd_channels(w)
        o['stage5.se.fc1.weight'] = (rd, w, 1, 1)
        o['stage5.se.fc1.bias'] = (rd,)
        o['stage5.se.fc2.weight'] = (w, rd, 1, 1)
        o['stage5.se.fc2.bias'] = (w,)
        o['stage5.conv3.weight'] = (cur, w, 1, 1)
        _bn_shapes('stage5.bn3.', cur, o)
    g, nb, gg = cfg['gram_dim'], cfg['branches'], cfg['gram_groups']
    ntri = (g + 1) * g // 2
    for k in range(nb):
        o[f'gram_contraction.{k}.0.weight'] = (g, cur // gg, 1, 1)
        o[f'gram_contraction.{k}.0.bias'] = (g,)
        _bn_shapes(f'gram_contraction.{k}.1.', g, o)
    for k in range(nb):
        _cswin_block_shapes(f'gram_layer.{k}.1.', g, branch_num(r14, cfg['split_size'][4]), cfg['qkv_bias'],
                            cfg['ga_layer_mlp_groups'], 4.0, o)
    for k in range(nb):
        o[f'gram_embedding.{k}.0.weight'] = (cur, ntri // gg, 1, 1)
        o[f'gram_embedding.{k}.0.bias'] = (cur,)
        _bn_shapes(f'gram_embedding.{k}.1.', cur, o)
    mg, ex = cfg['ga_mlp_groups'], cfg['ga_expansion']
    for k in range(nb):
        pre = f'ga.{k}.'
        o[pre + 'gamma_1'] = (cur,)
        o[pre + 'gamma_2'] = (cur,)
        _ln_shapes(pre + 'norm1.', cur, o)
        o[pre + 'attn.q.weight'] = (cur // ex, cur)
        o[pre + 'attn.k.weight'] = (cur // ex, cur)
        o[pre + 'attn.v.weight'] = (cur // ex, cur)
        o[pre + 'attn.proj.weight'] = (cur, cur // ex)
        o[pre + 'attn.proj.bias'] = (cur,)
        _ln_shapes(pre + 'norm2.', cur, o)
        o[pre + 'mlp.fc1.weight'] = (4 * cur, cur // mg, 1, 1)
        o[pre + 'mlp.fc1.bias'] = (4 * cur,)
        o[pre + 'mlp.fc2.weight'] = (cur, 4 * cur // mg, 1, 1)
        o[pre + 'mlp.fc2.bias'] = (cur,)
    for k in range(nb):
        o[f'fc.{k}.weight'] = (cfg['num_classes'], cur)
        o[f'fc.{k}.bias'] = (cfg['num_classes'],)
    return o


def fill_state(cfg, seed=0, dtype=torch.float32):
    """deterministic name-hashed fill, O(1) activations (same rule as ga_convnext_oracle.fill_state)"""
    sd = OrderedDict()
    for name, shape in state_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        if leaf == 'running_mean':
            v = rs.uniform(-0.1, 0.1, shape)
        elif leaf == 'running_var':
            v = rs.uniform(0.5, 1.5, shape)
        elif leaf in ('gamma_1', 'gamma_2'):
            v = rs.uniform(0.4, 0.9, shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(shape) * (1.0 / math.sqrt(fan_in))
        elif leaf == 'weight':
            v = rs.uniform(0.8, 1.2, shape)
        else:
            v = rs.uniform(-0.1, 0.1, shape)
        sd[name] = torch.tensor(v, dtype=dtype)
    return sd


# --------------------------------------------------------------------------------------
# forward restatement
# --------------------------------------------------------------------------------------
def img2windows(img, hs, ws):
    """ga_cswin.py:215-222: (B,C,H,W) -> (B*nwin, hs*ws, C), windows in row-major order"""
    b, c, h, w = img.shape
    t = img.view(b, c, h // hs, hs, w // ws, ws)
    return t.permute(0, 2, 4, 3, 5, 1).contiguous().reshape(-1, hs * ws, c)


def windows2img(t, hs, ws, h, w):
    """ga_cswin.py:225-233"""
    b = int(t.shape[0] / (h * w / hs / ws))
    img = t.view(b, h // hs, w // ws, hs, ws, -1)
    return img.permute(0, 1, 3, 2, 4, 5).contiguous().view(b, h, w, -1)


def stripe_shape(reso, idx, split):
    """(H_sp, W_sp) of LePEAttention (ga_cswin.py:71-81)"""
    if idx == -1:
        return reso, reso
    if idx == 0:
        return reso, split
    return split, reso


def lepe_attention(q, k, v, w_v, b_v, reso, idx, split, heads):
    """LePEAttention.forward (ga_cswin.py:110-136) on (B, L, C) q / k / v of ONE branch"""
    b, l, c = q.shape
    hs, ws = stripe_shape(reso, idx, split)
    hd = c // heads
    scale = hd ** -0.5

    def im2cswin(t):
        t = t.transpose(-2, -1).contiguous().view(b, c, reso, reso)
        t = img2windows(t, hs, ws)
        return t.reshape(-1, hs * ws, heads, hd).permute(0, 2, 1, 3).contiguous()

    qw, kw = im2cswin(q), im2cswin(k)
    # get_lepe (:95-108): depthwise 3x3 of v INSIDE each window (zero padding at the window border)
    vi = v.transpose(-2, -1).contiguous().view(b, c, reso // hs, hs, reso // ws, ws)
    vi = vi.permute(0, 2, 4, 1, 3, 5).contiguous().reshape(-1, c, hs, ws)
    lepe = F.conv2d(vi, w_v, b_v, stride=1, padding=1, groups=c)
    lepe = lepe.reshape(-1, heads, hd, hs * ws).permute(0, 1, 3, 2).contiguous()
    vw = vi.reshape(-1, heads, hd, hs * ws).permute(0, 1, 3, 2).contiguous()
    attn = (qw * scale) @ kw.transpose(-2, -1)
    attn = F.softmax(attn, dim=-1, dtype=attn.dtype)
    x = attn @ vw + lepe
    x = x.transpose(1, 2).reshape(-1, hs * ws, c)
    return windows2img(x, hs, ws, reso, reso).view(b, -1, c)


def group_conv_mlp(sd, pre, x, groups):
    """GroupConvMlp.forward on (B, L, C) tokens (ga_cswin.py:338-349)"""
    t = x.permute(0, 2, 1).unsqueeze(-1)
    t = F.conv2d(t, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'], groups=groups)
    t = F.gelu(t)
    t = channel_shuffle(t, groups)
    t = F.conv2d(t, sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'], groups=groups)
    return t.squeeze(-1).permute(0, 2, 1)


def cswin_block(sd, pre, x, reso, split, heads, last_stage=False, mlp_groups=1, dp_mask=None):
    """CSWinBlock.forward (ga_cswin.py:191-212); LayerNorm eps = nn.LayerNorm default 1e-5.
    dp_mask: None, one per-sample mask for both DropPath calls, or a pair (attention branch, MLP branch) -- the module
    draws a fresh mask at each of its two drop_path calls (:209-210)."""
    dp1, dp2 = dp_mask if isinstance(dp_mask, (tuple, list)) else (dp_mask, dp_mask)
    b, l, c = x.shape
    nbr = branch_num(reso, split, last_stage)
    img = F.layer_norm(x, (c,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-5)
    qkv = F.linear(img, sd[pre + 'qkv.weight'], sd.get(pre + 'qkv.bias')).reshape(b, -1, 3, c).permute(2, 0, 1, 3)
    if nbr == 2:
        h = c // 2
        x1 = lepe_attention(qkv[0][:, :, :h], qkv[1][:, :, :h], qkv[2][:, :, :h], sd[pre + 'attns.0.get_v.weight'],
                            sd[pre + 'attns.0.get_v.bias'], reso, 0, split, heads // 2)
        x2 = lepe_attention(qkv[0][:, :, h:], qkv[1][:, :, h:], qkv[2][:, :, h:], sd[pre + 'attns.1.get_v.weight'],
                            sd[pre + 'attns.1.get_v.bias'], reso, 1, split, heads // 2)
        att = torch.cat([x1, x2], dim=2)
    else:
        att = lepe_attention(qkv[0], qkv[1], qkv[2], sd[pre + 'attns.0.get_v.weight'], sd[pre + 'attns.0.get_v.bias'],
                             reso, -1, split, heads)
    att = F.linear(att, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])
    x = x + _dp(att, dp1)
    t = F.layer_norm(x, (c,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-5)
    if mlp_groups == 1:
        t = F.linear(F.gelu(F.linear(t, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'])),
                     sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias'])
    else:
        t = group_conv_mlp(sd, pre + 'mlp.', t, mlp_groups)
    return x + _dp(t, dp2)


def _tok(x):
    """'b c h w -> b (h w) c'"""
    b, c = x.shape[:2]
    return x.view(b, c, -1).transpose(-2, -1).contiguous()


def _img(x):
    """(B, L, C) -> (B, C, sqrt L, sqrt L) as the reference does it (ga_cswin.py:646)"""
    b, n, c = x.shape
    r = int(n ** 0.5)
    return x.transpose(-2, -1).reshape(b, c, r, r)


def conv_embed(sd, x, cfg):
    """stage1_conv_embed, deep stem (ga_cswin.py:463-477)"""
    p = 'stage1_conv_embed.'
    e = cfg['embed_dim']
    x = F.conv2d(x, sd[p + '0.weight'], None, stride=2, padding=1)
    hw = x.shape[2:]
    x = F.layer_norm(_tok(x), (e,), sd[p + '2.weight'], sd[p + '2.bias'], 1e-5)
    x = F.gelu(x.transpose(-2, -1).reshape(x.shape[0], e, *hw))
    x = F.conv2d(x, sd[p + '5.weight'], None, stride=1, padding=1)
    x = F.layer_norm(_tok(x), (e,), sd[p + '7.weight'], sd[p + '7.bias'], 1e-5)
    x = F.gelu(x.transpose(-2, -1).reshape(x.shape[0], e, *hw))
    x = F.conv2d(x, sd[p + '10.weight'], None, stride=2, padding=1)
    return F.layer_norm(_tok(x), (cfg['dims'][0],), sd[p + '12.weight'], sd[p + '12.bias'], 1e-5)


def merge_block(sd, pre, x, stride):
    """Merge_Block (3x3 s2 p1, ga_cswin.py:259-268) / Merge_Block_LCF (1x1, :242-251) on (B, L, C) tokens"""
    w = sd[pre + 'conv.weight']
    y = F.conv2d(_img(x), w, sd[pre + 'conv.bias'], stride=stride, padding=1 if w.shape[-1] == 3 else 0)
    return F.layer_norm(_tok(y), (w.shape[0],), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)


def forward_features(sd, x, cfg, training=False, new_stats=None, dp_masks=None):
    """GA_CSWinTransformer.forward_features (ga_cswin.py:636-671) -> (B, C, 14, 14)"""
    dp_masks = dp_masks or {}
    dep = cfg['depth']
    x = conv_embed(sd, x, cfg)
    xs = []
    st = stage_layout(cfg)
    taps = tap_after(dep[2], cfg['naggre'])
    for si, (pre, dim, reso, split, heads, last) in enumerate(st):
        if si > 0:
            x = merge_block(sd, f'merge{si}.', x, 2)
        for j in range(dep[si]):
            bp = f'{pre}{j}.'
            x = cswin_block(sd, bp, x, reso, split, heads, last, 1, dp_masks.get(bp))
            if si == 2 and (j + 1) in taps:
                xs.append(_img(x))
        xs.append(_img(x))
    cat = torch.cat([F.adaptive_avg_pool2d(xs[0], 14), F.adaptive_avg_pool2d(xs[1], 14)] + xs[2:-1] +
                    [F.interpolate(xs[-1], scale_factor=2, mode='bilinear')], dim=1)
    if cfg['stage5'] == 'CSWin':
        t = merge_block(sd, 'stage5.1.', _tok(cat), 1)
        r14 = cfg['img_size'] // 16
        t = cswin_block(sd, 'stage5.2.', t, r14, cfg['split_size'][4], cfg['num_heads'][4], False,
                        cfg['stage5_mlp_groups'], dp_masks.get('stage5.2.'))
        b, n, c = t.shape
        return t.transpose(-2, -1).reshape(b, c, r14, r14)
    return bottleneck(sd, 'stage5.', cat, training, new_stats, dp_masks.get('stage5.'))


def get_gram(x):
    """GA_CSWinTransformer.get_gram (ga_cswin.py:624-634): no float64 branch here"""
    b, c, h, w = x.shape
    x = (x / h).reshape(b, c, h * w)
    g = torch.bmm(x, x.transpose(1, 2)) / (h * w)
    g = g.reshape(b, c * c)[:, gram_index(c)]
    g = F.normalize(g)
    g = g if g.dtype == torch.float64 else g.float()
    return g.reshape(b, -1, 1, 1)


def class_attn_block(sd, pre, x_tok, x_cls, cfg, dp_mask=None):
    """LayerScaleBlockClassAttn + ClassAttn(expansion) + GroupConvMlp (ga_cswin.py:271-375)"""
    c = x_tok.shape[2]
    nh, ex = cfg['ga_heads'], cfg['ga_expansion']
    e = c // ex
    hd = (c // nh) // ex
    u = torch.cat((x_cls, x_tok), dim=1)
    un = F.layer_norm(u, (c,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-5)
    b, n, _ = un.shape
    q = F.linear(un[:, 0], sd[pre + 'attn.q.weight']).unsqueeze(1).reshape(b, 1, nh, e // nh).permute(0, 2, 1, 3)
    k = F.linear(un, sd[pre + 'attn.k.weight']).reshape(b, n, nh, e // nh).permute(0, 2, 1, 3)
    v = F.linear(un, sd[pre + 'attn.v.weight']).reshape(b, n, nh, e // nh).permute(0, 2, 1, 3)
    attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(b, 1, e)
    a = F.linear(a, sd[pre + 'attn.proj.weight'], sd[pre + 'attn.proj.bias'])
    x_cls = x_cls + _dp(sd[pre + 'gamma_1'] * a, dp_mask)
    t = F.layer_norm(x_cls, (c,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-5)
    t = group_conv_mlp(sd, pre + 'mlp.', t, cfg['ga_mlp_groups'])
    return x_cls + _dp(sd[pre + 'gamma_2'] * t, dp_mask)


def forward(sd, x, cfg, training=False, new_stats=None, dp_masks=None):
    """GA_CSWinTransformer.forward (ga_cswin.py:674-693): list of per-head logits"""
    dp_masks = dp_masks or {}
    x = forward_features(sd, x, cfg, training, new_stats, dp_masks)
    b, c = x.shape[:2]
    r14 = cfg['img_size'] // 16
    tok = x.view(b, c, -1).permute(0, 2, 1)
    outs = []
    for k in range(cfg['branches']):
        g = F.conv2d(x, sd[f'gram_contraction.{k}.0.weight'], sd[f'gram_contraction.{k}.0.bias'], groups=cfg['gram_groups'])
        g = _bn(sd, f'gram_contraction.{k}.1.', g, training, new_stats)
        bp = f'gram_layer.{k}.1.'
        t = cswin_block(sd, bp, _tok(g), r14, cfg['split_size'][4], cfg['gram_heads'], False, cfg['ga_layer_mlp_groups'],
                        dp_masks.get(bp))
        g = t.transpose(-2, -1).reshape(b, g.shape[1], r14, r14)
        g = get_gram(g)
        g = F.conv2d(g, sd[f'gram_embedding.{k}.0.weight'], sd[f'gram_embedding.{k}.0.bias'], groups=cfg['gram_groups'])
        g = _bn(sd, f'gram_embedding.{k}.1.', g, training, new_stats)
        cls = g.view(b, c, -1).permute(0, 2, 1)
        cls = class_attn_block(sd, f'ga.{k}.', tok, cls, cfg, dp_masks.get(f'ga.{k}.'))
        outs.append(F.linear(cls.reshape(b, -1), sd[f'fc.{k}.weight'], sd[f'fc.{k}.bias']))
    return outs


def no_weight_decay(name, shape):
    """timm rule + GA_CSWinTransformer.no_weight_decay() = {'pos_embed','cls_token'} (ga_cswin.py:607-609; neither exists)"""
    return len(shape) <= 1 or name.endswith('.bias')


def train_step_grads(sd, x, target, cfg, lam=-0.8, kind='ce', smoothing=0.0, dp_masks=None):
    """One training forward+backward of the restated path. Returns (loss, outputs, grads, new_bn_stats)."""
    names = [n for n in sd if not is_buffer(n)]
    leaf = OrderedDict((n, (sd[n].detach().clone().requires_grad_(True) if not is_buffer(n) else sd[n])) for n in sd)
    new_stats = {}
    outs = forward(leaf, x, cfg, training=True, new_stats=new_stats, dp_masks=dp_masks)
    loss = ga_loss(outs, target, lam, kind, smoothing)
    gs = torch.autograd.grad(loss, [leaf[n] for n in names])
    return loss.detach(), [o.detach() for o in outs], OrderedDict(zip(names, gs)), new_stats
