"""ORACLE (test infrastructure only -- never imported by the product package).

CPU restatement, in plain PyTorch ops over a flat ``state_dict``, of the reference's MAP hot path:

* head library .......... /root/reference/MAP/models/map.py:43-539 (GroupConvMlp, ClassAttention incl. `interactive`,
                          CABlock, GramToken, CAP, MultiScale, MAP, NormHead, MAPHead)
* ConvNeXt trunk ........ /root/reference/MAP/models/map_convnext.py:14-170
* loss .................. /root/reference/MAP/train.py:792-839 (multi_group_loss, distill_tokens == 0 branch)
* validate reduction .... /root/reference/MAP/train.py:1000-1006 (MEAN of the group logits; GA sums)

Backward is torch autograd over this restated forward.  Pinned by tests/golden/map_*.npz, which oracle/gen_golden_map.py
produced in the build container from the real reference classes (map.py imports with torch alone; map_convnext.py needs
oracle/timm_stub) -- with every nn.Dropout of the head set to p = 0 (CABlock hard-codes drop = attn_drop = 0.05,
map.py:149,464: the dropout masks are not reproducible across implementations; this oracle takes explicit masks instead).
Known answers: parameter counts 47,833,760 (map_convnext_tiny) and 82,837,664 (map_convnext_small), MAP/README.MD:308,373.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .ga_convnext_oracle import _bn, _dp, channel_shuffle, gen_input, grad_errors, head_loss, is_buffer, topk_indices  # noqa: F401

VARIANTS = {
    # map_convnext.py:201-205, 229-233
    'map_convnext_tiny': dict(depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), last_dim=384, n_groups=4, n_tokens=2, gram_group=24,
                              bp_dim=384, ca_dim=384, num_heads=12),
    'map_convnext_small': dict(depths=(3, 3, 27, 3), dims=(96, 192, 384, 768), last_dim=384, n_groups=4, n_tokens=3, gram_group=16,
                               bp_dim=384, ca_dim=384, num_heads=12),
}


def make_cfg(name=None, **over):
    """ConvNeXt(...) defaults of map_convnext.py:58-66 + the MAPHead arguments it hard-codes (:102-109)"""
    cfg = dict(in_chans=3, num_classes=1000, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.0, last_dim=384,
               n_groups=4, n_tokens=3, gram_group=8, bp_dim=192, bp_groups=1, gram_dim=None, ca_dim=128, num_heads=8,
               self_distill_token=True, multi_scale_level=3, mlp_ratio=4, mlp_groups=2, interactive=False,
               head_fn='norm')       # 'norm' NormHead | 'split' SplitNormHead (map_convnext split_norm=True) | 'linear' nn.Linear
    if name is not None:
        cfg.update(VARIANTS[name])
    cfg.update(over)
    cfg['depths'], cfg['dims'] = tuple(cfg['depths']), tuple(cfg['dims'])
    if cfg['gram_dim'] is None:
        cfg['gram_dim'] = cfg['last_dim']
    return cfg


def drop_path_rates(cfg):
    """map_convnext.py:85 -- linspace over sum(depths) of the 4 stages"""
    pts = torch.linspace(0, cfg['drop_path_rate'], sum(cfg['depths'])).split(list(cfg['depths']))
    return [p.tolist() for p in pts]


# --------------------------------------------------------------------------------------
# state_dict layout (registration order of the reference modules)
# --------------------------------------------------------------------------------------
def _bn_shapes(pre, c, o):
    o[pre + 'weight'] = (c,)
    o[pre + 'bias'] = (c,)
    o[pre + 'running_mean'] = (c,)
    o[pre + 'running_var'] = (c,)
    o[pre + 'num_batches_tracked'] = ()


def _ln_shapes(pre, c, o):
    o[pre + 'weight'] = (c,)
    o[pre + 'bias'] = (c,)


def head_shapes(pre, cfg, channels, o):
    """MAPHead (map.py:462-492): mmcap (MAP: mmcap list of CAP, then multi_scale), heads, self_dt_heads"""
    L, G, T = cfg['last_dim'], cfg['n_groups'], cfg['n_tokens']
    gd, bp, E, nh = cfg['gram_dim'], cfg['bp_dim'], cfg['ca_dim'], cfg['num_heads']
    mm = gd != L          # CABlock / ClassAttention dim_mismatch (map.py:77,153): separate projections and norms for class and image rows
    mg = cfg['mlp_groups']
    hid = int(L * cfg['mlp_ratio'])
    for i in range(G):
        cp = f'{pre}mmcap.mmcap.{i}.'
        ap = cp + 'attention.0.'
        # CABlock registration order (map.py:154-169): norm2, attn (proj, q, k, v [, w1, w2]), mlp, norm1
        _ln_shapes(ap + 'norm2.', L, o)
        o[ap + 'attn.proj.weight'] = (L, E)
        o[ap + 'attn.proj.bias'] = (L,)
        for n, cin in ((('q', gd), ('k1', gd), ('v1', gd), ('k2', L), ('v2', L)) if mm else (('q', L), ('k', L), ('v', L))):
            o[ap + f'attn.{n}.weight'] = (E, cin)
            o[ap + f'attn.{n}.bias'] = (E,)
        if cfg['interactive']:
            for n in ('w1', 'w2'):
                o[ap + f'attn.{n}.weight'] = (nh, nh)
                o[ap + f'attn.{n}.bias'] = (nh,)
        o[ap + 'mlp.fc1.weight'] = (hid, L // mg, 1, 1)
        o[ap + 'mlp.fc1.bias'] = (hid,)
        o[ap + 'mlp.fc2.weight'] = (L, hid // mg, 1, 1)
        o[ap + 'mlp.fc2.bias'] = (L,)
        if mm:
            _ln_shapes(ap + 'norm1_1.', gd, o)
            _ln_shapes(ap + 'norm1_2.', L, o)
        else:
            _ln_shapes(ap + 'norm1.', L, o)
        gp = cp + 'gram_token_extraction.'
        o[gp + 'bp_index'] = (bp * (bp + 1) // 2,)
        o[gp + 'ch_reduction.0.weight'] = (bp, L // cfg['bp_groups'], 1, 1)
        _bn_shapes(gp + 'ch_reduction.1.', bp, o)
        o[gp + 'bp_reduction.0.weight'] = (gd * T, bp * (bp + 1) // 2 // cfg['gram_group'], 1, 1)
        _bn_shapes(gp + 'bp_reduction.1.', gd * T, o)
    mp = f'{pre}mmcap.multi_scale.concat_conv.'
    o[mp + '0.weight'] = (L, sum(channels), 1, 1)
    _bn_shapes(mp + '1.', L, o)
    hfn = cfg.get('head_fn', 'norm')
    for i in range(G):
        if hfn == 'split':           # SplitNormHead (map.py:415-425): ModuleList norm, then ModuleList head
            for t in range(T):
                _ln_shapes(f'{pre}heads.{i}.norm.{t}.', L, o)
            for t in range(T):
                o[f'{pre}heads.{i}.head.{t}.weight'] = (cfg['num_classes'], L)
                o[f'{pre}heads.{i}.head.{t}.bias'] = (cfg['num_classes'],)
        elif hfn == 'linear':        # head_fn = nn.Linear (map.py:485-489)
            o[f'{pre}heads.{i}.weight'] = (cfg['num_classes'], L * T)
            o[f'{pre}heads.{i}.bias'] = (cfg['num_classes'],)
        else:
            _ln_shapes(f'{pre}heads.{i}.norm.', L * T, o)
            o[f'{pre}heads.{i}.head.weight'] = (cfg['num_classes'], L * T)
            o[f'{pre}heads.{i}.head.bias'] = (cfg['num_classes'],)
    if cfg['self_distill_token']:
        for i in range(G):
            _ln_shapes(f'{pre}self_dt_heads.{i}.norm.', L, o)
            o[f'{pre}self_dt_heads.{i}.head.weight'] = (cfg['num_classes'], L)
            o[f'{pre}self_dt_heads.{i}.head.bias'] = (cfg['num_classes'],)


def state_shapes(cfg):
    d, dep = cfg['dims'], cfg['depths']
    o = OrderedDict()
    o['downsample_layers.0.0.weight'] = (d[0], cfg['in_chans'], 4, 4)
    o['downsample_layers.0.0.bias'] = (d[0],)
    _ln_shapes('downsample_layers.0.1.', d[0], o)
    for i in range(3):
        _ln_shapes(f'downsample_layers.{i + 1}.0.', d[i], o)
        o[f'downsample_layers.{i + 1}.1.weight'] = (d[i + 1], d[i], 2, 2)
        o[f'downsample_layers.{i + 1}.1.bias'] = (d[i + 1],)
    for i in range(4):
        for j in range(dep[i]):
            p = f'stages.{i}.{j}.'
            o[p + 'gamma'] = (d[i],)
            o[p + 'dwconv.weight'] = (d[i], 1, 7, 7)
            o[p + 'dwconv.bias'] = (d[i],)
            _ln_shapes(p + 'norm.', d[i], o)
            o[p + 'pwconv1.weight'] = (4 * d[i], d[i])
            o[p + 'pwconv1.bias'] = (4 * d[i],)
            o[p + 'pwconv2.weight'] = (d[i], 4 * d[i])
            o[p + 'pwconv2.bias'] = (d[i],)
    head_shapes('head.', cfg, [d[0]] + list(d), o)
    return o


def is_index_buffer(name):
    return name.endswith('bp_index')


def fill_state(cfg, seed=0, dtype=torch.float32):
    """deterministic name-hashed fill, O(1) activations (same rule as ga_convnext_oracle.fill_state)"""
    sd = OrderedDict()
    for name, shape in state_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            sd[name] = torch.zeros((), dtype=torch.int64)
            continue
        if leaf == 'bp_index':
            bp = cfg['bp_dim']
            t = torch.triu_indices(bp, bp)
            sd[name] = t[0] * bp + t[1]
            continue
        if leaf == 'running_mean':
            v = rs.uniform(-0.1, 0.1, shape)
        elif leaf == 'running_var':
            v = rs.uniform(0.5, 1.5, shape)
        elif leaf == 'gamma':
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


def is_param(name):
    return not (is_buffer(name) or is_index_buffer(name))


# --------------------------------------------------------------------------------------
# forward restatement
# --------------------------------------------------------------------------------------
def _ln_cf(x, w, b, eps):
    """LayerNorm(data_format='channels_first') (map_convnext.py:165-170): explicit mean / biased variance over dim 1"""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def block(sd, pre, x, dp_mask=None):
    """map_convnext.Block.forward (:27-40): gamma applied in NHWC before the permute"""
    c = x.shape[1]
    y = F.conv2d(x, sd[pre + 'dwconv.weight'], sd[pre + 'dwconv.bias'], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-6)
    y = F.linear(y, sd[pre + 'pwconv1.weight'], sd[pre + 'pwconv1.bias'])
    y = F.gelu(y)
    y = F.linear(y, sd[pre + 'pwconv2.weight'], sd[pre + 'pwconv2.bias'])
    y = sd[pre + 'gamma'] * y
    y = y.permute(0, 3, 1, 2)
    return x + _dp(y, dp_mask)


def forward_features(sd, x, cfg, dp_masks=None):
    """ConvNeXt.forward_features with global_pool='mmcap' (:124-135): [stem output, stage 0..3 outputs]"""
    dp_masks = dp_masks or {}
    feats = []
    for i in range(4):
        p = f'downsample_layers.{i}.'
        if i == 0:
            x = F.conv2d(x, sd[p + '0.weight'], sd[p + '0.bias'], stride=4)
            x = _ln_cf(x, sd[p + '1.weight'], sd[p + '1.bias'], 1e-6)
            feats.append(x)
        else:
            x = _ln_cf(x, sd[p + '0.weight'], sd[p + '0.bias'], 1e-6)
            x = F.conv2d(x, sd[p + '1.weight'], sd[p + '1.bias'], stride=2)
        for j in range(cfg['depths'][i]):
            bp = f'stages.{i}.{j}.'
            x = block(sd, bp, x, dp_masks.get(bp))
        feats.append(x)
    return feats


def multi_scale(sd, pre, feats, level, training, new_stats, size=None):
    """MultiScale.forward (map.py:322-333): smaller maps are ENLARGED by adaptive_avg_pool2d, larger ones REDUCED by
    bilinear interpolation (align_corners=False, no antialias); then conv1x1 (no bias) + BN + GELU (non_linearity=nn.GELU).
    size: explicit target (the builder-defined MAP-ViT composition, whose maps all have one size, reduces to half of it)"""
    h, w = size if size is not None else feats[level].shape[2:]
    ms = []
    for f in feats:
        if h > f.size(2):
            f = F.adaptive_avg_pool2d(f, (h, w))
        elif h < f.size(2):
            f = F.interpolate(f, size=(h, w), mode='bilinear')
        ms.append(f)
    x = F.conv2d(torch.cat(ms, dim=1), sd[pre + '0.weight'])
    return F.gelu(_bn(sd, pre + '1.', x, training, new_stats))


def gram_token(sd, pre, x, cfg, training, new_stats):
    """GramToken.forward (map.py:210-234)"""
    T = cfg['n_tokens']
    x = F.conv2d(x, sd[pre + 'ch_reduction.0.weight'], None, groups=cfg['bp_groups'])
    x = _bn(sd, pre + 'ch_reduction.1.', x, training, new_stats)
    b, c, h, w = x.shape
    x = x.reshape(b, c, h * w) / (h * w)
    attn = x @ x.transpose(-1, -2).contiguous()
    attn = attn.reshape(b, c * c)[:, sd[pre + 'bp_index']]
    attn = F.normalize(attn, dim=-1)
    gdim = attn.shape[1]
    attn = attn.reshape(b, -1, T, 1, 1).permute(0, 2, 1, 3, 4).reshape(b, gdim, 1, 1)
    t = F.conv2d(attn, sd[pre + 'bp_reduction.0.weight'], None, groups=cfg['gram_group'])
    t = _bn(sd, pre + 'bp_reduction.1.', t, training, new_stats)
    return t.reshape(b, cfg['gram_dim'], T).permute(0, 2, 1)


def _drop(x, mask):
    """nn.Dropout with an explicit mask (already divided by keep) or None (= eval / p 0)"""
    return x if mask is None else x * mask.to(x.dtype)


def class_attention(sd, pre, x, n_tokens, cfg, attn_mask=None, proj_mask=None):
    """ClassAttention.forward (map.py:100-144): x = the normalised cat(cls, img) rows (in_dim == dim), or the pair
    (normalised cls rows, normalised img rows) of the dim_mismatch branch (:101-116: k = cat(k1(cls), k2(img)), v likewise)"""
    nh, E = cfg['num_heads'], cfg['ca_dim']
    hd = E // nh
    if isinstance(x, tuple):
        cls, img = x
        b, n1, _ = cls.shape
        n2 = img.shape[1]
        n = n1 + n2
        q = F.linear(cls, sd[pre + 'q.weight'], sd.get(pre + 'q.bias')).reshape(b, n_tokens, nh, hd).permute(0, 2, 1, 3)
        q = q * hd ** -0.5
        k = torch.cat([F.linear(cls, sd[pre + 'k1.weight'], sd.get(pre + 'k1.bias')).reshape(b, n1, nh, hd).permute(0, 2, 1, 3),
                       F.linear(img, sd[pre + 'k2.weight'], sd.get(pre + 'k2.bias')).reshape(b, n2, nh, hd).permute(0, 2, 1, 3)], dim=-2)
        v = torch.cat([F.linear(cls, sd[pre + 'v1.weight'], sd.get(pre + 'v1.bias')).reshape(b, n1, nh, hd).permute(0, 2, 1, 3),
                       F.linear(img, sd[pre + 'v2.weight'], sd.get(pre + 'v2.bias')).reshape(b, n2, nh, hd).permute(0, 2, 1, 3)], dim=-2)
    else:
        cls, img = x[:, :n_tokens], x
        b, n, _ = img.shape
        q = F.linear(cls, sd[pre + 'q.weight'], sd.get(pre + 'q.bias')).reshape(b, n_tokens, nh, hd).permute(0, 2, 1, 3)
        k = F.linear(img, sd[pre + 'k.weight'], sd.get(pre + 'k.bias')).reshape(b, n, nh, hd).permute(0, 2, 1, 3)
        q = q * hd ** -0.5
        v = F.linear(img, sd[pre + 'v.weight'], sd.get(pre + 'v.bias')).reshape(b, n, nh, hd).permute(0, 2, 1, 3)
    attn = q @ k.transpose(-2, -1).contiguous()
    if cfg['interactive']:
        attn = attn + F.linear(attn.permute(0, 2, 3, 1), sd[pre + 'w1.weight'], sd[pre + 'w1.bias']).permute(0, 3, 1, 2)
    attn = attn.softmax(dim=-1)
    if cfg['interactive']:
        attn = attn + F.linear(attn.permute(0, 2, 3, 1), sd[pre + 'w2.weight'], sd[pre + 'w2.bias']).permute(0, 3, 1, 2)
    attn = _drop(attn, attn_mask)
    o = (attn @ v).transpose(1, 2).contiguous().reshape(b, n_tokens, E)
    return _drop(F.linear(o, sd[pre + 'proj.weight'], sd[pre + 'proj.bias']), proj_mask)


def group_conv_mlp(sd, pre, x, groups, hidden_mask=None):
    """GroupConvMlp.forward (map.py:56-66) with act_layer = ReLU (MAPHead default, :467)"""
    t = x.permute(0, 2, 1).unsqueeze(-1)
    t = F.relu(F.conv2d(t, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'], groups=groups))
    if hidden_mask is not None:     # (B, tokens, hidden) -> (B, hidden, tokens, 1)
        t = t * hidden_mask.permute(0, 2, 1).unsqueeze(-1).to(t.dtype)
    t = channel_shuffle(t, groups)
    t = F.conv2d(t, sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias'], groups=groups)
    return t.squeeze(-1).permute(0, 2, 1)


def cap(sd, pre, x, cfg, training, new_stats, masks=None):
    """CAP.forward + CABlock.forward (map.py:264-278, 171-184); masks: dict(attn, proj, mlp) of dropout masks or None"""
    masks = masks or {}
    x_cls = gram_token(sd, pre + 'gram_token_extraction.', x, cfg, training, new_stats)
    b, c, h, w = x.shape
    img = x.reshape(b, c, h * w).permute(0, 2, 1)
    if cfg['self_distill_token']:
        x_cls = torch.cat([x_cls, x_cls.mean(dim=1, keepdim=True)], dim=1)
    nt = x_cls.shape[1]
    ap = pre + 'attention.0.'
    if cfg['gram_dim'] != c:        # dim_mismatch (map.py:174-177): own norms, and the attention output REPLACES the class rows
        cn = F.layer_norm(x_cls, (cfg['gram_dim'],), sd[ap + 'norm1_1.weight'], sd[ap + 'norm1_1.bias'], 1e-6)
        im = F.layer_norm(img, (c,), sd[ap + 'norm1_2.weight'], sd[ap + 'norm1_2.bias'], 1e-6)
        x_cls = class_attention(sd, ap + 'attn.', (cn, im), nt, cfg, masks.get('attn'), masks.get('proj'))
    else:
        u = torch.cat((x_cls, img), dim=1)
        un = F.layer_norm(u, (c,), sd[ap + 'norm1.weight'], sd[ap + 'norm1.bias'], 1e-6)
        x_cls = x_cls + class_attention(sd, ap + 'attn.', un, nt, cfg, masks.get('attn'), masks.get('proj'))
    t = F.layer_norm(x_cls, (c,), sd[ap + 'norm2.weight'], sd[ap + 'norm2.bias'], 1e-6)
    x_cls = x_cls + group_conv_mlp(sd, ap + 'mlp.', t, cfg['mlp_groups'], masks.get('mlp'))
    return x_cls.reshape(b, -1)


def norm_head(sd, pre, x):
    """NormHead.forward (map.py:402-412), nn.LayerNorm default eps 1e-5, dropout 0"""
    x = F.layer_norm(x, (x.shape[1],), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    return F.linear(x, sd[pre + 'head.weight'], sd[pre + 'head.bias'])


def group_head(sd, pre, x, cfg):
    """heads[i] of MAPHead: NormHead, SplitNormHead (map.py:427-441: per-token LayerNorm + Linear, summed) or nn.Linear"""
    hfn = cfg.get('head_fn', 'norm')
    if hfn == 'linear':
        return F.linear(x, sd[pre + 'weight'], sd[pre + 'bias'])
    if hfn == 'split':
        T = cfg['n_tokens']
        xs = x.reshape(x.shape[0], T, -1)
        out = 0
        for t in range(T):
            s = F.layer_norm(xs[:, t], (xs.shape[2],), sd[pre + f'norm.{t}.weight'], sd[pre + f'norm.{t}.bias'], 1e-5)
            out = out + F.linear(s, sd[pre + f'head.{t}.weight'], sd[pre + f'head.{t}.bias'])
        return out
    return norm_head(sd, pre, x)


def map_head(sd, pre, feats, cfg, training, new_stats=None, masks=None):
    """MAPHead.forward (map.py:514-539): train -> [[org_out, avg_out]] per group, eval -> [org_out] per group"""
    masks = masks or {}
    x = multi_scale(sd, pre + 'mmcap.multi_scale.concat_conv.', feats, cfg['multi_scale_level'], training, new_stats,
                    cfg.get('multi_scale_size'))
    out_ch = cfg['last_dim'] * cfg['n_tokens']
    outs = []
    for i in range(cfg['n_groups']):
        pool = cap(sd, f'{pre}mmcap.mmcap.{i}.', x, cfg, training, new_stats, masks.get(i))
        if cfg['self_distill_token']:
            org, avg = pool[:, :out_ch], pool[:, out_ch:]
            org_out = group_head(sd, f'{pre}heads.{i}.', org, cfg)
            outs.append([org_out, norm_head(sd, f'{pre}self_dt_heads.{i}.', avg)] if training else org_out)
        else:
            outs.append(group_head(sd, f'{pre}heads.{i}.', pool, cfg))
    return outs


def forward(sd, x, cfg, training=False, new_stats=None, dp_masks=None, drop_masks=None):
    """map_convnext ConvNeXt.forward (:137-140)"""
    return map_head(sd, 'head.', forward_features(sd, x, cfg, dp_masks), cfg, training, new_stats, drop_masks)


# --------------------------------------------------------------------------------------
# loss / metric
# --------------------------------------------------------------------------------------
def multi_group_loss(outputs, target, dec_lam, kind='ce', smoothing=0.0):
    """MAP/train.py:792-839 with distill_tokens == 0: per group L(y_hat) + KL_sum(log_softmax(y_mean_hat) ||
    log_softmax(y_hat).detach()) / numel; plus, with more than one group, dec_lam * KL_mean(log_softmax(y_hat) ||
    log_softmax(mean_groups y_hat.detach())) per group (reduction='mean': / numel)."""
    loss = 0
    aggre = 0
    for o in outputs:
        if isinstance(o, (tuple, list)):
            y_hat, y_mean_hat = o
            aggre = aggre + y_hat
            adv = F.kl_div(F.log_softmax(y_mean_hat, dim=1), F.log_softmax(y_hat, dim=1).detach(), reduction='sum',
                           log_target=True) / y_hat.numel()
            loss = loss + head_loss(y_hat, target, kind, smoothing) + adv
        else:
            aggre = aggre + o
            loss = loss + head_loss(o, target, kind, smoothing)
    if len(outputs) > 1:
        ref = F.log_softmax(aggre.detach() / len(outputs), dim=1)
        for o in outputs:
            y_hat = o[0] if isinstance(o, (tuple, list)) else o
            loss = loss + F.kl_div(F.log_softmax(y_hat, dim=1), ref, reduction='mean', log_target=True) * dec_lam
    return loss


def validate_output(outputs):
    """MAP/train.py:1000-1006 (and MAP/validate.py:275-279): MEAN of the group logits"""
    s = 0
    for o in outputs:
        s = s + o
    return s / len(outputs)


def no_weight_decay(name, shape):
    return len(shape) <= 1 or name.endswith('.bias')


def train_step_grads(sd, x, target, cfg, dec_lam=-0.8, kind='ce', smoothing=0.0, dp_masks=None, drop_masks=None):
    """One training forward+backward of the restated path. Returns (loss, outputs, grads, new_bn_stats)."""
    names = [n for n in sd if is_param(n)]
    leaf = OrderedDict((n, (sd[n].detach().clone().requires_grad_(True) if is_param(n) else sd[n])) for n in sd)
    new_stats = {}
    outs = forward(leaf, x, cfg, training=True, new_stats=new_stats, dp_masks=dp_masks, drop_masks=drop_masks)
    loss = multi_group_loss(outs, target, dec_lam, kind, smoothing)
    gs = torch.autograd.grad(loss, [leaf[n] for n in names])
    return (loss.detach(), [[a.detach() for a in o] if isinstance(o, (list, tuple)) else o.detach() for o in outs],
            OrderedDict(zip(names, gs)), new_stats)
