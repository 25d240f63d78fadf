"""ORACLE (test infrastructure only -- never imported by the product package).

CPU restatement, in plain PyTorch fp32 ops over a flat ``state_dict``, of the reference's
GA-ConvNeXt hot path:

* model forward ......... /root/reference/GA/ga_convnext.py:51-505
* training loss ......... /root/reference/GA/train.py:735-745
* validate reduction .... /root/reference/GA/train.py:848-860  (sum of head logits -> top-k)
* optimizer step ........ /root/reference/GA/train.py:466,769 (timm create_optimizer_v2 ->
                          torch.optim.SGD(nesterov) / torch.optim.AdamW; timm weight-decay rule)

Backward is torch autograd over this restated forward.  The restatement is *pinned* by
tests/golden/*.npz, which oracle/gen_golden.py produced in the build container by importing
the real reference classes (against oracle/timm_stub) -- see tests/test_oracle_golden.py.
timm itself is un-vendored: Mlp / DropPath / SEModule / optimizer
param-grouping semantics are restated from timm 0.9.x's published behaviour and are
"parity unpinned" with respect to timm.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# Variant table (ga_convnext.py:572-613)
# --------------------------------------------------------------------------------------
VARIANTS = {
    'ga_convnext_tiny_688': dict(depths=(3, 3, 9, 3, 1), dims=(96, 192, 384, 688, 688), dim_embed=168, naggre=2),
    'ga_convnext_tiny_768': dict(depths=(3, 3, 9, 3, 1), dims=(96, 192, 384, 768, 768), dim_embed=192, naggre=2),
    'ga_convnext_small_688': dict(depths=(3, 3, 27, 3, 1), dims=(96, 192, 384, 688, 688), dim_embed=168, naggre=4),
    'ga_convnext_small_768': dict(depths=(3, 3, 27, 3, 1), dims=(96, 192, 384, 768, 768), dim_embed=192, naggre=4),
    'ga_convnext_base_976': dict(depths=(3, 3, 27, 3, 1), dims=(128, 256, 512, 976, 976), dim_embed=240, naggre=4),
    'ga_convnext_base_1024': dict(depths=(3, 3, 27, 3, 1), dims=(128, 256, 512, 1024, 1024), dim_embed=256, naggre=4),
}


def make_cfg(name=None, **over):
    cfg = dict(in_chans=3, num_classes=1000, patch_size=4, depths=(3, 3, 9, 3, 1), dims=(96, 192, 384, 768, 768),
               branches=5, gram_groups=8, dim_embed=128, naggre=2, gram_dim=192, num_heads=8, mlp_groups=4,
               drop_path_rate=0.0)
    if name is not None:
        cfg.update(VARIANTS[name])
    cfg.update(over)
    return cfg


def se_rd_channels(c, rd_ratio=0.25, divisor=8):
    # timm make_divisible(c*rd_ratio, 8, round_limit=0.)  (ga_convnext.py:279)
    return max(divisor, int(c * rd_ratio + divisor / 2) // divisor * divisor)


def drop_path_rates(cfg):
    """ga_convnext.py:362 -- linspace over sum(depths) INCLUDING the trailing 1; the gram_layer
    blocks all take the last point (:413); the Bottleneck takes drop_path_rate itself (:376)."""
    depths = cfg['depths']
    pts = torch.linspace(0, cfg['drop_path_rate'], sum(depths)).split(list(depths))
    return [p.tolist() for p in pts]


def tap_indices(nblocks, naggre):
    """ga_convnext.py:141-147 -- block indices (0-based) after which a tap is taken."""
    taps = []
    if nblocks > 5:
        for i in range(nblocks):
            if (i + 1) % (nblocks // (naggre + 1)) == 0 and len(taps) < naggre:
                taps.append(i)
    return taps


# --------------------------------------------------------------------------------------
# state_dict layout (names + shapes identical to the reference nn.Module's state_dict)
# --------------------------------------------------------------------------------------
def _block_shapes(prefix, c, out):
    out[prefix + 'gamma'] = (c,)
    out[prefix + 'conv_dw.weight'] = (c, 1, 7, 7)
    out[prefix + 'conv_dw.bias'] = (c,)
    out[prefix + 'norm.weight'] = (c,)
    out[prefix + 'norm.bias'] = (c,)
    out[prefix + 'mlp.fc1.weight'] = (4 * c, c)
    out[prefix + 'mlp.fc1.bias'] = (4 * c,)
    out[prefix + 'mlp.fc2.weight'] = (c, 4 * c)
    out[prefix + 'mlp.fc2.bias'] = (c,)


def _bn_shapes(prefix, c, out):
    out[prefix + 'weight'] = (c,)
    out[prefix + 'bias'] = (c,)
    out[prefix + 'running_mean'] = (c,)
    out[prefix + 'running_var'] = (c,)
    out[prefix + 'num_batches_tracked'] = ()


def state_shapes(cfg):
    """Ordered name -> shape map, in the reference module's registration order."""
    d, dep = cfg['dims'], cfg['depths']
    o = OrderedDict()
    p = cfg['patch_size']
    o['stem.0.weight'] = (d[0], cfg['in_chans'], p, p)
    o['stem.0.bias'] = (d[0],)
    o['stem.1.weight'] = (d[0],)
    o['stem.1.bias'] = (d[0],)
    prev = d[0]
    for i in range(4):
        if i > 0:
            o[f'stages.{i}.downsample.0.weight'] = (prev,)
            o[f'stages.{i}.downsample.0.bias'] = (prev,)
            o[f'stages.{i}.downsample.1.weight'] = (d[i], prev, 2, 2)
            o[f'stages.{i}.downsample.1.bias'] = (d[i],)
        for j in range(dep[i]):
            _block_shapes(f'stages.{i}.blocks.{j}.', d[i], o)
        prev = d[i]
    cin = sum(d[:-1]) + d[2] * cfg['naggre']
    cout = d[4]
    w = cout // 4
    o['stages.4.downsample.0.weight'] = (cout, cin, 1, 1)
    o['stages.4.downsample.0.bias'] = (cout,)
    _bn_shapes('stages.4.downsample.1.', cout, o)
    o['stages.4.conv1.weight'] = (w, cin, 1, 1)
    _bn_shapes('stages.4.bn1.', w, o)
    o['stages.4.conv2.weight'] = (w, w, 3, 3)
    _bn_shapes('stages.4.bn2.', w, o)
    rd = se_rd_channels(w)
    o['stages.4.se.fc1.weight'] = (rd, w, 1, 1)
    o['stages.4.se.fc1.bias'] = (rd,)
    o['stages.4.se.fc2.weight'] = (w, rd, 1, 1)
    o['stages.4.se.fc2.bias'] = (w,)
    o['stages.4.conv3.weight'] = (cout, w, 1, 1)
    _bn_shapes('stages.4.bn3.', cout, o)
    g, nb, de = cfg['gram_dim'], cfg['branches'], cfg['dim_embed']
    ntri = (g + 1) * g // 2
    # ModuleList registration order in the reference: gram_contraction, gram_layer, gram_embedding, ga, fc
    for k in range(nb):
        o[f'gram_contraction.{k}.0.weight'] = (g, cout, 1, 1)
        o[f'gram_contraction.{k}.0.bias'] = (g,)
        _bn_shapes(f'gram_contraction.{k}.1.', g, o)
    for k in range(nb):
        _block_shapes(f'gram_layer.{k}.blocks.0.', g, o)
    for k in range(nb):
        o[f'gram_embedding.{k}.0.weight'] = (cout, ntri // cfg['gram_groups'], 1, 1)
        o[f'gram_embedding.{k}.0.bias'] = (cout,)
        _bn_shapes(f'gram_embedding.{k}.1.', cout, o)
    mg = cfg['mlp_groups']
    for k in range(nb):
        pre = f'ga.{k}.'
        o[pre + 'gamma_1'] = (cout,)
        o[pre + 'gamma_2'] = (cout,)
        o[pre + 'norm1.weight'] = (cout,)
        o[pre + 'norm1.bias'] = (cout,)
        o[pre + 'attn.q.weight'] = (de, cout)
        o[pre + 'attn.k.weight'] = (de, cout)
        o[pre + 'attn.v.weight'] = (de, cout)
        o[pre + 'attn.proj.weight'] = (cout, de)
        o[pre + 'attn.proj.bias'] = (cout,)
        o[pre + 'norm2.weight'] = (cout,)
        o[pre + 'norm2.bias'] = (cout,)
        o[pre + 'mlp.fc1.weight'] = (4 * cout, cout // mg, 1, 1)
        o[pre + 'mlp.fc1.bias'] = (4 * cout,)
        o[pre + 'mlp.fc2.weight'] = (cout, 4 * cout // mg, 1, 1)
        o[pre + 'mlp.fc2.bias'] = (cout,)
    for k in range(nb):
        o[f'fc.{k}.weight'] = (cfg['num_classes'], cout)
        o[f'fc.{k}.bias'] = (cfg['num_classes'],)
    return o


def is_buffer(name):
    return name.endswith('running_mean') or name.endswith('running_var') or name.endswith('num_batches_tracked')


def fill_state(cfg, seed=0, dtype=torch.float32):
    """Deterministic, name-hashed fill (independent of RNG draw order), scaled so that activations and
    logits are O(1) -- the reference's own init (trunc-normal .02, gamma=1e-6) gives ~1e-3 logits, for
    which a 1e-3 *relative* parity gate would be meaningless."""
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
        elif leaf in ('gamma', 'gamma_1', 'gamma_2'):
            v = rs.uniform(0.4, 0.9, shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(shape) * (1.0 / math.sqrt(fan_in))
        elif leaf == 'weight':  # norm scale
            v = rs.uniform(0.8, 1.2, shape)
        else:  # biases
            v = rs.uniform(-0.1, 0.1, shape)
        sd[name] = torch.tensor(v, dtype=dtype)
    return sd


def gen_input(batch, seed=0, size=224):
    """Closed-form-seeded input batch (not committed; regenerated identically on both sides)."""
    g = torch.Generator().manual_seed(1234 + seed)
    return torch.randn(batch, 3, size, size, generator=g)


# --------------------------------------------------------------------------------------
# forward restatement
# --------------------------------------------------------------------------------------
def _ln_c(x, w, b, eps):
    """LayerNorm over the channel dim of an NCHW tensor (ga_convnext.py:51-67)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def _bn(sd, pre, x, training, new_stats, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d (ga_convnext.py:261,270,276,283,409,420): batch stats in train (per process)."""
    if training:
        dims = [0] + list(range(2, x.ndim))
        mean = x.mean(dims)
        var_b = x.var(dims, unbiased=False)
        n = x.numel() // x.shape[1]
        if new_stats is not None:
            var_u = var_b * (n / max(n - 1, 1))
            new_stats[pre + 'running_mean'] = ((1 - momentum) * sd[pre + 'running_mean'] + momentum * mean).detach()
            new_stats[pre + 'running_var'] = ((1 - momentum) * sd[pre + 'running_var'] + momentum * var_u).detach()
            new_stats[pre + 'num_batches_tracked'] = sd[pre + 'num_batches_tracked'] + 1
    else:
        mean, var_b = sd[pre + 'running_mean'], sd[pre + 'running_var']
    shp = [1, -1] + [1] * (x.ndim - 2)
    xh = (x - mean.reshape(shp)) * torch.rsqrt(var_b.reshape(shp) + eps)
    return xh * sd[pre + 'weight'].reshape(shp) + sd[pre + 'bias'].reshape(shp)


def _dp(x, mask):
    """timm DropPath: per-sample mask already divided by keep-prob (or None = identity)."""
    if mask is None:
        return x
    return x * mask.reshape([-1] + [1] * (x.ndim - 1)).to(x.dtype)


# 'erf' = nn.GELU() of the reference (ga_convnext.py:94).  'tanh' exists ONLY so that tests can measure how much of the bf16
# throughput mode's deviation comes from the tanh-form GELU its fc1 epilogue evaluates (csrc/common.h gelu_both_fast).
GELU_FORM = 'erf'


def _gelu(y):
    return F.gelu(y, approximate='tanh') if GELU_FORM == 'tanh' else F.gelu(y)


def convnext_block(sd, pre, x, dp_mask=None):
    """ConvNeXtBlock.forward (ga_convnext.py:98-112)."""
    c = x.shape[1]
    y = F.conv2d(x, sd[pre + 'conv_dw.weight'], sd[pre + 'conv_dw.bias'], padding=3, groups=c)
    y = y.permute(0, 2, 3, 1)
    y = F.layer_norm(y, (c,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-6)
    y = F.linear(y, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'])
    y = _gelu(y)
    y = F.linear(y, sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias'])
    y = y.permute(0, 3, 1, 2)
    y = y * sd[pre + 'gamma'].reshape(1, -1, 1, 1)
    return _dp(y, dp_mask) + x


def channel_shuffle(x, group):
    """ga_convnext.py:557-566."""
    b, c, h, w = x.shape
    return x.reshape(b, c // group, group, h, w).permute(0, 2, 1, 3, 4).reshape(b, c, h, w)


def gram_index(g):
    """Row-major upper-triangular (i<=j) flat indices (ga_convnext.py:424-430)."""
    i, j = np.triu_indices(g)
    return torch.from_numpy((i * g + j).astype(np.int64))


def get_gram(x, training):
    """GA_ConvNeXt.get_gram (ga_convnext.py:452-467), incl. the fp64 branch for train & B<128."""
    b, c, h, w = x.shape
    in_dtype = x.dtype
    x = x / h
    if training and b < 128:
        x = x.to(torch.float64)
    x = x.reshape(b, c, h * w)
    g = torch.bmm(x, x.transpose(1, 2)) / (h * w)
    g = g.reshape(b, c * c)[:, gram_index(c)]
    g = F.normalize(g)
    # reference: .float(); a float64 *ground-truth* run of this oracle (tests only) stays in float64
    g = g.double() if in_dtype == torch.float64 else g.float()
    return g.reshape(b, -1, 1, 1)


def class_attn_block(sd, pre, x_tok, x_cls, cfg, dp_mask=None):
    """LayerScaleBlockClassAttn.forward + ClassAttn + GroupConvMlp (ga_convnext.py:153-248)."""
    c = x_tok.shape[2]
    nh, de = cfg['num_heads'], cfg['dim_embed']
    hd = de // nh
    u = torch.cat((x_cls, x_tok), dim=1)
    un = F.layer_norm(u, (c,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-5)
    b, n, _ = un.shape
    q = F.linear(un[:, 0], sd[pre + 'attn.q.weight']).reshape(b, 1, nh, hd).permute(0, 2, 1, 3)
    k = F.linear(un, sd[pre + 'attn.k.weight']).reshape(b, n, nh, hd).permute(0, 2, 1, 3)
    v = F.linear(un, sd[pre + 'attn.v.weight']).reshape(b, n, nh, hd).permute(0, 2, 1, 3)
    attn = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(b, 1, de)
    a = F.linear(a, sd[pre + 'attn.proj.weight'], sd[pre + 'attn.proj.bias'])
    x_cls = x_cls + _dp(sd[pre + 'gamma_1'] * a, dp_mask)
    t = F.layer_norm(x_cls, (c,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-5)
    mg = cfg['mlp_groups']
    t = t.permute(0, 2, 1).unsqueeze(-1)
    t = F.conv2d(t, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'], groups=mg)
    t = F.gelu(t)
    t = channel_shuffle(t, mg)
    t = F.conv2d(t, sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias'], groups=mg)
    t = t.squeeze(-1).permute(0, 2, 1)
    return x_cls + _dp(sd[pre + 'gamma_2'] * t, dp_mask)


def bottleneck(sd, pre, x, training, new_stats, dp_mask=None):
    """Bottleneck.forward (ga_convnext.py:294-318) with timm SEModule (rd_ratio 1/4)."""
    y = F.conv2d(x, sd[pre + 'conv1.weight'])
    y = F.relu(_bn(sd, pre + 'bn1.', y, training, new_stats))
    y = F.conv2d(y, sd[pre + 'conv2.weight'], padding=1)
    y = F.relu(_bn(sd, pre + 'bn2.', y, training, new_stats))
    s = y.mean((2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, sd[pre + 'se.fc1.weight'], sd[pre + 'se.fc1.bias']))
    s = F.conv2d(s, sd[pre + 'se.fc2.weight'], sd[pre + 'se.fc2.bias'])
    y = y * torch.sigmoid(s)
    y = F.conv2d(y, sd[pre + 'conv3.weight'])
    y = _bn(sd, pre + 'bn3.', y, training, new_stats)
    y = _dp(y, dp_mask)
    sc = F.conv2d(x, sd[pre + 'downsample.0.weight'], sd[pre + 'downsample.0.bias'])
    sc = _bn(sd, pre + 'downsample.1.', sc, training, new_stats)
    return F.relu(y + sc)


def forward_features(sd, x, cfg, training=False, new_stats=None, dp_masks=None, taps_out=None):
    """GA_ConvNeXt.forward_features (ga_convnext.py:469-485)."""
    dp_masks = dp_masks or {}
    dep = cfg['depths']
    x = F.conv2d(x, sd['stem.0.weight'], sd['stem.0.bias'], stride=cfg['patch_size'])
    x = _ln_c(x, sd['stem.1.weight'], sd['stem.1.bias'], 1e-6)
    feats, taps = [], []
    for i in range(4):
        if i > 0:
            x = _ln_c(x, sd[f'stages.{i}.downsample.0.weight'], sd[f'stages.{i}.downsample.0.bias'], 1e-6)
            x = F.conv2d(x, sd[f'stages.{i}.downsample.1.weight'], sd[f'stages.{i}.downsample.1.bias'], stride=2)
        tap_at = tap_indices(dep[i], cfg['naggre']) if i == 2 else []
        for j in range(dep[i]):
            pre = f'stages.{i}.blocks.{j}.'
            x = convnext_block(sd, pre, x, dp_masks.get(pre))
            if j in tap_at:
                taps.append(x)
        feats.append(x)
    cat = torch.cat([F.adaptive_avg_pool2d(feats[0], 14), F.adaptive_avg_pool2d(feats[1], 14)] + taps +
                    [feats[2], F.interpolate(feats[3], scale_factor=2, mode='bilinear')], dim=1)
    if taps_out is not None:
        taps_out['cat'] = cat
    return bottleneck(sd, 'stages.4.', cat, training, new_stats, dp_masks.get('stages.4.'))


def forward(sd, x, cfg, training=False, new_stats=None, dp_masks=None):
    """GA_ConvNeXt.forward (ga_convnext.py:487-505): returns the list of per-head logits."""
    dp_masks = dp_masks or {}
    x = forward_features(sd, x, cfg, training, new_stats, dp_masks)
    b, c = x.shape[:2]
    tok = x.reshape(b, c, -1).permute(0, 2, 1)
    outs = []
    for k in range(cfg['branches']):
        g = F.conv2d(x, sd[f'gram_contraction.{k}.0.weight'], sd[f'gram_contraction.{k}.0.bias'])
        g = _bn(sd, f'gram_contraction.{k}.1.', g, training, new_stats)
        pre = f'gram_layer.{k}.blocks.0.'
        g = convnext_block(sd, pre, g, dp_masks.get(pre))
        g = get_gram(g, training)
        g = F.conv2d(g, sd[f'gram_embedding.{k}.0.weight'], sd[f'gram_embedding.{k}.0.bias'],
                     groups=cfg['gram_groups'])
        g = _bn(sd, f'gram_embedding.{k}.1.', g, training, new_stats)
        cls = g.reshape(b, c, -1).permute(0, 2, 1)
        cls = class_attn_block(sd, f'ga.{k}.', tok, cls, cfg, dp_masks.get(f'ga.{k}.'))
        outs.append(F.linear(cls.reshape(b, -1), sd[f'fc.{k}.weight'], sd[f'fc.{k}.bias']))
    return outs


# --------------------------------------------------------------------------------------
# loss / metric / optimizer restatements
# --------------------------------------------------------------------------------------
def _bce_target(target, num_classes, smoothing):
    # timm BinaryCrossEntropy: off = s/C, on = 1 - s + off
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    t = torch.full((target.shape[0], num_classes), off, dtype=torch.float32)
    return t.scatter_(1, target.view(-1, 1), on)


def head_loss(out, target, kind='ce', smoothing=0.0, bce_threshold=None):
    """the per-head loss GA/train.py:608-630 selects.  A floating-point (B, C) target is the dense target mixup / cutmix
    produce: 'ce' then is timm SoftTargetCrossEntropy, 'bce' timm BinaryCrossEntropy on it (smoothing already inside)."""
    if target.dim() == 2:
        if kind in ('ce', 'soft'):
            return torch.sum(-target * F.log_softmax(out, dim=-1), dim=-1).mean()
        t = target.gt(bce_threshold).to(target.dtype) if bce_threshold is not None else target
        return F.binary_cross_entropy_with_logits(out, t, reduction='mean')
    if kind == 'ce':
        if smoothing > 0:  # timm LabelSmoothingCrossEntropy
            logp = F.log_softmax(out, dim=-1)
            nll = -logp.gather(1, target.view(-1, 1)).squeeze(1)
            return ((1 - smoothing) * nll + smoothing * (-logp.mean(dim=-1))).mean()
        return F.cross_entropy(out, target)
    if kind == 'bce':
        t = _bce_target(target, out.shape[1], smoothing).to(out.dtype)
        if bce_threshold is not None:
            t = t.gt(bce_threshold).to(t.dtype)
        return F.binary_cross_entropy_with_logits(out, t, reduction='mean')
    if kind == 'soft':  # timm SoftTargetCrossEntropy; target is (B, C) float
        return torch.sum(-target * F.log_softmax(out, dim=-1), dim=-1).mean()
    raise ValueError(kind)


def ga_loss(outputs, target, lam, kind='ce', smoothing=0.0, bce_threshold=None):
    """GA/train.py:735-745: sum_k L(out_k) + lam * sum_k KL_mean(log_softmax(out_k) || log_softmax(mean_j out_j.detach()))
    with reduction='mean' (divide by B*C) and log_target=True."""
    loss = 0
    summed = 0
    for out in outputs:
        loss = loss + head_loss(out, target, kind, smoothing, bce_threshold)
        summed = summed + out.detach()
    ref = F.log_softmax(summed / len(outputs), dim=1)
    for out in outputs:
        loss = loss + F.kl_div(F.log_softmax(out, dim=1), ref, reduction='mean', log_target=True) * lam
    return loss


def validate_output(outputs):
    """GA/train.py:848-851: sum of the head logits in fp32."""
    s = 0
    for o in outputs:
        s = s + o.float()
    return s


def topk_indices(output, k=5):
    """timm accuracy(): output.topk(k, 1, True, True) indices."""
    return output.topk(min(k, output.shape[1]), 1, True, True)[1]


def accuracy(output, target, topk=(1, 5)):
    maxk = min(max(topk), output.shape[1])
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.reshape(1, -1).expand_as(pred))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.shape[0] for k in topk]


def no_weight_decay(name, shape):
    """timm param_groups_weight_decay: ndim<=1 or name ends with '.bias' -> wd 0."""
    return len(shape) <= 1 or name.endswith('.bias')


def sgd_nesterov_step(params, grads, bufs, lr, momentum=0.9, weight_decay=0.0, first=True):
    """torch.optim.SGD(nesterov=True, dampening=0) over a name->tensor dict; returns new (params, bufs)."""
    newp, newb = OrderedDict(), OrderedDict()
    for n, p in params.items():
        g = grads[n]
        wd = 0.0 if no_weight_decay(n, p.shape) else weight_decay
        g = g + wd * p
        b = g.clone() if (first or n not in bufs) else momentum * bufs[n] + g
        g = g + momentum * b
        newp[n] = p - lr * g
        newb[n] = b
    return newp, newb


def adamw_step(params, grads, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW (decoupled wd, bias-corrected) -- one step; `step` is 1-based."""
    b1, b2 = betas
    newp, newm, newv = OrderedDict(), OrderedDict(), OrderedDict()
    for n, p in params.items():
        g = grads[n]
        wd = 0.0 if no_weight_decay(n, p.shape) else weight_decay
        p1 = p * (1 - lr * wd)
        m1 = b1 * m[n] + (1 - b1) * g if n in m else (1 - b1) * g
        v1 = b2 * v[n] + (1 - b2) * g * g if n in v else (1 - b2) * g * g
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = v1.sqrt() / math.sqrt(bc2) + eps
        newp[n] = p1 - (lr / bc1) * m1 / denom
        newm[n], newv[n] = m1, v1
    return newp, newm, newv


def lamb_step(params, grads, m, v, step, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, max_grad_norm=1.0,
              grad_averaging=True):
    """timm.optim.Lamb (timm 0.9.x, un-vendored: restated from the published algorithm -- You et al. 2020 -- and the
    timm source as recalled; PARITY UNPINNED against timm itself) -- one step, `step` 1-based:
    global grad-norm clip to max_grad_norm, Adam moments (bias-corrected), update += wd * p, per-tensor trust ratio
    ||p|| / ||update|| (only where weight decay applies), p -= lr * trust * update.
    Weight-decay grouping as create_optimizer_v2 (GA/train.py:466): none for ndim <= 1 / *.bias."""
    b1, b2 = betas
    gn = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values()))
    clip = gn / max_grad_norm if gn > max_grad_norm else 1.0
    beta3 = 1 - b1 if grad_averaging else 1.0
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    newp, newm, newv = OrderedDict(), OrderedDict(), OrderedDict()
    for n, p in params.items():
        g = grads[n] / clip
        m1 = b1 * m[n] + beta3 * g if n in m else beta3 * g
        v1 = b2 * v[n] + (1 - b2) * g * g if n in v else (1 - b2) * g * g
        upd = (m1 / bc1) / (v1.sqrt() / math.sqrt(bc2) + eps)
        wd = 0.0 if no_weight_decay(n, p.shape) else weight_decay
        if wd != 0:
            upd = upd + wd * p
            wn, un = float(p.norm()), float(upd.norm())
            trust = wn / un if (wn > 0 and un > 0) else 1.0
            upd = upd * trust
        newp[n] = p - lr * upd
        newm[n], newv[n] = m1, v1
    return newp, newm, newv


def grad_errors(got, ref):
    """Per-parameter normalised gradient error used by every parity test.

    Parameters whose reference gradient is analytically zero (biases feeding a train-mode BatchNorm, the last
    stage-3 fc2 bias) hold only round-off noise: for |g_ref|_max < 1e-4 * (global max) the error is taken
    absolutely against 0.1 * global max; everything else is max|a-b| / max|b|."""
    gmax = max(float(g.abs().max()) for g in ref.values())
    errs = {}
    for n, b in ref.items():
        d = float((got[n].to(b.dtype) - b).abs().max())
        bm = float(b.abs().max())
        errs[n] = d / bm if bm >= 1e-4 * gmax else d / (0.1 * gmax)
    return errs


def train_step_grads(sd, x, target, cfg, lam=-0.8, kind='ce', smoothing=0.0, dp_masks=None):
    """One training forward+backward of the restated path. Returns (loss, outputs, grads, new_bn_stats)."""
    names = [n for n in sd if not is_buffer(n)]
    leaf = OrderedDict((n, (sd[n].detach().clone().requires_grad_(True) if not is_buffer(n) else sd[n]))
                       for n in sd)
    new_stats = {}
    outs = forward(leaf, x, cfg, training=True, new_stats=new_stats, dp_masks=dp_masks)
    loss = ga_loss(outs, target, lam, kind, smoothing)
    gs = torch.autograd.grad(loss, [leaf[n] for n in names])
    return loss.detach(), [o.detach() for o in outs], OrderedDict(zip(names, gs)), new_stats
