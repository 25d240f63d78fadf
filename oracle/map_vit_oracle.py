"""TEST INFRASTRUCTURE (CPU oracle): MAP-ViT -- a timm VisionTransformer trunk feeding the reference's MAPHead.

The trunk is the arithmetic of timm.models.vision_transformer (PatchEmbed, cls_token + pos_embed, `Block`: the block the
reference imports as `transformer_block`, /root/reference/MAP/models/map_pit.py:14,35-44 -- LayerNorm(eps 1e-6), qkv with bias,
softmax(q k^T / sqrt(d)) v, proj, DropPath, LayerNorm, Mlp(GELU)), restated from timm's published code (timm is not vendored in
the reference and not installed: parity of the BLOCK is unpinned).  The COMPOSITION is builder-defined (BASELINE configs[4]
"MAP-ViT-B/16 @ 384"; the reference registers no such model): it follows how map_pit.py hands features to MAPHead
(PoolingTransformer.forward_features :185-201) -- the position-embedded patch tokens and the output of every "stage" (the
blocks cut into three equal runs, the last one after the final norm), class token dropped -- and MultiScale (map.py:322-333)
reduces every map to half the token grid.  The head is oracle.map_oracle's restatement of map.py (pinned there)."""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import map_oracle as MO

VARIANTS = {
    'map_vit_base_patch16_384': dict(img_size=384, patch_size=16, embed_dim=768, depth=12, vit_heads=12),
    'map_vit_base_patch16_224': dict(img_size=224, patch_size=16, embed_dim=768, depth=12, vit_heads=12),
    'map_vit_small_patch16_224': dict(img_size=224, patch_size=16, embed_dim=384, depth=12, vit_heads=6),
}


def make_cfg(name=None, **over):
    cfg = dict(in_chans=3, num_classes=1000, img_size=224, patch_size=16, embed_dim=768, depth=12, vit_heads=12, drop_path_rate=0.0,
               # MAPHead arguments: the ones map_convnext_tiny uses (map_convnext.py:201-205), mlp_groups / GELU as map_pit.py:137-144
               last_dim=384, n_groups=4, n_tokens=2, gram_group=24, bp_dim=384, bp_groups=1, gram_dim=None, ca_dim=384, num_heads=12,
               self_distill_token=True, multi_scale_level=0, mlp_ratio=4, mlp_groups=2, interactive=False)
    if name is not None:
        cfg.update(VARIANTS[name])
    cfg.update(over)
    if cfg['gram_dim'] is None:
        cfg['gram_dim'] = cfg['last_dim']
    d = cfg['depth']
    cfg.setdefault('taps', (d // 3, 2 * d // 3, d))
    gw = cfg['img_size'] // cfg['patch_size']
    cfg['multi_scale_size'] = (gw // 2, gw // 2)
    return cfg


def state_shapes(cfg):
    C, ps, gw = cfg['embed_dim'], cfg['patch_size'], cfg['img_size'] // cfg['patch_size']
    o = OrderedDict()
    o['cls_token'] = (1, 1, C)
    o['pos_embed'] = (1, gw * gw + 1, C)
    o['patch_embed.proj.weight'] = (C, cfg['in_chans'], ps, ps)
    o['patch_embed.proj.bias'] = (C,)
    for i in range(cfg['depth']):
        p = f'blocks.{i}.'
        MO._ln_shapes(p + 'norm1.', C, o)
        o[p + 'attn.qkv.weight'] = (3 * C, C)
        o[p + 'attn.qkv.bias'] = (3 * C,)
        o[p + 'attn.proj.weight'] = (C, C)
        o[p + 'attn.proj.bias'] = (C,)
        MO._ln_shapes(p + 'norm2.', C, o)
        o[p + 'mlp.fc1.weight'] = (4 * C, C)
        o[p + 'mlp.fc1.bias'] = (4 * C,)
        o[p + 'mlp.fc2.weight'] = (C, 4 * C)
        o[p + 'mlp.fc2.bias'] = (C,)
    MO._ln_shapes('norm.', C, o)
    MO.head_shapes('head.', cfg, [C] * (len(cfg['taps']) + 1), o)
    return o


def fill_state(cfg, seed=0, dtype=torch.float32):
    sd = OrderedDict()
    for name, shape in state_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif leaf == 'bp_index':
            bp = cfg['bp_dim']
            t = torch.triu_indices(bp, bp)
            sd[name] = t[0] * bp + t[1]
        else:
            if leaf == 'running_mean':
                v = rs.uniform(-0.1, 0.1, shape)
            elif leaf == 'running_var':
                v = rs.uniform(0.5, 1.5, shape)
            elif name in ('cls_token', 'pos_embed'):
                v = rs.standard_normal(shape) * 0.5
            elif len(shape) >= 2:
                v = rs.standard_normal(shape) * (1.0 / math.sqrt(int(np.prod(shape[1:]))))
            elif leaf == 'weight':
                v = rs.uniform(0.8, 1.2, shape)
            else:
                v = rs.uniform(-0.1, 0.1, shape)
            sd[name] = torch.tensor(v, dtype=dtype)
    return sd


def gen_input(batch, seed=0, size=224):
    g = torch.Generator().manual_seed(1234 + seed)
    return torch.randn(batch, 3, size, size, generator=g)


def vit_block(sd, pre, x, heads, dp_mask=None):
    """timm vision_transformer.Block.forward: x + dp(attn(norm1(x))); x + dp(mlp(norm2(x)))   (no LayerScale: init_values None)"""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-6)
    qkv = F.linear(h, sd[pre + 'attn.qkv.weight'], sd[pre + 'attn.qkv.bias']).reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q * (C // heads) ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (att @ v).transpose(1, 2).reshape(B, N, C)
    a = F.linear(a, sd[pre + 'attn.proj.weight'], sd[pre + 'attn.proj.bias'])
    m1, m2 = dp_mask if dp_mask is not None else (None, None)
    x = x + (a if m1 is None else a * m1.reshape(-1, 1, 1).to(a.dtype))
    h = F.layer_norm(x, (C,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-6)
    h = F.linear(F.gelu(F.linear(h, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'])), sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias'])
    return x + (h if m2 is None else h * m2.reshape(-1, 1, 1).to(h.dtype))


def forward_features(sd, x, cfg, dp_masks=None):
    """list of (B, C, gw, gw) maps: the embedded tokens, then the output after every tap (the last through the final norm)"""
    dp_masks = dp_masks or {}
    C, ps = cfg['embed_dim'], cfg['patch_size']
    B = x.shape[0]
    t = F.conv2d(x, sd['patch_embed.proj.weight'], sd['patch_embed.proj.bias'], stride=ps)      # (B, C, gw, gw)
    gw = t.shape[-1]
    t = t.flatten(2).transpose(1, 2)
    x = torch.cat([sd['cls_token'].expand(B, -1, -1).to(t.dtype), t], dim=1) + sd['pos_embed']

    def to_map(z):
        return z[:, 1:].transpose(1, 2).reshape(B, C, gw, gw)
    feats = [to_map(x)]
    for i in range(cfg['depth']):
        x = vit_block(sd, f'blocks.{i}.', x, cfg['vit_heads'], (dp_masks.get(f'blocks.{i}.#1'), dp_masks.get(f'blocks.{i}.#2')))
        if i + 1 in cfg['taps']:
            z = x
            if i + 1 == cfg['depth']:
                z = F.layer_norm(x, (C,), sd['norm.weight'], sd['norm.bias'], 1e-6)
            feats.append(to_map(z))
    return feats


def forward(sd, x, cfg, training=False, new_stats=None, dp_masks=None, drop_masks=None):
    return MO.map_head(sd, 'head.', forward_features(sd, x, cfg, dp_masks), cfg, training, new_stats, drop_masks)


def is_param(name):
    return not (name.endswith('running_mean') or name.endswith('running_var') or name.endswith('num_batches_tracked') or name.endswith('bp_index'))


def train_step_grads(sd, x, target, cfg, dec_lam=-0.8, dp_masks=None, drop_masks=None):
    names = [n for n in sd if is_param(n)]
    leaf = OrderedDict((n, sd[n].detach().clone().requires_grad_(True) if is_param(n) else sd[n]) for n in sd)
    new_stats = {}
    outs = forward(leaf, x, cfg, training=True, new_stats=new_stats, dp_masks=dp_masks, drop_masks=drop_masks)
    loss = MO.multi_group_loss(outs, target, dec_lam)
    gs = torch.autograd.grad(loss, [leaf[n] for n in names])
    det = [[o[0].detach(), o[1].detach()] if isinstance(o, (list, tuple)) else o.detach() for o in outs]
    return loss.detach(), det, OrderedDict(zip(names, gs)), new_stats
