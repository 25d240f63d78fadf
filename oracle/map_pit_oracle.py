"""TEST INFRASTRUCTURE (CPU oracle): functional restatement of /root/reference/MAP/models/map_pit.py -- PoolingTransformer with
pool_type='map' (map_pit_s, :224-251): conv_embedding (:71-81, patch 16 / stride 8), pos_embed added in NCHW (:190-191),
Transformer stages of timm ViT Blocks (:23-55; the Block is timm's, restated in map_vit_oracle.vit_block), conv_head_pooling
(:58-68: 3x3 / stride 2 / pad 1 depthwise conv with channel multiplier 2), the feature list of forward_features (:185-201)
handed to MAPHead (:133-144; restated in map_oracle).  Pinned by tests/golden/pit_*.npz, written by oracle/gen_golden_pit.py
from the REAL reference classes (timm's Block comes from the test-only stub)."""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import map_oracle as MO
from .map_vit_oracle import vit_block

VARIANTS = {
    # map_pit.py:224-242
    'map_pit_s': dict(image_size=224, patch_size=16, stride=8, base_dims=(48, 48, 48), depth=(2, 6, 4), heads=(3, 6, 12), last_dim=384,
                      n_groups=2, n_tokens=4, gram_group=32),
}


def make_cfg(name=None, **over):
    cfg = dict(in_chans=3, num_classes=1000, image_size=224, patch_size=16, stride=8, base_dims=(48, 48, 48), depth=(2, 6, 4),
               heads=(3, 6, 12), mlp_ratio=4, drop_path_rate=0.0,
               # PoolingTransformer defaults (:88-91) and the MAPHead arguments it hard-codes (:137-143)
               last_dim=384, n_groups=4, n_tokens=3, gram_group=24, self_distill_token=True, multi_scale_level=2, mlp_groups=2,
               bp_groups=1, ca_dim=192, num_heads=12, interactive=False)
    if name is not None:
        cfg.update(VARIANTS[name])
    cfg.update(over)
    cfg['base_dims'], cfg['depth'], cfg['heads'] = tuple(cfg['base_dims']), tuple(cfg['depth']), tuple(cfg['heads'])
    cfg['bp_dim'] = cfg['gram_dim'] = cfg['last_dim']
    cfg['dims'] = tuple(b * h for b, h in zip(cfg['base_dims'], cfg['heads']))
    cfg['width'] = (cfg['image_size'] - cfg['patch_size']) // cfg['stride'] + 1
    return cfg


def drop_path_rates(cfg):
    """map_pit.py:116-118: drop_path_rate * i / total_block over all blocks"""
    tot = sum(cfg['depth'])
    out, i = {}, 0
    for s, d in enumerate(cfg['depth']):
        for j in range(d):
            out[f'transformers.{s}.blocks.{j}.'] = cfg['drop_path_rate'] * i / tot
            i += 1
    return out


def state_shapes(cfg):
    d, ps = cfg['dims'], cfg['patch_size']
    o = OrderedDict()
    o['pos_embed'] = (1, d[0], cfg['width'], cfg['width'])
    o['patch_embed.conv.weight'] = (d[0], cfg['in_chans'], ps, ps)
    o['patch_embed.conv.bias'] = (d[0],)
    for s in range(len(d)):
        C = d[s]
        for j in range(cfg['depth'][s]):
            p = f'transformers.{s}.blocks.{j}.'
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
    # nn.ModuleList registration order in the reference: all transformers, then all pools (map_pit.py:113-131 appends alternately,
    # but `transformers` is registered before `pools`)
    for s in range(len(d) - 1):
        o[f'pools.{s}.conv.weight'] = (d[s + 1], 1, 3, 3)
        o[f'pools.{s}.conv.bias'] = (d[s + 1],)
    MO.head_shapes('head.', cfg, [d[0]] + list(d), o)
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
            elif name == 'pos_embed':
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


def forward_features(sd, x, cfg, dp_masks=None):
    """PoolingTransformer.forward_features (:185-201)"""
    dp_masks = dp_masks or {}
    x = F.conv2d(x, sd['patch_embed.conv.weight'], sd['patch_embed.conv.bias'], stride=cfg['stride']) + sd['pos_embed']
    feats = [x]
    ns = len(cfg['dims'])
    for s in range(ns):
        B, C, H, W = x.shape
        t = x.flatten(2).transpose(1, 2)                        # rearrange 'b c h w -> b (h w) c'
        for j in range(cfg['depth'][s]):
            p = f'transformers.{s}.blocks.{j}.'
            m = dp_masks.get(p)
            t = vit_block(sd, p, t, cfg['heads'][s], (m, m) if m is not None and not isinstance(m, tuple) else m)
        x = t.transpose(1, 2).reshape(B, C, H, W)
        feats.append(x)
        if s < ns - 1:
            x = F.conv2d(x, sd[f'pools.{s}.conv.weight'], sd[f'pools.{s}.conv.bias'], stride=2, padding=1, groups=C)
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
