"""TEST INFRASTRUCTURE (CPU oracle): the plain ConvNeXt of /root/reference/MAP/models/map_convnext.py with global_pool='avg'
(registered there as convnext_tiny / convnext_small, :186-196,214-224): the trunk of map_oracle.forward_features, then
forward_features' `self.norm(x.mean([-2, -1]))` (:134-135; nn.LayerNorm(dims[-1], eps=1e-6), :112) and `self.head` (nn.Linear,
:113).  Pinned by tests/golden/cnx_*.npz, written by oracle/gen_golden_cnx.py from the REAL reference class."""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import map_oracle as MO

VARIANTS = {
    'convnext_tiny': dict(depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)),
    'convnext_small': dict(depths=(3, 3, 27, 3), dims=(96, 192, 384, 768)),
}


def make_cfg(name=None, **over):
    base = dict(VARIANTS[name]) if name else {}
    base.update(over)
    cfg = MO.make_cfg(**base)
    cfg['family'] = 'convnext'
    return cfg


def state_shapes(cfg):
    o = OrderedDict((k, v) for k, v in MO.state_shapes(cfg).items() if not k.startswith('head.'))
    c = cfg['dims'][-1]
    o['norm.weight'] = (c,)
    o['norm.bias'] = (c,)
    o['head.weight'] = (cfg['num_classes'], c)
    o['head.bias'] = (cfg['num_classes'],)
    return o


def fill_state(cfg, seed=0, dtype=torch.float32):
    """the same name-hashed fill as map_oracle.fill_state (the trunk entries are identical to the MAP model's)"""
    import math
    import zlib

    import numpy as np
    sd = OrderedDict()
    for name, shape in state_shapes(cfg).items():
        rs = np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = name.rsplit('.', 1)[-1]
        if leaf == 'gamma':
            v = rs.uniform(0.4, 0.9, shape)
        elif len(shape) >= 2:
            v = rs.standard_normal(shape) * (1.0 / math.sqrt(int(np.prod(shape[1:]))))
        elif leaf == 'weight':
            v = rs.uniform(0.8, 1.2, shape)
        else:
            v = rs.uniform(-0.1, 0.1, shape)
        sd[name] = torch.tensor(v, dtype=dtype)
    return sd


gen_input = MO.gen_input
drop_path_rates = MO.drop_path_rates


def forward(sd, x, cfg, dp_masks=None):
    f = MO.forward_features(sd, x, cfg, dp_masks)[-1]
    p = F.layer_norm(f.mean([-2, -1]), (f.shape[1],), sd['norm.weight'], sd['norm.bias'], 1e-6)
    return F.linear(p, sd['head.weight'], sd['head.bias'])


def train_step_grads(sd, x, target, cfg, dp_masks=None, smoothing=0.0):
    """cross entropy on the single output (MAP/train.py: a non-list output takes the plain criterion)"""
    names = list(sd.keys())
    leaf = OrderedDict((n, sd[n].detach().clone().requires_grad_(True)) for n in names)
    out = forward(leaf, x, cfg, dp_masks)
    loss = F.cross_entropy(out, target, label_smoothing=smoothing)
    gs = torch.autograd.grad(loss, [leaf[n] for n in names])
    return loss.detach(), out.detach(), OrderedDict(zip(names, gs))
