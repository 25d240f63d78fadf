"""MAPHead variants of /root/reference/MAP/models/map.py from the REAL reference classes (build container only):

  split   head_fn = SplitNormHead           (map_convnext.ConvNeXt(split_norm=True), map_convnext.py:97-98)
  nosdt   self_distill_token = False        (ConvNeXt(self_distill_token=False); MAPHead then returns plain logits, map.py:536-537)
  linear  head_fn = nn.Linear, no self-distillation token, ONE group  (the map_mobilenet_v1 head options, map_mobilenet.py:67-83)
  inter   interactive = True                (ClassAttention's head-mixing linears w1 / w2, map.py:96-98,130-136; map_resnet50 /
                                             map_faster_vit_3_224 use it)
  mismatch gram_dim != last_dim             (CABlock / ClassAttention dim_mismatch, map.py:85-90,101-116,165-177: class rows of
                                             width gram_dim, own q / k1 / v1 and norm1_1, image rows k2 / v2 and norm1_2, the
                                             attention output replaces the class rows)

split / nosdt are constructor arguments of the reference's ConvNeXt.  `linear` and `inter` are MAPHead arguments the reference's
ConvNeXt does not expose: the fixture model is the reference ConvNeXt with its `.head` replaced by a reference MAPHead built
with the same arguments plus the variant (so every class in the fixture is the reference's own).  Dropouts set to p = 0.

Run:  python oracle/gen_golden_map_variants.py        -> tests/golden/mapvar_{split,nosdt,linear,inter,mismatch}_{eval,train_b4}.npz"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'timm_stub'))
sys.path.insert(0, '/root/reference/MAP/models')
sys.path.insert(0, os.path.dirname(HERE))

import map as refmap  # noqa: E402
import map_convnext as ref  # noqa: E402
from oracle import map_oracle as O  # noqa: E402
from oracle.gen_golden import grad_stats, rel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
BASE = dict(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), last_dim=64, n_groups=2, n_tokens=2, gram_group=8, bp_dim=64, ca_dim=64,
            num_heads=8, num_classes=40)
VARIANTS = {
    'split': dict(head_fn='split'),
    'nosdt': dict(self_distill_token=False),
    'linear': dict(head_fn='linear', self_distill_token=False, n_groups=1),
    'inter': dict(interactive=True),
    'mismatch': dict(gram_dim=96),     # (32 puts one ReLU unit of the head MLP on its kink: an ill-conditioned test point)
}


def build_ref(cfg):
    m = ref.ConvNeXt(num_classes=cfg['num_classes'], depths=list(cfg['depths']), dims=list(cfg['dims']), drop_path_rate=0.0,
                     global_pool='mmcap', last_dim=cfg['last_dim'], n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'],
                     gram_group=cfg['gram_group'], bp_dim=cfg['bp_dim'], bp_groups=cfg['bp_groups'], ca_dim=cfg['ca_dim'],
                     num_heads=cfg['num_heads'], split_norm=cfg['head_fn'] == 'split', self_distill_token=cfg['self_distill_token'])
    if cfg['head_fn'] == 'linear' or cfg['interactive'] or cfg['gram_dim'] != cfg['last_dim']:
        dims = list(cfg['dims'])
        head_fn = {'norm': refmap.NormHead, 'split': refmap.SplitNormHead, 'linear': nn.Linear}[cfg['head_fn']]
        m.head = refmap.MAPHead(multi_scale_level=3, channels=[dims[0]] + dims, last_dim=cfg['last_dim'], n_tokens=cfg['n_tokens'],
                                n_groups=cfg['n_groups'], self_distill_token=cfg['self_distill_token'], mlp_ratio=4, mlp_groups=2,
                                head_fn=head_fn, fc_drop=0, num_classes=cfg['num_classes'], non_linearity=nn.GELU, gram=True,
                                bp_dim=cfg['bp_dim'], bp_groups=cfg['bp_groups'], gram_group=cfg['gram_group'],
                                gram_dim=cfg['gram_dim'] if cfg['gram_dim'] != cfg['last_dim'] else None,
                                concat_blk=None, gram_blk=nn.Identity, ca_dim=cfg['ca_dim'], num_heads=cfg['num_heads'],
                                interactive=cfg['interactive'])
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    sd = O.fill_state(cfg)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), 'state_dict key order differs from the reference:\n' + '\n'.join(
        f'{a} | {b}' for a, b in zip(ref_sd.keys(), sd.keys()) if a != b)
    for k in sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, tuple(ref_sd[k].shape), tuple(sd[k].shape))
    m.load_state_dict(sd)
    return m, sd


def ref_loss(outputs, target, dec_lam):
    """MAP/train.py:792-839 (distill_tokens == 0), both output forms"""
    loss, agg = 0, 0
    for o in outputs:
        if isinstance(o, (list, tuple)):
            y, ym = o
            agg = agg + y
            loss = loss + F.cross_entropy(y, target) + F.kl_div(F.log_softmax(ym, dim=1), F.log_softmax(y, dim=1).detach(), reduction='sum',
                                                                  log_target=True) / y.numel()
        else:
            agg = agg + o
            loss = loss + F.cross_entropy(o, target)
    if len(outputs) > 1:
        for o in outputs:
            y = o[0] if isinstance(o, (list, tuple)) else o
            loss = loss + F.kl_div(F.log_softmax(y, dim=1), F.log_softmax(agg.detach() / len(outputs), dim=1), reduction='mean',
                                   log_target=True) * dec_lam
    return loss


def flat(outs):
    f = []
    for o in outs:
        f.extend(o if isinstance(o, (list, tuple)) else [o])
    return f


def run(tag, cfg):
    m, sd = build_ref(cfg)
    m.eval()
    x = O.gen_input(2, seed=0)
    with torch.no_grad():
        outs = m(x)
        mine = O.forward(sd, x, cfg, training=False)
    err = max(rel(a, b) for a, b in zip(mine, outs))
    assert err < 1e-4, err
    s = sum(outs) / len(outs)
    np.savez_compressed(os.path.join(OUT, f'mapvar_{tag}_eval.npz'), cfg=json.dumps(cfg), batch=2, n_state=len(sd),
                        param_count=sum(p.numel() for p in m.parameters()), logits=torch.stack(outs)[:, :, :40].numpy(),
                        top5=s.topk(5, 1, True, True)[1].numpy())
    m, sd = build_ref(cfg)
    m.train()
    B = 4
    x = O.gen_input(B, seed=1)
    target = torch.randint(0, cfg['num_classes'], (B,), generator=torch.Generator().manual_seed(99))
    outs = m(x)
    loss = ref_loss(outs, target, -0.8)
    loss.backward()
    grads = {n: p.grad.detach() for n, p in m.named_parameters()}
    oloss, oouts, ograds, _ = O.train_step_grads(sd, x, target, cfg, dec_lam=-0.8)
    e_out = max(rel(a, b.detach()) for a, b in zip(flat(oouts), flat(outs)))
    e_loss = abs(float(oloss) - float(loss.detach())) / abs(float(loss.detach()))
    e_g = max(O.grad_errors(ograds, grads).values())
    print(f'[mapvar_{tag}] eval {err:.2e}; train B={B}: oracle vs reference logits {e_out:.2e} loss {e_loss:.2e} grads {e_g:.2e}; '
          f'{len(grads)} parameters')
    assert max(e_out, e_loss) < 1e-4 and e_g < 1e-2
    names, norm, ssum, head = grad_stats(grads)
    np.savez_compressed(os.path.join(OUT, f'mapvar_{tag}_train_b4.npz'), cfg=json.dumps(cfg), batch=B, dec_lam=-0.8, target=target.numpy(),
                        loss=float(loss), logits=torch.stack([o.detach() for o in flat(outs)])[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=norm, grad_sum=ssum, grad_head=head)


if __name__ == '__main__':
    torch.manual_seed(0)
    only = sys.argv[1:]
    for tag, over in VARIANTS.items():
        if only and tag not in only:
            continue
        run(tag, O.make_cfg(**dict(BASE, **over)))
