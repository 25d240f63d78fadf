"""GPU: the MAPHead options of /root/reference/MAP/models/map.py beyond the map_convnext defaults:

  split   SplitNormHead (map.py:415-441; map_convnext.ConvNeXt(split_norm=True))
  nosdt   self_distill_token = False: no mean token, no self_dt_heads, plain logits in train mode (map.py:273-275,490-491,536-537)
  linear  head_fn = nn.Linear + no self-distillation token + ONE group (the head options of map_mobilenet_v1)
  inter   ClassAttention(interactive=True): head-mixing linears w1 / w2 around the softmax (map.py:96-98,130-136)
  mismatch gram_dim != last_dim: the dim_mismatch CABlock / ClassAttention (map.py:85-90,101-116,165-177): class rows of width
          gram_dim with their own q / k1 / v1 and norm1_1, image rows k2 / v2 and norm1_2, attention output replaces the class rows

against tests/golden/mapvar_*.npz, written by oracle/gen_golden_map_variants.py from the REAL reference classes (logits, loss,
top-5, per-parameter gradient norms), and against the oracle restatement for every gradient tensor.
fp32 mode: logits / loss 1e-3, gradients 2e-2, top-5 bit-exact; bf16 mode: logits 6e-2, loss 2e-2, whole-tensor gradient gates.
"""
import json
import os

import numpy as np
import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
TAGS = ['split', 'nosdt', 'linear', 'inter', 'mismatch']


def _cfg(tag, kind):
    from oracle import map_oracle as O
    z = np.load(os.path.join(GOLDEN, f'mapvar_{tag}_{kind}.npz'))
    cfg = json.loads(str(z['cfg']))
    cfg['dims'], cfg['depths'] = tuple(cfg['dims']), tuple(cfg['depths'])
    return z, cfg, O


def _build(cfg, mode, O):
    import imagenet_models_amd as A
    m = A.MAP_ConvNeXt(num_classes=cfg['num_classes'], depths=cfg['depths'], dims=cfg['dims'], last_dim=cfg['last_dim'],
                       n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'], gram_group=cfg['gram_group'], bp_dim=cfg['bp_dim'],
                       ca_dim=cfg['ca_dim'], num_heads=cfg['num_heads'], head_drop=0.0, head_attn_drop=0.0, math_mode=mode,
                       head_fn=cfg['head_fn'], self_distill_token=cfg['self_distill_token'], interactive=cfg['interactive'],
                       gram_dim=cfg['gram_dim'])
    sd = O.fill_state(cfg)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    return m.cuda(), sd


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _flat(outs):
    f = []
    for o in outs:
        f.extend(o if isinstance(o, (list, tuple)) else [o])
    return f


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_eval_vs_reference_fixture(tag, mode, tol):
    import imagenet_models_amd as A
    z, cfg, O = _cfg(tag, 'eval')
    m, sd = _build(cfg, mode, O)
    assert sum(p.numel() for p in m.parameters()) == int(z['param_count'])
    m.eval()
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = m(x.cuda())
    e = rel(torch.stack(outs)[:, :, :40], z['logits'])
    print(f'[mapvar_{tag} {mode}] eval logits vs reference fixture {e:.2e}')
    assert len(outs) == cfg['n_groups'] and e < tol
    if mode == 'fp32':
        _, idx = A.heads_mean_topk(outs, 5)
        assert np.array_equal(idx.cpu().numpy(), z['top5'])


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('mode,tols', [('fp32', (1e-3, 1e-3, 2e-2)), ('bf16', (6e-2, 2e-2, None))])
def test_train_step_vs_reference_fixture_and_oracle(tag, mode, tols):
    import imagenet_models_amd as A
    from oracle import ga_convnext_oracle as GO
    z, cfg, O = _cfg(tag, 'train_b4')
    B = int(z['batch'])
    m, sd = _build(cfg, mode, O)
    m.train()
    x = O.gen_input(B, seed=1)
    target = torch.from_numpy(z['target'])
    m.zero_grad()
    outs = m(x.cuda())
    assert isinstance(outs[0], (list, tuple)) == bool(cfg['self_distill_token'])
    loss = A.map_loss(outs, target.cuda(), float(z['dec_lam']))
    loss.backward()
    e_out = rel(torch.stack(_flat(outs))[:, :, :40], z['logits'])
    e_loss = abs(float(loss.detach()) - float(z['loss'])) / abs(float(z['loss']))
    P = dict(m.named_parameters())
    grads = {n: p.grad.detach().cpu() for n, p in P.items()}
    gmax = float(z['grad_norm'].max())
    e_n = {n: abs(float(grads[n].double().norm()) - w) / max(w, 1e-3 * gmax) for n, w in zip(z['grad_names'].tolist(), z['grad_norm'].tolist())}
    worst_n = sorted(e_n.items(), key=lambda kv: -kv[1])[:3]
    oloss, oouts, ograds, _ = O.train_step_grads(sd, x, target, cfg, dec_lam=float(z['dec_lam']))
    errs = GO.grad_errors(grads, ograds)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
    print(f'[mapvar_{tag} {mode}] train: logits {e_out:.2e} loss {e_loss:.2e} grad norms vs reference {worst_n[0]} grads vs oracle {worst[0]}')
    assert e_out < tols[0] and e_loss < tols[1]
    if mode == 'fp32':
        assert worst_n[0][1] < tols[2] and worst[0][1] < tols[2], (worst_n, worst)
    else:
        assert_grads_close(grads, ograds, BF16_REL, BF16_COS, f'mapvar_{tag} bf16')
