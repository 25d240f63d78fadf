"""GPU parity of map_pit_s (/root/reference/MAP/models/map_pit.py, PoolingTransformer with pool_type='map') through the C ABI:
  * against tests/golden/pit_*.npz, written by oracle/gen_golden_pit.py from the REAL reference classes (logits, top-5, loss,
    per-parameter gradient norms);
  * against the oracle restatement (oracle/map_pit_oracle.py, itself checked against the reference when the fixtures were made):
    every logit, the loss, every gradient, the BatchNorm running statistics, with and without DropPath.
fp32 mode: logits / loss 1e-3, gradients 2e-2; bf16 mode 6e-2 / 2e-2 (reported)."""
import json
import os

import numpy as np
import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _oracle():
    from oracle import map_pit_oracle as O
    return O


def build(cfg, mode, drop_path=0.0):
    import imagenet_models_amd as A
    O = _oracle()
    m = A.MAP_PiT(image_size=cfg['image_size'], patch_size=cfg['patch_size'], stride=cfg['stride'], base_dims=cfg['base_dims'],
                  depth=cfg['depth'], heads=cfg['heads'], num_classes=cfg['num_classes'], drop_path_rate=drop_path,
                  last_dim=cfg['last_dim'], n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'], gram_group=cfg['gram_group'],
                  head_drop=0.0, head_attn_drop=0.0, math_mode=mode)
    sd = O.fill_state(cfg)
    m.load_state_dict(sd)
    return m.cuda(), sd


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _gold(tag):
    z = np.load(os.path.join(GOLD, tag + '.npz'))
    cfg = json.loads(str(z['cfg']))
    O = _oracle()
    return z, O.make_cfg(**{k: cfg[k] for k in ('image_size', 'patch_size', 'stride', 'base_dims', 'depth', 'heads', 'num_classes',
                                                  'last_dim', 'n_groups', 'n_tokens', 'gram_group')})


@pytest.mark.parametrize('tag', ['pit_v8_eval', 'pit_s_eval'])
@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_eval_against_reference_fixture(tag, mode, tol):
    O = _oracle()
    z, cfg = _gold(tag)
    m, sd = build(cfg, mode)
    assert sum(p.numel() for p in m.parameters()) == int(z['param_count'])
    m.eval()
    x = O.gen_input(int(z['batch']), seed=0, size=cfg['image_size'])
    with torch.no_grad():
        outs = m(x.cuda())
    got = torch.stack([o.float().cpu() for o in outs])
    e = rel(got[:, :, :40], z['logits'])
    print(f'[{tag} {mode}] eval logits vs reference fixture: {e:.3e}')
    assert e < tol
    if mode == 'fp32':
        assert np.array_equal((got.mean(0)).topk(5, 1, True, True)[1].numpy(), z['top5'])


@pytest.mark.parametrize('mode,tols', [('fp32', (1e-3, 1e-3, 2e-2)), ('bf16', (6e-2, 2e-2, 0.25))])
def test_train_step_against_reference_fixture(mode, tols):
    import imagenet_models_amd as A
    O = _oracle()
    z, cfg = _gold('pit_v8_train_b4')
    B = int(z['batch'])
    m, sd = build(cfg, mode)
    m.train()
    x = O.gen_input(B, seed=1, size=cfg['image_size'])
    target = torch.from_numpy(z['target'])
    m.zero_grad()
    outs = m(x.cuda())
    loss = A.map_loss(outs, target.cuda(), float(z['dec_lam']))
    loss.backward()
    org, avg = torch.stack([o[0].float().cpu() for o in outs]), torch.stack([o[1].float().cpu() for o in outs])
    e_out = max(rel(org[:, :, :40], z['org']), rel(avg[:, :, :40], z['avg']))
    e_loss = abs(float(loss.detach()) - float(z['loss'])) / abs(float(z['loss']))
    grads = dict(m.named_parameters())
    norms = {n: float(grads[n].grad.double().norm()) for n in z['grad_names'].tolist()}
    gmax = float(z['grad_norm'].max())
    e_g = {n: abs(norms[n] - w) / max(w, 1e-3 * gmax) for n, w in zip(z['grad_names'].tolist(), z['grad_norm'].tolist())}
    worst = sorted(e_g.items(), key=lambda kv: -kv[1])[:4]
    print(f'[pit_v8 {mode}] vs reference fixture: logits {e_out:.2e} loss {e_loss:.2e} worst grad norms {worst}')
    assert e_out < tols[0] and e_loss < tols[1] and worst[0][1] < tols[2], worst      # (the fixture holds gradient NORMS: a whole-tensor measure)


V8 = dict(image_size=64, patch_size=16, stride=8, base_dims=(48, 48, 48), depth=(1, 2, 1), heads=(1, 2, 4), num_classes=40, last_dim=64,
          n_groups=2, n_tokens=4, gram_group=8)
V8B = dict(image_size=96, patch_size=16, stride=8, base_dims=(32, 32, 32), depth=(2, 2, 2), heads=(2, 4, 8), num_classes=24, last_dim=64,
           n_groups=3, n_tokens=2, gram_group=4)


@pytest.mark.parametrize('over', [V8, V8B])
@pytest.mark.parametrize('mode,tols,dp', [('fp32', (1e-3, 1e-3, 2e-2), 0.0), ('fp32', (1e-3, 1e-3, 2e-2), 0.3), ('bf16', (6e-2, 2e-2, 1.0), 0.0)])
def test_train_step_against_oracle(over, mode, tols, dp):
    import imagenet_models_amd as A
    O = _oracle()
    cfg = O.make_cfg(**over)
    cfg['drop_path_rate'] = dp
    B = 4
    m, sd = build(cfg, mode, dp)
    m.train()
    x = O.gen_input(B, seed=1, size=cfg['image_size'])
    target = torch.randint(0, cfg['num_classes'], (B,), generator=torch.Generator().manual_seed(5))
    omasks = None
    if dp > 0:
        eng = m.engine(B, True)
        g = torch.Generator().manual_seed(5)
        masks, omasks = {}, {}
        for site in eng.dp_scale:
            keep = 1 - eng.dp_rates[site]
            masks[site] = (torch.rand(B, generator=g) < keep).float() / keep
        for site in masks:
            pre = site[:-2]
            omasks[pre] = (masks.get(pre + '#1'), masks.get(pre + '#2'))
        eng.set_drop_path_masks(masks)
        eng.fixed_masks = True
    m.zero_grad()
    outs = m(x.cuda())
    loss = A.map_loss(outs, target.cuda(), -0.8)
    loss.backward()
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, dec_lam=-0.8, dp_masks=omasks)
    e_out = max(max(rel(a, b) for a, b in zip(o, oo)) for o, oo in zip(outs, oouts))
    e_loss = abs(float(loss.detach()) - float(oloss)) / abs(float(oloss))
    from oracle import ga_convnext_oracle as GO
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    errs = GO.grad_errors(grads, ograds)
    if mode == 'bf16':
        gmax = max(float(g_.abs().max()) for g_ in ograds.values())
        errs = {n: e for n, e in errs.items() if float(ograds[n].abs().max()) >= 1e-4 * gmax}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    new_sd = m.state_dict()
    e_bn = max(rel(new_sd[n], ostats[n].float()) for n in ostats if not n.endswith('num_batches_tracked'))
    print(f'[map_pit {over["image_size"]} {mode} dp={dp}] logits {e_out:.2e} loss {e_loss:.2e} bn {e_bn:.2e} worst grads {worst}')
    assert e_out < tols[0] and e_loss < tols[1]
    if mode == 'bf16':      # whole-tensor gates (tests/_gradcheck.py): norm-relative error and direction of every gradient
        assert_grads_close(grads, ograds, BF16_REL, BF16_COS, 'bf16 train step')
    else:
        assert worst[0][1] < tols[2], worst
    assert e_bn < max(tols[0], 2e-3)


def test_registry_and_param_layout():
    import imagenet_models_amd as A
    O = _oracle()
    m = A.create_model('map_pit_s')
    shapes = O.state_shapes(O.make_cfg('map_pit_s'))
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
    assert sum(p.numel() for p in m.parameters()) == 36147424
