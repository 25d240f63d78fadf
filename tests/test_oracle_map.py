"""CPU: the MAP oracle restatement (oracle/map_oracle.py) against the golden vectors that oracle/gen_golden_map.py produced
from the REAL reference classes (/root/reference/MAP/models/map.py:43-539, map_convnext.py:14-170; loss arithmetic of
MAP/train.py:792-839) with the head's nn.Dropout modules at p = 0, and the README's parameter-count known answers.
Tolerances: outputs / loss 1e-4 relative (fp32 CPU on both sides), gradients 1e-2, top-k indices bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import map_oracle as O


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(str(z['cfg']))
    cfg['depths'], cfg['dims'] = tuple(cfg['depths']), tuple(cfg['dims'])
    return z, cfg


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_readme_param_counts():
    # MAP/README.MD:308,373 (validation transcripts of the release checkpoints)
    for name, n in (('map_convnext_tiny', 47833760), ('map_convnext_small', 82837664)):
        shapes = O.state_shapes(O.make_cfg(name))
        assert sum(int(np.prod(s)) for k, s in shapes.items() if O.is_param(k)) == n, name


@pytest.mark.parametrize('tag', ['map_v5', 'map_v5s', 'map_tiny'])
def test_eval_logits_and_topk(tag):
    z, cfg = _load(f'{tag}_eval.npz')
    sd = O.fill_state(cfg)
    assert len(sd) == int(z['n_state'])
    assert sum(v.numel() for k, v in sd.items() if O.is_param(k)) == int(z['param_count'])
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = O.forward(sd, x, cfg, training=False)
    nlog = z['logits'].shape[2]
    assert _rel(torch.stack(outs)[:, :, :nlog].numpy(), z['logits']) < 1e-4
    assert np.array_equal(O.topk_indices(O.validate_output(outs), 5).numpy(), z['top5'])


@pytest.mark.parametrize('name', ['map_v5_train_b4.npz', 'map_v5s_train_b4.npz'])
def test_train_step_against_reference(name):
    z, cfg = _load(name)
    sd = O.fill_state(cfg)
    x = O.gen_input(int(z['batch']), seed=1)
    target = torch.from_numpy(z['target'])
    loss, outs, grads, stats = O.train_step_grads(sd, x, target, cfg, dec_lam=float(z['dec_lam']))
    assert abs(float(loss) - float(z['loss'])) / abs(float(z['loss'])) < 1e-4
    assert _rel(torch.stack([o[0] for o in outs])[:, :, :40].numpy(), z['org']) < 1e-4
    assert _rel(torch.stack([o[1] for o in outs])[:, :, :40].numpy(), z['avg']) < 1e-4
    names = [str(n) for n in z['grad_names']]
    assert names == list(grads.keys())
    gmax = float(np.abs(z['grad_head']).max())
    for i, n in enumerate(names):
        ref_norm = float(z['grad_norm'][i])
        if ref_norm > 1e-3 * gmax:
            assert abs(float(grads[n].double().norm()) - ref_norm) / ref_norm < 1e-2, n
    for i, n in enumerate([str(s) for s in z['bn_names']]):
        got = stats[n].reshape(-1)[:8].numpy()
        assert np.abs(got - z['bn_head'][i]).max() < 1e-4 * max(1.0, np.abs(z['bn_head'][i]).max()), n


def test_multi_group_loss_matches_torch_formula():
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    outs = [[torch.randn(6, 11, generator=g), torch.randn(6, 11, generator=g)] for _ in range(4)]
    tgt = torch.randint(0, 11, (6,), generator=g)
    want = 0
    agg = sum(o[0] for o in outs)
    for y, ym in outs:
        want = want + F.cross_entropy(y, tgt) + F.kl_div(F.log_softmax(ym, 1), F.log_softmax(y, 1), reduction='sum',
                                                         log_target=True) / y.numel()
    for y, ym in outs:
        want = want + (-0.8) * F.kl_div(F.log_softmax(y, 1), F.log_softmax(agg / 4, 1), reduction='mean', log_target=True)
    assert abs(float(O.multi_group_loss(outs, tgt, -0.8)) - float(want)) < 1e-6


# ---- plain ConvNeXt (global_pool='avg' branch of map_convnext.py; oracle/convnext_oracle.py vs fixtures from the reference class)
@pytest.mark.parametrize('tag', ['cnx_v9', 'cnx_tiny'])
def test_plain_convnext_eval_fixture(tag):
    from oracle import convnext_oracle as CO
    z = np.load(os.path.join(GOLDEN, f'{tag}_eval.npz'))
    c = json.loads(str(z['cfg']))
    cfg = CO.make_cfg(dims=tuple(c['dims']), depths=tuple(c['depths']), num_classes=c['num_classes'])
    sd = CO.fill_state(cfg)
    assert sum(v.numel() for v in sd.values()) == int(z['param_count'])
    with torch.no_grad():
        out = CO.forward(sd, CO.gen_input(int(z['batch']), seed=0), cfg)
    assert _rel(out[:, :40].numpy(), z['logits']) < 1e-4
    assert np.array_equal(out.topk(5, 1, True, True)[1].numpy(), z['top5'])


def test_plain_convnext_train_fixture():
    from oracle import convnext_oracle as CO
    z = np.load(os.path.join(GOLDEN, 'cnx_v9_train_b4.npz'))
    c = json.loads(str(z['cfg']))
    cfg = CO.make_cfg(dims=tuple(c['dims']), depths=tuple(c['depths']), num_classes=c['num_classes'])
    sd = CO.fill_state(cfg)
    loss, out, grads = CO.train_step_grads(sd, CO.gen_input(int(z['batch']), seed=1), torch.from_numpy(z['target']), cfg)
    assert abs(float(loss) - float(z['loss'])) / abs(float(z['loss'])) < 1e-4
    assert _rel(out[:, :40].numpy(), z['logits']) < 1e-4
    gmax = float(z['grad_norm'].max())
    for n, w in zip(z['grad_names'].tolist(), z['grad_norm'].tolist()):
        if w > 1e-3 * gmax:
            assert abs(float(grads[n].double().norm()) - w) / w < 1e-2, n
    # README known answers of the architecture: ConvNeXt-T / -S parameter counts
    assert sum(int(np.prod(s)) for s in CO.state_shapes(CO.make_cfg('convnext_tiny')).values()) == 28589128
