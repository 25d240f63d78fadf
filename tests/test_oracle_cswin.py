"""CPU: the GA-CSWin oracle restatement (oracle/ga_cswin_oracle.py) against the golden vectors that
oracle/gen_golden_cswin.py produced from the REAL reference classes of /root/reference/GA/ga_cswin.py
(LePEAttention :59-136, CSWinBlock :139-212, GA_CSWinTransformer :447-693; loss formula GA/train.py:735-745).
Tolerances: outputs / loss 1e-4 relative (fp32 CPU on both sides), gradients 1e-2, top-k indices bit-exact.
The full-size configuration ("tiny") is the survey's candidate: its arithmetic is pinned, the config is not."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ga_cswin_oracle as O


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(str(z['cfg']))
    for k in ('depth', 'split_size', 'num_heads', 'dims'):
        cfg[k] = tuple(cfg[k])
    return z, cfg


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize('tag', ['cswin_v6', 'cswin_v6b', 'cswin_tiny'])
def test_eval_logits_and_topk(tag):
    z, cfg = _load(f'{tag}_eval.npz')
    sd = O.fill_state(cfg)
    assert len(sd) == int(z['n_state'])
    assert sum(v.numel() for k, v in sd.items() if not O.is_buffer(k)) == int(z['param_count'])
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = O.forward(sd, x, cfg, training=False)
    nlog = z['logits'].shape[2]
    assert _rel(torch.stack(outs)[:, :, :nlog].numpy(), z['logits']) < 1e-4
    assert np.array_equal(O.topk_indices(O.validate_output(outs), 5).numpy(), z['top5'])


def test_candidate_tiny_param_count():
    # SURVEY.md F3 [probe]: 41.86 M parameters for the candidate configuration (README: 42.0 M)
    shapes = O.state_shapes(O.make_cfg('ga_CSWin_64_12211_tiny_224'))
    assert sum(int(np.prod(s)) for k, s in shapes.items() if not O.is_buffer(k)) == 41858952


@pytest.mark.parametrize('name', ['cswin_v6_train_b4.npz', 'cswin_v6b_train_b4.npz'])
def test_train_step_against_reference(name):
    z, cfg = _load(name)
    sd = O.fill_state(cfg)
    b = int(z['batch'])
    x = O.gen_input(b, seed=1)
    target = torch.from_numpy(z['target'])
    loss, outs, grads, stats = O.train_step_grads(sd, x, target, cfg, lam=float(z['lam']))
    assert abs(float(loss) - float(z['loss'])) / abs(float(z['loss'])) < 1e-4
    assert _rel(torch.stack(outs)[:, :, :40].numpy(), z['logits']) < 1e-4
    names = [str(n) for n in z['grad_names']]
    assert names == list(grads.keys())
    gmax = float(np.abs(z['grad_head']).max())
    for i, n in enumerate(names):
        g = grads[n]
        ref_norm = float(z['grad_norm'][i])
        if ref_norm > 1e-3 * gmax:
            assert abs(float(g.double().norm()) - ref_norm) / ref_norm < 1e-2, n
        head = g.reshape(-1)[:16].numpy()
        ref_head = z['grad_head'][i][:head.size]
        assert np.abs(head - ref_head).max() <= 1e-2 * max(np.abs(ref_head).max(), 1e-2 * gmax), n
    for i, n in enumerate([str(s) for s in z['bn_names']]):
        got = stats[n].reshape(-1)[:8].numpy()
        assert np.abs(got - z['bn_head'][i]).max() < 1e-4 * max(1.0, np.abs(z['bn_head'][i]).max()), n


def module_input(shape, seed):
    g = torch.Generator().manual_seed(4321 + seed)
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize('name', ['lepe_v', 'lepe_h', 'lepe_full', 'lepe_s1', 'lepe_s2h'])
def test_lepe_attention_module_vectors(name):
    z = np.load(os.path.join(GOLDEN, 'cswin_modules.npz'))
    reso, idx, split, dim, heads = [int(v) for v in z[f'{name}.cfg']]
    qkv = module_input((3, 2, reso * reso, dim), seed=len(name)).requires_grad_(True)
    w = torch.from_numpy(z[f'{name}.w']).requires_grad_(True)
    b = torch.from_numpy(z[f'{name}.b']).requires_grad_(True)
    y = O.lepe_attention(qkv[0], qkv[1], qkv[2], w, b, reso, idx, split, heads)
    assert _rel(y.detach().numpy(), z[f'{name}.y']) < 1e-5
    y.backward(module_input(tuple(y.shape), seed=100 + len(name)))
    assert _rel(qkv.grad.numpy(), z[f'{name}.dqkv']) < 1e-4
    assert _rel(w.grad.numpy(), z[f'{name}.dw']) < 1e-4
    assert _rel(b.grad.numpy(), z[f'{name}.db']) < 1e-4
