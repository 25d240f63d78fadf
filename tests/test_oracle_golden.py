"""CPU: the oracle restatement (oracle/ga_convnext_oracle.py) against the golden vectors that
oracle/gen_golden.py produced from the REAL reference classes (/root/reference/GA/ga_convnext.py,
loss formula of GA/train.py:735-745).  Tolerances: logits/loss 1e-4 relative (fp32 CPU both sides),
gradients 1e-2 (the reference's own fp32 backward is 4e-3 from a float64 run), top-k indices bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import ga_convnext_oracle as O


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(str(z['cfg']))
    cfg['depths'] = tuple(cfg['depths'])
    cfg['dims'] = tuple(cfg['dims'])
    return z, cfg


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.mark.parametrize('tag', ['v2', 't768'])
def test_eval_logits_and_topk(tag):
    z, cfg = _load(f'{tag}_eval.npz')
    sd = O.fill_state(cfg)
    assert len(sd) == int(z['n_state'])
    assert sum(v.numel() for k, v in sd.items() if not O.is_buffer(k)) == int(z['param_count'])
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = O.forward(sd, x, cfg, training=False)
    nlog = z['logits'].shape[2]
    got = torch.stack(outs)[:, :, :nlog].numpy()
    assert _rel(got, z['logits']) < 1e-4
    top5 = O.topk_indices(O.validate_output(outs), 5).numpy()
    assert np.array_equal(top5, z['top5'])  # bit-exact indices


def test_param_counts_known_answers():
    # SURVEY.md F9 / BASELINE.md section 2 [probe]: parameter counts of the registered variants
    want = {'ga_convnext_tiny_768': 54354584, 'ga_convnext_tiny_688': 47821324,
            'ga_convnext_base_1024': 128839176, 'ga_convnext_small_768': 76726424}
    for name, n in want.items():
        shapes = O.state_shapes(O.make_cfg(name))
        got = sum(int(np.prod(s)) for k, s in shapes.items() if not O.is_buffer(k))
        assert got == n, name


def test_b1024_param_count_golden():
    z, cfg = _load('b1024_eval.npz')
    shapes = O.state_shapes(cfg)
    assert sum(int(np.prod(s)) for k, s in shapes.items() if not O.is_buffer(k)) == int(z['param_count'])


@pytest.mark.parametrize('name', ['v2_train_b4.npz', 'v2_train_b128.npz'])
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


def test_loss_matches_torch_formula():
    # GA/train.py:735-745 re-evaluated with torch primitives on random logits
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    outs = [torch.randn(6, 11, generator=g) for _ in range(5)]
    tgt = torch.randint(0, 11, (6,), generator=g)
    want = sum(F.cross_entropy(o, tgt) for o in outs)
    mean = sum(outs) / 5
    for o in outs:
        want = want + (-0.8) * F.kl_div(F.log_softmax(o, 1), F.log_softmax(mean, 1), reduction='mean', log_target=True)
    assert abs(float(O.ga_loss(outs, tgt, -0.8)) - float(want)) < 1e-6


def test_optimizer_restatements_match_torch_optim():
    g = torch.Generator().manual_seed(5)
    shapes = {'a.weight': (7, 5), 'a.bias': (7,), 'n.weight': (5,), 'g.gamma': (5,)}
    p0 = {n: torch.randn(s, generator=g) for n, s in shapes.items()}
    grads = [{n: torch.randn(s, generator=g) for n, s in shapes.items()} for _ in range(3)]
    # torch reference with timm's weight-decay grouping
    def groups(ps, wd):
        decay = [p for n, p in ps.items() if not O.no_weight_decay(n, p.shape)]
        nodecay = [p for n, p in ps.items() if O.no_weight_decay(n, p.shape)]
        return [{'params': nodecay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': wd}]
    ps = {n: v.clone().requires_grad_(True) for n, v in p0.items()}
    opt = torch.optim.SGD(groups(ps, 0.05), lr=0.1, momentum=0.9, nesterov=True)
    mine, bufs = {n: v.clone() for n, v in p0.items()}, {}
    for i, gr in enumerate(grads):
        for n in ps:
            ps[n].grad = gr[n].clone()
        opt.step()
        mine, bufs = O.sgd_nesterov_step(mine, gr, bufs, 0.1, 0.9, 0.05, first=(i == 0))
        for n in ps:
            assert torch.allclose(mine[n], ps[n].detach(), atol=1e-6), n
    ps = {n: v.clone().requires_grad_(True) for n, v in p0.items()}
    opt = torch.optim.AdamW(groups(ps, 0.05), lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    mine, m, v = {n: t.clone() for n, t in p0.items()}, {}, {}
    for i, gr in enumerate(grads):
        for n in ps:
            ps[n].grad = gr[n].clone()
        opt.step()
        mine, m, v = O.adamw_step(mine, gr, m, v, i + 1, 1e-2, (0.9, 0.999), 1e-8, 0.05)
        for n in ps:
            assert torch.allclose(mine[n], ps[n].detach(), atol=1e-6), n
