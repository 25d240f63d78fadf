"""GPU parity of the whole MAP-ConvNeXt path (HIP kernels through the C ABI) against the oracle restatement of
/root/reference/MAP/models/{map,map_convnext}.py + MAP/train.py:792-839 and the committed golden vectors produced from the
real reference classes (head dropout at p = 0 there; the dropout path is checked with explicit masks against the oracle).

Tolerances (north_star: 1e-3 relative fp32, bit-exact top-k):
  fp32 math mode: logits / loss 1e-3 relative to the tensor max, gradients 2e-2 under oracle.grad_errors, top-5 bit-exact;
  bf16 mode: logits 6e-2, loss 2e-2 (reported, not the parity gate)."""
import json
import os

import numpy as np
import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import map_oracle as O
    return O


def build(cfg, mode, drop_path=0.0, head_drop=0.0):
    import imagenet_models_amd as A
    m = A.MAP_ConvNeXt(num_classes=cfg['num_classes'], depths=cfg['depths'], dims=cfg['dims'], drop_path_rate=drop_path,
                       last_dim=cfg['last_dim'], n_groups=cfg['n_groups'], n_tokens=cfg['n_tokens'], gram_group=cfg['gram_group'],
                       bp_dim=cfg['bp_dim'], ca_dim=cfg['ca_dim'], num_heads=cfg['num_heads'], head_drop=head_drop,
                       head_attn_drop=head_drop, math_mode=mode)
    O = _oracle()
    sd = O.fill_state(cfg)
    m.load_state_dict(sd)
    return m.cuda(), sd


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(str(z['cfg']))
    cfg['depths'], cfg['dims'] = tuple(cfg['depths']), tuple(cfg['dims'])
    return z, cfg


def test_registry_state_dict_and_readme_param_counts():
    import imagenet_models_amd as A
    O = _oracle()
    for name, n in (('map_convnext_tiny', 47833760), ('map_convnext_small', 82837664)):     # MAP/README.MD:308,373
        assert A.is_model(name)
        m = A.create_model(name, pretrained=False, num_classes=1000, drop_path_rate=None)
        shapes = O.state_shapes(O.make_cfg(name))
        sd = m.state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)
        assert sum(p.numel() for p in m.parameters()) == n


@pytest.mark.parametrize('tag', ['map_v5', 'map_v5s'])
@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_eval_logits_topk(tag, mode, tol):
    import imagenet_models_amd as A
    O = _oracle()
    z, cfg = load_golden(f'{tag}_eval.npz')
    m, sd = build(cfg, mode)
    m.eval()
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = m(x.cuda())
        ref = O.forward(sd, x, cfg, training=False)
    assert len(outs) == cfg['n_groups'] and outs[0].shape == (2, 40) and outs[0].dtype == torch.float32
    err = max(rel(a, b) for a, b in zip(outs, ref))
    gerr = rel(torch.stack(outs), torch.from_numpy(z['logits']))
    print(f'[{tag} {mode}] eval logits rel err vs oracle {err:.3e}, vs reference golden {gerr:.3e}')
    assert err < tol and gerr < tol
    if mode == 'fp32':
        mean, idx = A.heads_mean_topk(outs, 5)
        assert np.array_equal(idx.cpu().numpy(), z['top5'])      # bit-exact vs the reference (mean of the group logits)
        assert rel(mean, O.validate_output(ref)) < 1e-3


def _train_compare(mode, golden, tol_out, tol_loss, tol_grad, drop_path=0.0, head_drop=0.0):
    import imagenet_models_amd as A
    O = _oracle()
    z, cfg = load_golden(golden)
    batch = int(z['batch'])
    cfg['drop_path_rate'] = drop_path
    m, sd = build(cfg, mode, drop_path, head_drop)
    m.train()
    x = O.gen_input(batch, seed=1)
    target = torch.from_numpy(z['target'])
    lam = float(z['dec_lam'])
    dp_masks = drop_masks = None
    eng = m.engine(batch, True)
    g = torch.Generator().manual_seed(5)
    if drop_path > 0:
        dp_masks = {}
        for site in eng.dp_scale:
            keep = 1 - eng.dp_rates[site]
            dp_masks[site] = (torch.rand(batch, generator=g) < keep).float() / keep
        eng.set_drop_path_masks(dp_masks)
        eng.fixed_masks = True
    if head_drop > 0:
        keep = 1 - head_drop
        L, Tq, nh = cfg['last_dim'], cfg['n_tokens'] + 1, cfg['num_heads']
        N = Tq + 196

        def bern(*shape):
            return (torch.rand(*shape, generator=g) < keep).float() / keep
        drop_masks = {k: dict(attn=bern(batch, nh, Tq, N), proj=bern(batch, Tq, L), mlp=bern(batch, Tq, 4 * L))
                      for k in range(cfg['n_groups'])}
        eng.set_dropout_masks(drop_masks)
        eng.fixed_masks = True
    m.zero_grad()
    outs = m(x.cuda())
    assert isinstance(outs[0], list) and len(outs[0]) == 2
    loss = A.map_loss(outs, target.cuda(), lam)
    loss.backward()
    oloss, oouts, ograds, ostats = O.train_step_grads(sd, x, target, cfg, dec_lam=lam, dp_masks=dp_masks, drop_masks=drop_masks)
    e_out = max(max(rel(a, b) for a, b in zip(o, oo)) for o, oo in zip(outs, oouts))
    e_loss = abs(float(loss.detach()) - float(oloss)) / abs(float(oloss))
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    errs = O.grad_errors(grads, ograds)
    if mode == 'bf16':
        gmax = max(float(g_.abs().max()) for g_ in ograds.values())
        errs = {n: e for n, e in errs.items() if float(ograds[n].abs().max()) >= 1e-4 * gmax}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    new_sd = m.state_dict()
    e_bn = max(rel(new_sd[n], ostats[n].float()) for n in ostats if not n.endswith('num_batches_tracked'))
    print(f'[{golden} {mode} dp={drop_path} drop={head_drop}] logits {e_out:.2e} loss {e_loss:.2e} bn {e_bn:.2e} worst grads {worst}')
    assert e_out < tol_out and e_loss < tol_loss
    if mode == 'bf16':      # whole-tensor gates (tests/_gradcheck.py): norm-relative error and direction of every gradient
        assert_grads_close(grads, ograds, BF16_REL, BF16_COS, f'{golden} bf16')
    else:
        assert worst[0][1] < tol_grad, worst
    assert e_bn < max(tol_out, 2e-3)
    return z, outs, loss, grads


@pytest.mark.parametrize('golden', ['map_v5_train_b4.npz', 'map_v5s_train_b4.npz'])
def test_train_step_fp32_vs_oracle_and_reference(golden):
    z, outs, loss, grads = _train_compare('fp32', golden, 1e-3, 1e-3, 2e-2)
    assert rel(torch.stack([o[0] for o in outs])[:, :, :40], torch.from_numpy(z['org'])) < 1e-3
    assert rel(torch.stack([o[1] for o in outs])[:, :, :40], torch.from_numpy(z['avg'])) < 1e-3
    assert abs(float(loss.detach()) - float(z['loss'])) / abs(float(z['loss'])) < 1e-3
    names = [str(n) for n in z['grad_names']]
    gmax = float(np.abs(z['grad_head']).max())
    for i, n in enumerate(names):
        ref_norm = float(z['grad_norm'][i])
        if ref_norm > 1e-3 * gmax:
            assert abs(float(grads[n].double().norm()) - ref_norm) / ref_norm < 2e-2, n


def test_train_step_fp32_with_drop_path_and_dropout_masks():
    """DropPath (trunk) and the head's nn.Dropout sites (attention probabilities, projection, MLP hidden) with injected masks"""
    _train_compare('fp32', 'map_v5_train_b4.npz', 1e-3, 1e-3, 2e-2, drop_path=0.3, head_drop=0.1)


def test_train_step_bf16_reported():
    _train_compare('bf16', 'map_v5_train_b4.npz', 6e-2, 2e-2, 1.0)   # ReLU masks of the 3-token MLP flip under bf16 at B=4: reported only


def test_dropout_sampler_runs_in_training():
    """default head dropout (0.05, map.py:149,464): the mask buffer is refilled every step; loss stays finite"""
    import imagenet_models_amd as A
    O = _oracle()
    z, cfg = load_golden('map_v5_train_b4.npz')
    m, sd = build(cfg, 'fp32', 0.0, 0.05)
    m.train()
    x = O.gen_input(4, seed=1).cuda()
    eng = m.engine(4, True)
    l1 = A.map_loss(m(x), torch.from_numpy(z['target']).cuda(), -0.8)
    a = eng.drop['buf'].clone()
    l2 = A.map_loss(m(x), torch.from_numpy(z['target']).cuda(), -0.8)
    assert not torch.equal(a, eng.drop['buf']) and torch.isfinite(l1) and torch.isfinite(l2)
    keep = float((a > 0).float().mean())
    assert abs(keep - 0.95) < 0.01


@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_tiny_eval_vs_reference_golden(mode, tol):
    import imagenet_models_amd as A
    O = _oracle()
    z, cfg = load_golden('map_tiny_eval.npz')
    m, sd = build(cfg, mode)
    m.eval()
    x = O.gen_input(int(z['batch']), seed=0)
    with torch.no_grad():
        outs = m(x.cuda())
    gerr = rel(torch.stack(outs)[:, :, :16], torch.from_numpy(z['logits']))
    print(f'[map tiny {mode}] eval logits rel err vs reference golden {gerr:.3e}')
    assert gerr < tol
    if mode == 'fp32':
        _, idx = A.heads_mean_topk(outs, 5)
        assert np.array_equal(idx.cpu().numpy(), z['top5'])


@pytest.mark.parametrize('mode,tols', [('fp32', (1e-3, 1e-3, 2e-2)), ('bf16', (6e-2, 2e-2, 0.6))])
def test_tiny_train_step_vs_oracle(mode, tols):
    z, outs, loss, grads = _train_compare(mode, 'map_tiny_train_b4.npz', *tols)
    if mode == 'fp32':
        assert rel(torch.stack([o[0] for o in outs])[:, :, :40], torch.from_numpy(z['org'])) < 1e-3
        assert abs(float(loss.detach()) - float(z['loss'])) / abs(float(z['loss'])) < 1e-3
