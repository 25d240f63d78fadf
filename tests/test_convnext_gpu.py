"""GPU parity of the plain ConvNeXt (global_pool='avg' branch of /root/reference/MAP/models/map_convnext.py; registered as
convnext_tiny / convnext_small) through the C ABI: against fixtures written from the REAL reference class
(tests/golden/cnx_*.npz) and against the oracle restatement (every gradient, with DropPath).  fp32 mode 1e-3 / 2e-2."""
import json
import os

import numpy as np
import pytest
import torch

from _gradcheck import assert_grads_close, BF16_REL, BF16_COS
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _oracle():
    from oracle import convnext_oracle as O
    return O


def build(cfg, mode, dp=0.0):
    import imagenet_models_amd as A
    O = _oracle()
    m = A.ConvNeXt(num_classes=cfg['num_classes'], depths=cfg['depths'], dims=cfg['dims'], drop_path_rate=dp, math_mode=mode)
    sd = O.fill_state(cfg)
    m.load_state_dict(sd)
    return m.cuda(), sd


def rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _gold(tag):
    z = np.load(os.path.join(GOLD, tag + '.npz'))
    c = json.loads(str(z['cfg']))
    return z, _oracle().make_cfg(dims=tuple(c['dims']), depths=tuple(c['depths']), num_classes=c['num_classes'])


@pytest.mark.parametrize('tag', ['cnx_v9_eval', 'cnx_tiny_eval'])
@pytest.mark.parametrize('mode,tol', [('fp32', 1e-3), ('bf16', 6e-2)])
def test_eval_against_reference_fixture(tag, mode, tol):
    O = _oracle()
    z, cfg = _gold(tag)
    m, sd = build(cfg, mode)
    assert sum(p.numel() for p in m.parameters()) == int(z['param_count'])
    m.eval()
    with torch.no_grad():
        out = m(O.gen_input(int(z['batch']), seed=0).cuda())
    assert isinstance(out, torch.Tensor) and out.shape == (int(z['batch']), cfg['num_classes'])
    e = rel(out[:, :40], z['logits'])
    print(f'[{tag} {mode}] eval logits vs reference fixture: {e:.3e}')
    assert e < tol
    if mode == 'fp32':
        assert np.array_equal(out.float().cpu().topk(5, 1, True, True)[1].numpy(), z['top5'])


@pytest.mark.parametrize('mode,tols', [('fp32', (1e-3, 1e-3, 2e-2)), ('bf16', (6e-2, 2e-2, 0.25))])
def test_train_step_against_reference_fixture(mode, tols):
    O = _oracle()
    z, cfg = _gold('cnx_v9_train_b4')
    m, sd = build(cfg, mode)
    m.train()
    x = O.gen_input(int(z['batch']), seed=1)
    m.zero_grad()
    out = m(x.cuda())
    loss = F.cross_entropy(out, torch.from_numpy(z['target']).cuda())
    loss.backward()
    e_out = rel(out[:, :40], z['logits'])
    e_loss = abs(float(loss) - float(z['loss'])) / abs(float(z['loss']))
    P = dict(m.named_parameters())
    gmax = float(z['grad_norm'].max())
    e_g = {n: abs(float(P[n].grad.double().norm()) - w) / max(w, 1e-3 * gmax) for n, w in zip(z['grad_names'].tolist(), z['grad_norm'].tolist())}
    worst = sorted(e_g.items(), key=lambda kv: -kv[1])[:4]
    print(f'[cnx_v9 {mode}] vs reference fixture: logits {e_out:.2e} loss {e_loss:.2e} worst grad norms {worst}')
    assert e_out < tols[0] and e_loss < tols[1] and worst[0][1] < tols[2], worst      # (the fixture holds gradient NORMS: a whole-tensor measure)


@pytest.mark.parametrize('mode,tols,dp', [('fp32', (1e-3, 1e-3, 2e-2), 0.0), ('fp32', (1e-3, 1e-3, 2e-2), 0.3), ('bf16', (6e-2, 2e-2, 1.0), 0.0)])
def test_train_step_against_oracle(mode, tols, dp):
    O = _oracle()
    cfg = O.make_cfg(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), num_classes=40)
    cfg['drop_path_rate'] = dp
    B = 4
    m, sd = build(cfg, mode, dp)
    m.train()
    x = O.gen_input(B, seed=1)
    target = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(5))
    masks = None
    if dp > 0:
        eng = m.engine(B, True)
        g = torch.Generator().manual_seed(5)
        masks = {}
        for site in eng.dp_scale:
            keep = 1 - eng.dp_rates[site]
            masks[site] = (torch.rand(B, generator=g) < keep).float() / keep
        eng.set_drop_path_masks(masks)
        eng.fixed_masks = True
    m.zero_grad()
    out = m(x.cuda())
    loss = F.cross_entropy(out, target.cuda(), label_smoothing=0.1)
    loss.backward()
    oloss, oout, ograds = O.train_step_grads(sd, x, target, cfg, dp_masks=masks, smoothing=0.1)
    from oracle import ga_convnext_oracle as GO
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters()}
    errs = GO.grad_errors(grads, ograds)
    if mode == 'bf16':
        gmax = max(float(g_.abs().max()) for g_ in ograds.values())
        errs = {n: e for n, e in errs.items() if float(ograds[n].abs().max()) >= 1e-4 * gmax}
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:4]
    e_out, e_loss = rel(out, oout), abs(float(loss) - float(oloss)) / abs(float(oloss))
    print(f'[convnext {mode} dp={dp}] logits {e_out:.2e} loss {e_loss:.2e} worst grads {worst}')
    assert e_out < tols[0] and e_loss < tols[1]
    if mode == 'bf16':      # whole-tensor gates (tests/_gradcheck.py): norm-relative error and direction of every gradient
        assert_grads_close(grads, ograds, BF16_REL, BF16_COS, 'bf16 train step')
    else:
        assert worst[0][1] < tols[2], worst


def test_fused_train_step_runs_cross_entropy():
    """TrainStep's fused loss on the single-output model is plain cross entropy (one head: no cross-head term)"""
    import imagenet_models_amd as A
    O = _oracle()
    cfg = O.make_cfg(dims=(16, 32, 64, 128), depths=(1, 1, 2, 1), num_classes=40)
    m, sd = build(cfg, 'fp32')
    m.train()
    opt = A.create_optimizer_v2(m, opt='sgd', lr=0.0, weight_decay=0.0, momentum=0.0)
    step = A.TrainStep(m, opt, 4, lam=-0.8)
    x = O.gen_input(4, seed=1)
    target = torch.randint(0, 40, (4,), generator=torch.Generator().manual_seed(5))
    loss = step(x.cuda(), target.cuda())
    ref = F.cross_entropy(O.forward(sd, x, cfg), target)
    assert abs(float(loss) - float(ref)) / float(ref) < 1e-3


def test_registry():
    import imagenet_models_amd as A
    assert sum(p.numel() for p in A.create_model('convnext_tiny').parameters()) == 28589128
    assert sum(p.numel() for p in A.create_model('convnext_small').parameters()) == 50223688
