"""GPU: fp32-mode gradients of the GA-CSWin, MAP-ConvNeXt and MAP-PiT engines against FLOAT64 ground truth.

tests/golden/{cswin_v6,map_v5,pit_v8}_train_b4_fp64.npz hold one train step (B = 4) of the narrow parity configurations run
through the oracle restatements in float64 (oracle/gen_golden_fp64.py).  Against an fp32 oracle a gradient gate has to sit
at 2e-2, because the fp32 reference's own backward is only good to ~4e-3; against float64 the library's fp32 math mode is
gated at 5e-3 per tensor -- norm relative and 16-value head relative to the tensor's max -- exactly as
test_model_gpu.py::test_fp32_mode_gradients_vs_fp64_ground_truth does for GA-ConvNeXt."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _flat(outs):
    f = []
    for o in outs:
        f.extend(o if isinstance(o, (list, tuple)) else [o])
    return f


def _cases():
    import test_cswin_model_gpu as TC
    import test_map_model_gpu as TM
    import test_map_pit_gpu as TP
    from oracle import ga_cswin_oracle as CS, map_oracle as MP, map_pit_oracle as PT
    return {'cswin_v6': (CS, TC.build, 'ga', ('depth', 'split_size', 'num_heads', 'dims')),
            'map_v5': (MP, TM.build, 'map', ('dims', 'depths')),
            'pit_v8': (PT, TP.build, 'map', ('base_dims', 'depth', 'heads'))}


@pytest.mark.parametrize('tag', ['cswin_v6', 'map_v5', 'pit_v8'])
def test_fp32_mode_gradients_vs_fp64_ground_truth(tag):
    import imagenet_models_amd as A
    O, build, kind, tuples = _cases()[tag]
    z = np.load(os.path.join(GOLDEN, f'{tag}_train_b4_fp64.npz'))
    cfg = json.loads(str(z['cfg']))
    for k in tuples:
        if k in cfg:
            cfg[k] = tuple(cfg[k])
    m, _ = build(cfg, 'fp32')
    m.train()
    m.zero_grad()
    size = cfg.get('image_size')
    x = O.gen_input(4, seed=1, size=size) if size else O.gen_input(4, seed=1)
    outs = m(x.cuda())
    kw = json.loads(str(z['step_kw']))
    target = torch.from_numpy(z['target']).cuda()
    loss = A.ga_loss(outs, target, kw['lam']) if kind == 'ga' else A.map_loss(outs, target, kw['dec_lam'])
    loss.backward()
    e_out = _rel(torch.stack(_flat(outs))[:, :, :40], z['logits'])
    e_loss = abs(float(loss) - float(z['loss'])) / abs(float(z['loss']))
    grads = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters()}
    names = [str(n) for n in z['grad_names']]
    assert set(names) == set(grads), sorted(set(names) ^ set(grads))[:6]
    gmax = float(z['grad_absmax'].max())
    worst = []
    for i, n in enumerate(names):
        amax, nref = float(z['grad_absmax'][i]), float(z['grad_norm'][i])
        head = grads[n].reshape(-1)[:16].numpy()
        dh = float(np.abs(head - z['grad_head'][i][:head.size]).max())
        if amax >= 1e-4 * gmax:
            worst.append((max(abs(float(grads[n].norm()) - nref) / nref, dh / amax), n))
        else:
            worst.append((dh / (1e-4 * gmax) * 5e-3, n))      # analytically-zero gradients: |.| < 1e-4 of the global max
    worst.sort(reverse=True)
    print(f'[{tag} fp32 mode vs fp64 oracle] logits {e_out:.2e} loss {e_loss:.2e} worst grads {worst[:5]}')
    assert e_out < 1e-3 and e_loss < 1e-3
    assert worst[0][0] < 5e-3, worst[:10]
