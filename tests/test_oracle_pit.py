"""CPU: the PiT oracle restatement (oracle/map_pit_oracle.py) against the golden vectors oracle/gen_golden_pit.py produced from
the REAL reference classes (/root/reference/MAP/models/map_pit.py:84-251 + map.py's MAPHead; timm's Block restated in a test-only
stub) with the head's nn.Dropout modules at p = 0.  Outputs / loss 1e-4 relative, gradient norms 1e-2, top-5 bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import map_oracle as MO
from oracle import map_pit_oracle as O

KEYS = ('image_size', 'patch_size', 'stride', 'base_dims', 'depth', 'heads', 'num_classes', 'last_dim', 'n_groups', 'n_tokens', 'gram_group')


def _load(name):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(str(z['cfg']))
    return z, O.make_cfg(**{k: cfg[k] for k in KEYS})


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def test_registered_variant_shape_facts():
    # map_pit.py:224-242: 224 / patch 16 / stride 8 -> 27 x 27 tokens, channels 144 / 288 / 576, head_dim 48
    cfg = O.make_cfg('map_pit_s')
    assert cfg['width'] == 27 and cfg['dims'] == (144, 288, 576)
    shapes = O.state_shapes(cfg)
    assert sum(int(np.prod(s)) for k, s in shapes.items() if O.is_param(k)) == 36147424
    assert shapes['pools.0.conv.weight'] == (288, 1, 3, 3) and shapes['pos_embed'] == (1, 144, 27, 27)


@pytest.mark.parametrize('tag', ['pit_v8', 'pit_s'])
def test_eval_logits_and_topk(tag):
    z, cfg = _load(f'{tag}_eval.npz')
    sd = O.fill_state(cfg)
    assert sum(v.numel() for k, v in sd.items() if O.is_param(k)) == int(z['param_count'])
    x = O.gen_input(int(z['batch']), seed=0, size=cfg['image_size'])
    with torch.no_grad():
        outs = O.forward(sd, x, cfg, training=False)
    assert _rel(torch.stack(outs)[:, :, :40].numpy(), z['logits']) < 1e-4
    assert np.array_equal((sum(outs) / len(outs)).topk(5, 1, True, True)[1].numpy(), z['top5'])


def test_train_step_against_reference():
    z, cfg = _load('pit_v8_train_b4.npz')
    sd = O.fill_state(cfg)
    x = O.gen_input(int(z['batch']), seed=1, size=cfg['image_size'])
    loss, outs, grads, stats = O.train_step_grads(sd, x, torch.from_numpy(z['target']), cfg, dec_lam=float(z['dec_lam']))
    assert abs(float(loss) - float(z['loss'])) / abs(float(z['loss'])) < 1e-4
    assert _rel(torch.stack([o[0] for o in outs])[:, :, :40].numpy(), z['org']) < 1e-4
    assert _rel(torch.stack([o[1] for o in outs])[:, :, :40].numpy(), z['avg']) < 1e-4
    names = [str(n) for n in z['grad_names']]
    assert names == list(grads.keys())
    gmax = float(z['grad_norm'].max())
    for n, ref_norm in zip(names, z['grad_norm'].tolist()):
        if ref_norm > 1e-3 * gmax:
            assert abs(float(grads[n].double().norm()) - ref_norm) / ref_norm < 1e-2, n


def test_drop_path_schedule():
    # map_pit.py:116-118: rate * i / total_block over all blocks in order
    cfg = O.make_cfg('map_pit_s', drop_path_rate=0.12)
    r = O.drop_path_rates(cfg)
    assert len(r) == 12 and r['transformers.0.blocks.0.'] == 0.0 and abs(r['transformers.2.blocks.3.'] - 0.12 * 11 / 12) < 1e-12
    assert MO is not None
