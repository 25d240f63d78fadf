#!/usr/bin/env python3
"""Float64 ground truth for the gradient gates of the GA-CSWin, MAP-ConvNeXt and MAP-PiT engines (TEST INFRASTRUCTURE).

One train step (B = 4) of the narrow parity configurations -- GA-CSWin V6, MAP V5, MAP-PiT V8, the ones the reference
fixtures of gen_golden_cswin / _map / _pit use -- run through the ORACLE restatements in float64 (the restatements were
checked against the imported reference classes when those fixtures were made; a float64 run of the reference itself is not
possible for the MAP families without timm).  Stored per gradient tensor: norm, sum, max |.|, first 16 values; logits, loss.
tests/test_fp64_truth_gpu.py gates the library's fp32 mode at 5e-3 against these, as test_model_gpu.py does for GA-ConvNeXt
(the fp32 reference's own backward is only good to ~4e-3, which is why the gates against fp32 oracles sit at 2e-2).

    python oracle/gen_golden_fp64.py            # writes tests/golden/{cswin_v6,map_v5,pit_v8}_train_b4_fp64.npz"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import ga_cswin_oracle as CS, map_oracle as MP, map_pit_oracle as PT  # noqa: E402
from oracle.gen_golden_cswin import V6  # noqa: E402
from oracle.gen_golden_map import V5  # noqa: E402
from oracle.gen_golden_pit import V8  # noqa: E402


def flat(outs):
    f = []
    for o in outs:
        f.extend(o if isinstance(o, (list, tuple)) else [o])
    return f


def run(tag, O, cfg, step_kw, size=None):
    B = 4
    sd = O.fill_state(cfg, dtype=torch.float64)
    x = (O.gen_input(B, seed=1, size=size) if size else O.gen_input(B, seed=1)).double()
    target = torch.randint(0, cfg['num_classes'], (B,), generator=torch.Generator().manual_seed(99))
    res = O.train_step_grads(sd, x, target, cfg, **step_kw)
    loss, outs, grads = res[0], flat(res[1]), res[2]
    assert all(g.dtype == torch.float64 for g in grads.values()) and outs[0].dtype == torch.float64
    names = list(grads.keys())
    head = np.zeros((len(names), 16))
    for i, n in enumerate(names):
        f = grads[n].reshape(-1)[:16]
        head[i, :f.numel()] = f.numpy()
    print(f'[{tag}] fp64 oracle train B={B}: loss {float(loss):.12f}, {len(names)} gradient tensors')
    np.savez_compressed(os.path.join(OUT, f'{tag}_train_b4_fp64.npz'), cfg=json.dumps(cfg), batch=B, target=target.numpy(), loss=float(loss),
                        step_kw=json.dumps(step_kw), logits=torch.stack([o.detach() for o in outs])[:, :, :40].numpy(),
                        grad_names=np.array(names), grad_norm=np.array([float(grads[n].norm()) for n in names]),
                        grad_sum=np.array([float(grads[n].sum()) for n in names]),
                        grad_absmax=np.array([float(grads[n].abs().max()) for n in names]), grad_head=head)


if __name__ == '__main__':
    torch.manual_seed(0)
    run('cswin_v6', CS, CS.make_cfg(**V6), dict(lam=-0.8))
    run('map_v5', MP, MP.make_cfg(**V5), dict(dec_lam=-0.8))
    v8 = PT.make_cfg(**V8)
    run('pit_v8', PT, v8, dict(dec_lam=-0.8), size=v8['image_size'])
