#!/usr/bin/env python3
"""A/B of the 3-slot ring NT form (knob NT_R3) against the default dispatch on the NT GEMM launches of one train step,
interleaved rounds in ONE process (guide rule 24): per (site, shape) median device time of each arm and TFLOP/s.
    python tools/r3_ab.py [out.json]            R3_STAGES=1,2,3  R3_ROUNDS=7  R3_ITERS=6  R3_EXTRA=1 (heads / ViT shapes)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops, _lib  # noqa: E402

DT = torch.bfloat16
dt = ops.GA_BF16
ROUNDS = int(os.environ.get('R3_ROUNDS', '7'))
ITERS = int(os.environ.get('R3_ITERS', '6'))
B = int(os.environ.get('R3_BATCH', '256'))


def rnd(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).to(DT)


def time_once(plan):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS


ARMS = {'base': dict(NT_R3=0)}
for v in os.environ.get('R3_ARMS', '-1').split(','):
    ARMS['r3' if v == '-1' else f'r3s{v}'] = dict(NT_R3=15, R3_STAGGER=int(v))


def ab(name, build, flops, rows):
    plan = build()
    t = {k: [] for k in ARMS}
    for k, kv in ARMS.items():
        with _lib.knobs(**kv):
            plan.run()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, kv in ARMS.items():
            with _lib.knobs(**kv):
                t[k].append(time_once(plan))
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    row = dict(site=name, **{f'{k}_ms': round(v, 4) for k, v in med.items()}, **{f'{k}_tf': round(flops / v / 1e9, 1) for k, v in med.items()})
    rows.append(row)
    print(f"{name:34s} " + ' | '.join(f"{k} {v:.4f} ms {flops / v / 1e9:6.1f} TF x{med['base'] / v:.3f}" for k, v in med.items()), flush=True)


rows = []
STAGES = [int(x) for x in os.environ.get('R3_STAGES', '1,2,3').split(',') if x]
for stage, (C, res) in enumerate([(96, 56), (192, 28), (384, 14), (768, 7)]):
    if stage not in STAGES:
        continue
    for half in (1, 2):       # full batch and the half-batch chains of the forward trunk
        M = B * res * res // half
        x, h, y, h2 = rnd(M, C), rnd(M, 4 * C), rnd(M, C), rnd(M, 4 * C)
        W1, W2, W1T, W2T = rnd(4 * C, C), rnd(C, 4 * C), rnd(C, 4 * C), rnd(4 * C, C)
        b1, b2 = torch.randn(4 * C, device='cuda'), torch.randn(C, device='cuda')
        gb = torch.zeros(4 * C, device='cuda')
        rs = torch.rand(B, device='cuda') + 0.5
        fl = 2.0 * M * C * 4 * C
        tag = f's{stage}/M{M}'

        def fc1():
            p = ops.Plan(); p.gemm(x, W1, h, M, 4 * C, C, dt, bias=b1, act=ops.ACT_GELU, C2=h2, c2_mode=2); return p

        def fc2():
            p = ops.Plan(); p.gemm(h, W2, y, M, C, 4 * C, dt, bias=b2, R=x, ldr=C, rowscale=rs, rows_per_scale=res * res); return p

        def dg2():
            p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt, H=h2, ldh=4 * C, h_is_deriv=True, colsum=gb); return p

        def dg1():
            p = ops.Plan(); p.gemm(h, W1T, y, M, C, 4 * C, dt); return p

        ab(f'{tag} fc1(+gelu,gelu\')', fc1, fl, rows)
        ab(f'{tag} fc2(+rs,+res)', fc2, fl, rows)
        if half == 1:
            ab(f'{tag} dgrad2(*g,colsum)', dg2, fl, rows)
            ab(f'{tag} dgrad1(plain)', dg1, fl, rows)
        del x, h, y, h2
if os.environ.get('R3_EXTRA', '1') != '0':
    for (M, N, K) in [(50176, 768, 2208), (50176, 2208, 768), (50176, 1920, 768), (50176, 768, 1920), (50176, 960, 768), (50176, 192, 2208),
                      (50176, 2208, 192), (200704, 384, 192), (8192, 8192, 8192), (73856, 2304, 768), (73856, 768, 3072), (73856, 3072, 768)]:
        a, b, c = rnd(M, K), rnd(N, K), rnd(M, N)

        def plain():
            p = ops.Plan(); p.gemm(a, b, c, M, N, K, dt); return p
        ab(f'plain M{M} N{N} K{K}', plain, 2.0 * M * N * K, rows)
        del a, b, c
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
