#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernel families of one ga_convnext_tiny_768 train step at batch 256 (bf16):
depthwise 7x7 (fwd / bwd-data / bwd-weight), LayerNorm (fwd / bwd), BatchNorm pieces.  Per (site, stage): device time
(HIP events, median), algorithmic GB/s (bytes each tensor is read / written once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops  # noqa: E402

B = int(os.environ.get('GB_BATCH', '256'))
ITERS = int(os.environ.get('GB_ITERS', '20'))
WHAT = os.environ.get('EW_WHAT', 'dw,ln,bn').split(',')
STAGES = [int(x) for x in os.environ.get('GB_STAGES', '0,1,2,3').split(',') if x != '']
DT = torch.bfloat16
dt = ops.GA_BF16


def timeit(plan):
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(ITERS):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


def rnd(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).to(DT)


def report(name, ms, bytes_, flops=0.0):
    extra = f'  {flops / ms / 1e9:7.1f} TF/s' if flops else ''
    print(f'{name:28s} {ms:8.3f} ms  {bytes_ / ms / 1e6:8.1f} GB/s{extra}', flush=True)


for stage, (C, res) in enumerate([(96, 56), (192, 28), (384, 14), (768, 7)]):
    if stage not in STAGES:
        continue
    M = B * res * res
    e = 2
    x, y, z = rnd(M, C), rnd(M, C), rnd(M, C)
    if 'dw' in WHAT:
        w49 = torch.randn(49, C, device='cuda')
        bias = torch.randn(C, device='cuda')
        p = ops.Plan(); p.dwconv7_fwd(x, w49, bias, y, B, res, res, C, dt)
        report(f's{stage} dwconv7 fwd', timeit(p), e * M * C * 2, 2.0 * 49 * M * C)
        p = ops.Plan(); p.dwconv7_bwd_data(y, w49, x, z, B, res, res, C, dt)
        report(f's{stage} dwconv7 bwd-data(+res)', timeit(p), e * M * C * 3, 2.0 * 49 * M * C)
        dw = torch.zeros(49, C, device='cuda'); db = torch.zeros(C, device='cuda')
        p = ops.Plan(); p.dwconv7_bwd_weight(y, x, dw, db, B, res, res, C, dt)
        report(f's{stage} dwconv7 bwd-weight', timeit(p), e * M * C * 2, 2.0 * 49 * M * C)
    if 'ln' in WHAT:
        rstd = torch.empty(M, device='cuda')
        p = ops.Plan(); p.layernorm_fwd(x, None, None, y, None, rstd, M, C, 1e-6, dt)
        report(f's{stage} layernorm fwd', timeit(p), e * M * C * 2)
        dwv = torch.zeros(C, device='cuda'); dbv = torch.zeros(C, device='cuda')
        p = ops.Plan(); p.layernorm_bwd(z, y, None, rstd, None, None, x, dwv, dbv, M, C, True, dt)
        report(f's{stage} layernorm bwd (affine grads)', timeit(p), e * M * C * 3)
        p = ops.Plan(); p.layernorm_bwd(z, y, None, rstd, None, None, x, None, None, M, C, True, dt)
        report(f's{stage} layernorm bwd (block norm)', timeit(p), e * M * C * 3)
        sc = torch.rand(B, device='cuda')
        p = ops.Plan(); p.rowscale(x, sc, y, M * C, res * res * C, dt)
        report(f's{stage} rowscale', timeit(p), e * M * C * 2)
    if 'bn' in WHAT and stage == 2:
        # bottleneck-like shapes: (B*196, 192) and (B*196, 768)
        for Cb in (192, 768):
            xb, yb, zb = rnd(M, Cb), rnd(M, Cb), rnd(M, Cb)
            sc, sh = torch.randn(Cb, device='cuda'), torch.randn(Cb, device='cuda')
            p = ops.Plan(); p.affine_act(xb, sc, sh, None, yb, M, Cb, True, dt)
            report(f'bn affine+relu C={Cb}', timeit(p), e * M * Cb * 2)
            p = ops.Plan(); p.affine_act(xb, sc, sh, zb, yb, M, Cb, True, dt)
            report(f'bn affine+res+relu C={Cb}', timeit(p), e * M * Cb * 3)
            mean, rs_ = torch.randn(Cb, device='cuda'), torch.rand(Cb, device='cuda') + 0.5
            s1, s2 = torch.zeros(Cb, device='cuda'), torch.zeros(Cb, device='cuda')
            p = ops.Plan(); p.bn_bwd_reduce(zb, yb, xb, mean, rs_, s1, s2, M, Cb, dt)
            report(f'bn bwd reduce C={Cb}', timeit(p), e * M * Cb * 3)
            w = torch.randn(Cb, device='cuda')
            dx = torch.empty_like(xb)
            p = ops.Plan(); p.bn_bwd_apply(zb, yb, xb, mean, rs_, w, s1, s2, M, dx, M, Cb, dt)
            report(f'bn bwd apply C={Cb}', timeit(p), e * M * Cb * 4)
