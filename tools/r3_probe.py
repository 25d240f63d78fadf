#!/usr/bin/env python3
"""Where the 3-slot ring NT form spends its time: the debug build's R3_DBG switches (results deliberately wrong) one at a time.
    GAEXT_LIB=$PWD/imagenet-models_amd/csrc/libgaext_dbg.so python tools/r3_probe.py
bits: 1 no epilogue, 2 no global stores, 4 no GELU, 8 no operand loads, 16 no DMA, 32 wait-free K loop"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops, _lib  # noqa: E402

DT, dt = torch.bfloat16, ops.GA_BF16
assert 'DEBUG-BUILD' in _lib.config_string(), 'needs the -DGAEXT_DEBUG build (GAEXT_LIB)'


def rnd(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).to(DT)


def t(plan, iters=8):
    plan.run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


B = 256
for stage, (C, res) in [(1, (192, 28)), (2, (384, 14)), (3, (768, 7))]:
    M = B * res * res
    x, h, y, h2 = rnd(M, C), rnd(M, 4 * C), rnd(M, C), rnd(M, 4 * C)
    W1, W2, W2T = rnd(4 * C, C), rnd(C, 4 * C), rnd(4 * C, C)
    b1, b2 = torch.randn(4 * C, device='cuda'), torch.randn(C, device='cuda')
    gb = torch.zeros(4 * C, device='cuda')
    rs = torch.rand(B, device='cuda') + 0.5
    fl = 2.0 * M * C * 4 * C
    sites = {}
    p = ops.Plan(); p.gemm(x, W1, h, M, 4 * C, C, dt, bias=b1, act=ops.ACT_GELU, C2=h2, c2_mode=2); sites['fc1'] = p
    p = ops.Plan(); p.gemm(h, W2, y, M, C, 4 * C, dt, bias=b2, R=x, ldr=C, rowscale=rs, rows_per_scale=res * res); sites['fc2'] = p
    p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt, H=h2, ldh=4 * C, h_is_deriv=True, colsum=gb); sites['dg2'] = p
    p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt); sites['plainN4C'] = p
    for name, plan in sites.items():
        print(f's{stage} {name:9s} ', end='', flush=True)
        for dbg in (0, 2, 4, 6, 8, 10, 1, 33, 17):
            with _lib.knobs(NT_R3=15, R3_DBG=dbg):
                ms = t(plan)
            print(f'{dbg}:{ms * 1e3:6.1f}us/{fl / ms / 1e9:5.0f}TF  ', end='', flush=True)
        print(flush=True)
