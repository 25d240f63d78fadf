#!/usr/bin/env python3
"""Micro-benchmark of the GEMM-family launches of one ga_convnext_tiny_768 train step at batch 256 (bf16):
per (site, shape): device time (HIP events, median of N), TFLOP/s, algorithmic GB/s."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops  # noqa: E402

B = int(os.environ.get('GB_BATCH', '256'))
ITERS = int(os.environ.get('GB_ITERS', '20'))
DT = torch.bfloat16
dt = ops.GA_BF16


def timeit(plan, iters=None):
    iters = iters or ITERS
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2]


def rnd(*shape):
    return (torch.randn(*shape, device='cuda') * 0.5).to(DT)


rows = []


def report(name, ms, flops, bytes_):
    rows.append(dict(site=name, ms=round(ms, 4), tflops=round(flops / ms / 1e9, 1), gbs=round(bytes_ / ms / 1e6, 1)))
    print(f'{name:28s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TF/s  {bytes_ / ms / 1e6:8.1f} GB/s', flush=True)


STAGES = [int(x) for x in os.environ.get('GB_STAGES', '0,1,2,3').split(',') if x != '']
ITERS = int(os.environ.get('GB_ITERS', '20'))
for stage, (C, res) in enumerate([(96, 56), (192, 28), (384, 14), (768, 7)]):
    if stage not in STAGES:
        continue
    M = B * res * res
    x, h, y = rnd(M, C), rnd(M, 4 * C), rnd(M, C)
    W1, W2 = rnd(4 * C, C), rnd(C, 4 * C)
    W1T, W2T = rnd(C, 4 * C), rnd(4 * C, C)
    b1, b2 = torch.randn(4 * C, device='cuda'), torch.randn(C, device='cuda')
    G = torch.zeros(4 * C, C, device='cuda'); G2 = torch.zeros(C, 4 * C, device='cuda'); gb = torch.zeros(4 * C, device='cuda')
    fl = 2.0 * M * C * 4 * C
    e = 2
    p = ops.Plan(); p.gemm(x, W1, h, M, 4 * C, C, dt, bias=b1)
    report(f's{stage} fc1 fwd', timeit(p), fl, e * M * 5 * C)
    h2 = rnd(M, 4 * C)
    p = ops.Plan(); p.gemm(x, W1, h, M, 4 * C, C, dt, bias=b1, act=ops.ACT_GELU, C2=h2, c2_mode=2)
    report(f's{stage} fc1 fwd(+gelu,+gelu\')', timeit(p), fl, e * M * 9 * C)
    p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt, H=h2, ldh=4 * C, h_is_deriv=True, colsum=gb)
    report(f's{stage} dgrad2(*g)', timeit(p), fl, e * M * 9 * C)
    del h2
    p = ops.Plan(); p.gemm(h, W2, y, M, C, 4 * C, dt, a_act=ops.ACT_GELU, bias=b2, R=x, ldr=C)
    report(f's{stage} fc2 fwd(gelu,res)', timeit(p), fl, e * M * 6 * C)
    p = ops.Plan(); p.gemm(h, W2, y, M, C, 4 * C, dt, bias=b2, R=x, ldr=C)
    report(f's{stage} fc2 fwd(no gelu)', timeit(p), fl, e * M * 6 * C)
    p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt, H=h, ldh=4 * C, colsum=gb)
    report(f's{stage} dgrad2(gelu\')', timeit(p), fl, e * M * 9 * C)
    p = ops.Plan(); p.gemm(y, W2T, h, M, 4 * C, C, dt)
    report(f's{stage} dgrad2(plain)', timeit(p), fl, e * M * 5 * C)
    p = ops.Plan(); p.gemm(h, W1T, y, M, C, 4 * C, dt)
    report(f's{stage} dgrad1', timeit(p), fl, e * M * 5 * C)
    p = ops.Plan(); p.wgrad(h, x, G, M, 4 * C, C, dt)
    report(f's{stage} wgrad1', timeit(p), fl, e * M * 5 * C)
    p = ops.Plan(); p.wgrad(y, h, G2, M, C, 4 * C, dt, dbias=b2)
    report(f's{stage} wgrad2(+dbias)', timeit(p), fl, e * M * 5 * C)
    del x, h, y
# big square-ish reference
M = N = K = 4096
a, b, c = rnd(M, K), rnd(N, K), rnd(M, N)
p = ops.Plan(); p.gemm(a, b, c, M, N, K, dt)
report('square 4096^3', timeit(p), 2.0 * M * N * K, 2 * 3 * M * N)
p = ops.Plan(); p.wgrad(a, b, torch.zeros(N, K, device='cuda'), M, N, K, dt)
report('wgrad 4096^3', timeit(p), 2.0 * M * N * K, 2 * 2 * M * N)
del a, b, c
for (M, N, K) in [(8192, 8192, 8192), (73856, 2304, 768), (73856, 768, 768), (73856, 3072, 768), (73856, 768, 3072), (50176, 768, 1920)]:
    a, b, c = rnd(M, K), rnd(N, K), rnd(M, N)
    p = ops.Plan(); p.gemm(a, b, c, M, N, K, dt)
    report(f'plain M{M} N{N} K{K}', timeit(p), 2.0 * M * N * K, 2 * (M * N + M * K + N * K))
    del a, b, c
for (M, N, K) in [(73856, 2304, 768), (73856, 768, 768), (73856, 3072, 768), (73856, 768, 3072)]:
    y, x = rnd(M, N), rnd(M, K)
    G = torch.zeros(N, K, device='cuda')
    p = ops.Plan(); p.wgrad(y, x, G, M, N, K, dt)
    report(f'wgrad M{M} N{N} K{K}', timeit(p), 2.0 * M * N * K, 2 * (M * N + M * K))
    del y, x, G
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
