#!/usr/bin/env python3
"""weight-gradient launches of the CSWin / PiT shapes: time per launch; the split / form knobs are read from the environment
by the library (GAEXT_TN2_WGS, GAEXT_TN2_PART_MIN, GAEXT_TN2)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops  # noqa: E402

dt = ops.GA_BF16


def timeit(plan, iters=20):
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


for (M, N, K) in [(50176, 768, 256), (50176, 256, 256), (50176, 1024, 256), (50176, 256, 1024), (200704, 384, 128), (200704, 512, 128),
                  (186624, 432, 144), (186624, 576, 144), (186624, 144, 576), (50176, 864, 288), (50176, 1152, 288)]:
    y = (torch.randn(M, N, device='cuda') * .5).bfloat16()
    x = (torch.randn(M, K, device='cuda') * .5).bfloat16()
    G = torch.zeros(N, K, device='cuda')
    p = ops.Plan()
    p.wgrad(y, x, G, M, N, K, dt)
    t = timeit(p)
    print(f'wgrad M{M} N{N} K{K}: {t:.3f} ms  {2.0 * M * N * K / t / 1e9:.0f} TF/s  {2.0 * (M * N + M * K) / t / 1e6:.0f} GB/s', flush=True)
