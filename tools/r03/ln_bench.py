"""LayerNorm forward with affine parameters + mean / rstd outputs at the downsample shapes (half batch) vs the block form"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops

dt = ops.GA_BF16
for M, C in ((401408, 96), (100352, 192), (25088, 384), (50176, 384), (12544, 768)):
    x = torch.randn(M, C, device='cuda').to(torch.bfloat16)
    y = torch.empty_like(x)
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    mean, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    for name, args in (('affine+mean+rstd', (x, w, b, y, mean, rstd)), ('affine+rstd', (x, w, b, y, None, rstd)), ('affine only', (x, w, b, y, None, None)),
                       ('mean+rstd', (x, None, None, y, mean, rstd)), ('block (rstd only)', (x, None, None, y, None, rstd))):
        p = ops.Plan(); p.layernorm_fwd(*args, M, C, 1e-6, dt)
        for _ in range(3): p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f'M={M:7d} C={C:4d} {name:18s} {us:7.1f} us  {2 * M * C * 2 / us / 1e6:7.2f} TB/s', flush=True)
