"""weight gradient with a small output (GA-CSWin proj: M 50176, N 256, K 256) against the row split of the wide TN form"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops, _lib

dt = ops.GA_BF16
lib = _lib.load()
for (M, N, K) in ((50176, 256, 256), (50176, 256, 768), (200704, 128, 128), (50176, 512, 512)):
    Y = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    X = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    dW = torch.zeros(N, K, device='cuda')
    for wgs in (0, 32, 48, 64, 96, 128, 192, 256):
        if wgs: lib.ga_set_knob(b'TN2_WGS', wgs)
        else: lib.ga_unset_knob(b'TN2_WGS')
        p = ops.Plan(); p.wgrad(Y, X, dW, M, N, K, dt)
        for _ in range(3): p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.run()
        e1.record(); torch.cuda.synchronize()
        print(f'M={M} N={N} K={K} TN2_WGS={wgs:3d} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us', flush=True)
