import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops, _lib
B, H, W, C = 256, 112, 112, 64
M = B * H * W
x = (torch.randn(M, C, device='cuda') * 0.5).bfloat16()
dy = (torch.randn(M, C, device='cuda') * 0.5).bfloat16()
dW = torch.zeros(64, 576, device='cuda')
for form in (1, 0, 1, 0):
    with _lib.knobs(CONV3_DIRECT=form):
        p = ops.Plan(); p.wgrad(dy, x, dW, M, 64, 576, ops.GA_BF16, x_kind=ops.A_CONV3, x_dims=(H, W, C))
        for _ in range(3): p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): p.run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'wgrad direct={form}: {ms:.3f} ms  {2.0 * M * 64 * 576 / ms / 1e9:.0f} TFLOP/s', flush=True)
