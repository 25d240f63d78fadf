#!/usr/bin/env python3
"""host-side issue time of one train step vs its device time (is the Python launcher the bottleneck?)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagenet_models_amd as A
torch.manual_seed(0)
m = A.create_model('ga_convnext_tiny_768', drop_path_rate=0.2, math_mode='bf16').cuda().train()
opt = A.create_optimizer_v2(m, opt='adamw', lr=1e-3, weight_decay=0.05)
step = A.TrainStep(m, opt, 256, lam=-0.8, loss='ce')
x = torch.randn(256, 3, 224, 224).cuda(); y = torch.randint(0, 1000, (256,)).cuda()
for _ in range(5):
    step(x, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue {1e3 * (t1 - t0) / n:.2f} ms/step, until device idle {1e3 * (t2 - t0) / n:.2f} ms/step')
# pure host cost: issue one step with the device idle and measure the call time only
torch.cuda.synchronize()
t0 = time.perf_counter(); step(x, y); t1 = time.perf_counter(); torch.cuda.synchronize()
print(f'one step issued on an idle device: {1e3 * (t1 - t0):.2f} ms of host time')
