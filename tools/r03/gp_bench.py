"""gram pack forward / backward at the MAP shapes (C = 384, 24 groups, B = 256), token interleave 1 / 2 / 3"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops

B, C, groups = 256, int(os.environ.get('GP_C', '384')), int(os.environ.get('GP_G', '24'))
ntri = C * (C + 1) // 2
Kg = ntri // groups
Kp = (Kg + 7) // 8 * 8
dt = ops.GA_BF16
G = torch.randn(B, C, C, device='cuda')
for T in (1, 2, 3):
    out = torch.empty(B, groups * Kp, dtype=torch.bfloat16, device='cuda')
    inv = torch.empty(B, device='cuda')
    dvec = torch.randn(B, groups * Kp, device='cuda').to(torch.bfloat16)
    S = torch.empty(B, C, C, dtype=torch.bfloat16, device='cuda')
    pf = ops.Plan(); pf.gram_pack_fwd2(G, out, inv, B, C, groups, Kp, T, dt)
    pb = ops.Plan(); pb.gram_pack_bwd2(dvec, out, inv, S, B, C, groups, Kp, T, dt)
    for name, p in (('fwd', pf), ('bwd', pb)):
        for _ in range(3): p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): p.run()
        e1.record(); torch.cuda.synchronize()
        print(f'T={T} {name} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us', flush=True)
