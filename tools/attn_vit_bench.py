"""global attention kernels at the ViT-B/16 @ 384 shape (development aid)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagenet_models_amd import ops

def main():
    for B, N, H in ((128, 577, 12), (256, 197, 12)):
        hd, dt = 64, torch.bfloat16
        C = H * hd
        qkv = torch.randn(B * N, 3 * C, device='cuda').to(dt)
        do = torch.randn(B * N, C, device='cuda').to(dt)
        out = torch.empty(B * N, C, dtype=dt, device='cuda'); lse = torch.empty(B, H, N, device='cuda')
        dqkv = torch.empty(B * N, 3 * C, dtype=dt, device='cuda'); ws = torch.empty(B * H * N, device='cuda')
        p = ops.Plan(eager=True)
        d = p.attn_desc(qkv, out, lse, B, N, H, hd, hd ** -0.5, ops.GA_BF16)
        for name, fn, mult in (('fwd', lambda: p.attn_fwd(d), 1.0), ('bwd', lambda: p.attn_bwd(d, do, dqkv, ws), 2.5)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            gf = 4.0 * N * N * hd * B * H * mult / 1e9
            print(f'B{B} N{N} H{H} {name}: {ms:.3f} ms  {gf / ms:.0f} TFLOP/s', flush=True)

if __name__ == '__main__':
    main()
