"""timing probe of the stripe-attention kernels at the GA-CSWin-T stage shapes (development aid)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagenet_models_amd as A
from imagenet_models_amd import ops

def run(B, reso, C, heads, stripes, probe, bwd=False, iters=20):
    os.environ['GAEXT_PROBE'] = str(probe)
    dt = torch.bfloat16
    L = reso * reso
    qkv = torch.randn(B * L, 3 * C, device='cuda').to(dt)
    out = torch.empty(B * L, C, device='cuda', dtype=dt)
    do = torch.randn(B * L, C, device='cuda').to(dt)
    dqkv = torch.empty(B * L, 3 * C, device='cuda', dtype=dt)
    nb = len(stripes)
    lw = [torch.randn(C // nb, 1, 3, 3, device='cuda') for _ in range(nb)]
    lb = [torch.randn(C // nb, device='cuda') for _ in range(nb)]
    p = ops.Plan(eager=True)
    d = p.cswin_desc(qkv, out, B, reso, C, heads, stripes, list(zip(lw, lb)), 32 ** -0.5, ops.ga_dtype(dt))
    need = ops.cswin_attn_bwd_workspace(d)
    lws = torch.empty(max(need // 4, 1), device='cuda')
    def once():
        if bwd:
            p.cswin_attn_bwd(d, do, dqkv, lepe_ws=lws if (need and not os.environ.get('PROBE_NOWG')) else None)
        else:
            p.cswin_attn_fwd(d)
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

if __name__ == '__main__':
    shapes = {'s3': (256, 14, 256, 8, [(14, 7), (7, 14)]), 's1': (256, 56, 64, 2, [(56, 1), (1, 56)]),
              's2': (256, 28, 128, 4, [(28, 2), (2, 28)]), 's4': (256, 7, 512, 16, [(7, 7)])}
    probes = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [0]
    for bwd in (False, True):
        for name, shp in shapes.items():
            print('bwd' if bwd else 'fwd', name, ' '.join(f'p{pr}={run(*shp, pr, bwd):.1f}us' for pr in probes), flush=True)
