import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops
torch.manual_seed(0)
for (Bn, H, W, Cc) in [(2, 14, 14, 96), (1, 28, 28, 72), (3, 7, 7, 128), (1, 56, 56, 32)]:
    X = torch.randn(Bn, H, W, Cc).cuda().bfloat16()
    R = torch.randn(Bn, H, W, Cc).cuda().bfloat16()
    w49 = (torch.randn(49, Cc) * 0.1).cuda()
    P = ops.Plan(eager=True)
    DX = torch.full_like(X, 777.0); DXa = torch.full_like(X, 777.0); DX2 = torch.full_like(X, 777.0); REF2 = torch.empty_like(X)
    P.dwconv7_bwd_data(X, w49, R, DX, Bn, H, W, Cc, ops.GA_BF16)
    sc = torch.tensor([0.0, 1.25, 2.0, 1.0][:Bn], device='cuda')
    P.dwconv7_bwd_data(X, w49, R, DXa, Bn, H, W, Cc, ops.GA_BF16, dx2=DX2, scale2=sc)
    P.rowscale(DXa, sc, REF2, DXa.numel(), H * W * Cc, ops.GA_BF16)
    torch.cuda.synchronize()
    a = (DXa != DX).nonzero(); b = (DX2 != REF2).nonzero()
    print((Bn, H, W, Cc), 'DXa!=DX', len(a), a[:4].tolist(), 'DX2!=REF2', len(b), b[:4].tolist())
    if len(b):
        i = tuple(b[0].tolist()); print('   ', float(DX2[i]), float(REF2[i]), float(DXa[i]), float(sc[i[0]]))
        print('   rows', sorted(set(b[:, 1].tolist()))[:20], 'cols', sorted(set(b[:, 2].tolist()))[:20], 'ch', sorted(set(b[:, 3].tolist()))[:16])
