import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops
Bn, H, W, Cc = 1, 14, 56, 32
# unique exactly-representable values: value = row*64 + col + (ch % 2) * 0.5 ; ch pair id separately
yy, xx, cc = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(Cc), indexing='ij')
for name, val in (('row', yy), ('col', xx), ('ch', cc)):
    x = val.float()[None].contiguous() + 1
    X = x.cuda().bfloat16()
    w = torch.zeros(Cc, 1, 7, 7); w[:, 0, 3, 3] = 1.0
    w49 = w.reshape(Cc, 49).t().contiguous().cuda()
    Y = torch.full_like(X, 777.0)
    P = ops.Plan(eager=True)
    P.dwconv7_fwd(X, w49, torch.zeros(Cc).cuda(), Y, Bn, H, W, Cc, ops.GA_BF16)
    torch.cuda.synchronize()
    Yc = Y.float().cpu()[0]
    for r in (1, 2, 6):
        print(name, 'out row', r, 'col 12 ch 0..7:', Yc[r, 12, :8].tolist(), ' col 13:', Yc[r, 13, :8].tolist(), ' col 28 ch 8..15', Yc[r, 28, 8:16].tolist())
