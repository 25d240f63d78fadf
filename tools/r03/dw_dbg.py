import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imagenet_models_amd import ops
torch.manual_seed(0)
Bn, H, W, Cc = 1, 14, 56, 32
# x encodes its own coordinates: value = row * 64 + col + ch/64 (exact in bf16? use small ints): use row*8+col/8... keep exact: ints < 256
yy, xx, cc = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(Cc), indexing='ij')
for name, val in (('row', yy), ('col', xx), ('ch', cc)):
    x = val.float()[None].contiguous()
    X = x.cuda().bfloat16()
    for (ky, kx) in [(3, 3), (0, 3), (6, 3), (3, 0), (3, 6)]:
        w = torch.zeros(Cc, 1, 7, 7); w[:, 0, ky, kx] = 1.0
        w49 = w.reshape(Cc, 49).t().contiguous().cuda()
        Y = torch.full_like(X, 777.0)
        P = ops.Plan(eager=True)
        P.dwconv7_fwd(X, w49, torch.zeros(Cc).cuda(), Y, Bn, H, W, Cc, ops.GA_BF16)
        torch.cuda.synchronize()
        y = F.conv2d(x.permute(0, 3, 1, 2), w, None, padding=3, groups=Cc).permute(0, 2, 3, 1)
        d = (Y.float().cpu() - y).abs()
        bad = (d > 0.01).nonzero()
        print(name, (ky, kx), 'bad', len(bad), end=' ')
        if len(bad):
            i = bad[0].tolist()
            print('first', i, 'got', float(Y[tuple(i)]), 'want', float(y[tuple(i)]), 'rows', sorted(set(bad[:, 1].tolist()))[:8], 'cols', sorted(set(bad[:, 2].tolist()))[:12], 'ch', sorted(set(bad[:, 3].tolist()))[:8])
        else:
            print()
