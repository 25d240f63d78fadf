import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagenet_models_amd as A
from oracle import ga_convnext_oracle as O
cfg = O.make_cfg(dims=(16, 32, 64, 128, 128), depths=(1, 1, 6, 1, 1), gram_dim=32, dim_embed=64, num_classes=40, naggre=2)
sd = O.fill_state(cfg)
B = 8
x = O.gen_input(B, seed=2).cuda()
y = torch.randint(0, 40, (B,), generator=torch.Generator().manual_seed(2)).cuda()
res = {}
for tag in ('plain', 'plain2', 'fb', 'fb_comm', 'fb_comm_nan'):
    m = A.GA_ConvNeXt(num_classes=40, depths=cfg['depths'], dims=cfg['dims'], gram_embedding_gropus=cfg['gram_groups'],
                      dim_embed=cfg['dim_embed'], stage3_naggre=cfg['naggre'], gram_dim=cfg['gram_dim'], math_mode='fp32')
    m.load_state_dict(sd)
    m = m.cuda().train()
    opt = A.create_optimizer_v2(m, opt='sgd', lr=1e-2, momentum=0.9, weight_decay=0.05)
    kw = {}
    if tag.startswith('fb'):
        kw = dict(force_buckets=True, bucket_elems=50_000)
    if 'comm' in tag:
        kw['comm'] = A.NativeComm()
    if 'nan' in tag:
        kw['nan_guard'] = True
    step = A.TrainStep(m, opt, B, lam=-0.8, **kw)
    loss = step(x, y)
    torch.cuda.synchronize()
    res[tag] = (float(loss), m.flat_state()['params'].clone(), m.flat_state()['slices'])
ref = res['plain']
for tag in res:
    d = (res[tag][1] - ref[1]).abs()
    i = int(d.argmax())
    name = [n for n, (o, k) in ref[2].items() if o <= i < o + k]
    print(tag, res[tag][0], float(d.max()), i, name, float(res[tag][1][i]), float(ref[1][i]))
