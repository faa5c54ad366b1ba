import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from sucre_amd import engine, synth, sfm
scene = synth.make_scene(96, 64, 4, seed=7, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(scene.height, scene.width, len(views), light=True)
r.match(views[scene.target], views)
torch.cuda.synchronize()
for k, v in enumerate(scene.views):
    q = r.match_map(k)
    v1, u1 = torch.where(q >= 0)
    p2 = q[v1, u1].long(); W2 = scene.width
    u2, v2 = (p2 % W2).cpu(), torch.div(p2, W2, rounding_mode='floor').cpu()
    d = views[k].depth[v2.cuda(), u2.cuda()].cpu()
    K = scene.K
    Kinv = K.inverse()
    cP = Kinv @ (d * torch.stack([u2 + 0.5, v2 + 0.5, torch.ones_like(u2)]))   # sfm.py:90-93
    ext = r.export_view_ext(k)           # (3, H, W) dense planes
    got = ext[:, v1, u1].cpu()
    diff = (got - cP).abs().max().item()
    nz = int((got != cP).sum())
    print(k, len(u1), 'max abs diff', diff, 'differing', nz)
