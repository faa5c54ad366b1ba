"""Where does the engine's match set differ from the oracle's?  (config-2 scene, CPU-rendered; run on the GPU box)"""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
import helpers
from oracle import oracle
from sucre_amd import engine, synth

W, H, N = 1920, 1080, 64
scene = synth.make_scene(W, H, N, seed=0)
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(H, W, len(views))
r.match(views[scene.target], views)
counts = r.view_counts().cpu().numpy()
tgt = scene.views[scene.target]
cam1 = helpers.oracle_cam(scene, tgt)
d1 = tgt.depth_f32().numpy()
tot = 0
for k, v in enumerate(scene.views):
    cam2 = helpers.oracle_cam(scene, v)
    d2 = v.depth_f32().numpy()
    m = oracle.match_view(d1, cam1, d2, cam2)
    if len(m) == counts[k]:
        continue
    mo = helpers.dense_map(m, H, W)
    me = r.match_map(k).cpu().numpy()
    diff = np.argwhere(mo != me)
    tot += len(diff)
    print(f'view {k} {v.name}: oracle {len(m)} engine {counts[k]} map diffs {len(diff)} (match_map_kernel agrees with count: {(me >= 0).sum() == counts[k]})')
    for (y, x) in diff[:6]:
        print('   pixel', x, y, 'oracle q', mo[y, x], 'engine q', me[y, x], 'depth1', d1[y, x], 'tile', x // 16, y // 16, 'in-tile', x % 16, y % 16)
print('total differing pixels', tot, 'target', scene.target)
