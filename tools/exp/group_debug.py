import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
import numpy as np, torch
import helpers
from sucre_amd import engine
golden = helpers.load_fixture('plane_64x48_n4')
rt = golden['shared_trace']; T = rt.shape[0]
views = engine.device_views_from_scene(golden.scene, 'cuda')
groups, traces = [], []
for tgt in (int(x) for x in golden['shared_targets']):
    r = engine.Restoration(golden.scene.height, golden.scene.width, len(views))
    r.match(views[tgt], views); r.fit_init(views[tgt])
    tr = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
    groups.append(engine.HipWaterGroup([r], trace=tr)); traces.append(tr)
total = sum(g.n_obs() for g in groups)
print('total', total, int(golden['shared_n_total']))
for g in groups: g.set_n_obs_total(total)
for it in range(1, T + 1):
    sums = [g.grad(it) for g in groups]
    h = [s.cpu() for s in sums]
    red = h[0] + h[1]
    for s in sums: s.copy_(red)
    for g in groups: g.step(it)
for g in groups: g.finish()
tr = traces[0].cpu().numpy()
print('ranks equal', np.array_equal(tr, traces[1].cpu().numpy()))
print('max dparam vs golden', np.abs(tr[:, 1:] - rt[:, 1:]).max(), 'per row', np.abs(tr[:, 1:] - rt[:, 1:]).max(axis=1)[:6])
