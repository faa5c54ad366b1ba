#!/usr/bin/env python3
"""How many chunks would the fit stream if observations were compacted per lane across views? (analysis only)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from sucre_amd import engine, synth

W, H, NN = 1920, 1080, 64
scene = synth.make_scene(W, H, NN, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(H, W, len(views))
r.match(views[scene.target], views)
keep = r.view_keep().cpu().bool()
Hp, Wp = (H + 15) // 16 * 16, (W + 15) // 16 * 16
mask = torch.zeros((len(views), Hp, Wp), dtype=torch.bool, device='cuda')
for k in range(len(views)):
    if keep[k]:
        z, _ = r.export_view(k)
        mask[k, :H, :W] = z > 0
n_obs = int(mask.sum())
quad = mask.view(len(views), Hp, Wp // 4, 4).any(dim=3)
tiles = quad.view(len(views), Hp // 16, 16, Wp // 16, 4).permute(0, 1, 3, 2, 4).reshape(len(views), Hp // 16, Wp // 16, 64)
tile_any = tiles.any(dim=3)
chunks_now = int(tile_any.sum())
lane_cnt = tiles.sum(dim=0)
chunks_lane = int(lane_cnt.max(dim=2).values.sum())
quads_nonempty = int(tiles.sum())
print(f'n_obs {n_obs}  full-chunk equivalents {n_obs / 256:.0f}')
print(f'chunks now (tile x view touching)      {chunks_now}   slots/obs {chunks_now * 256 / n_obs:.3f}')
print(f'chunks lane-compacted (max over lanes) {chunks_lane}   slots/obs {chunks_lane * 256 / n_obs:.3f}')
print(f'non-empty quads / 64                   {quads_nonempty / 64:.0f}   slots/obs {quads_nonempty * 4 / n_obs:.3f}')
