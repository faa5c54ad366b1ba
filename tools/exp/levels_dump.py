#!/usr/bin/env python3
"""Dumps the per-pixel observation counts (sorted, descending) of the bench's config-2 scene so that tile / wave
assignments can be modelled offline.  Experiment tool.  usage: python tools/exp/levels_dump.py out.npz [W H NN seed]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from sucre_amd import engine, synth  # noqa: E402

out = sys.argv[1]
W, H, NN, seed = (int(x) for x in (sys.argv[2:6] + ['1920', '1080', '64', '0'][len(sys.argv) - 2:]))
scene = synth.make_scene(W, H, NN, seed=seed, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(H, W, len(views))
r.match(views[scene.target], views)
keep = r.view_keep().cpu().numpy().astype(bool)
cnt = torch.zeros((H, W), dtype=torch.int32, device='cuda')
for k in range(len(views)):
    if keep[k]:
        z, _ = r.export_view(k)
        cnt += (z > 0).to(torch.int32)
# tile-padded pixel list as the engine sorts it: (tiles_y*16, tiles_x*16) with zeros outside the image
ty, tx = (H + 15) // 16, (W + 15) // 16
pad = torch.zeros((ty * 16, tx * 16), dtype=torch.int32, device='cuda')
pad[:H, :W] = cnt
flat = torch.sort(pad.flatten(), descending=True).values.cpu().numpy().astype(np.uint16)
np.savez_compressed(out, sorted_counts=flat, n_obs=r.n_obs(), W=W, H=H, n_views=len(views))
print('n_obs', r.n_obs(), 'pixels', flat.size, 'max', flat.max(), 'mean', flat.mean())
