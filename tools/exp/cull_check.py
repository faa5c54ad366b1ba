#!/usr/bin/env python3
"""Does the tile-level cull bite?  Times the match stage of a scene with many non-overlapping (far) views."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import engine, synth  # noqa: E402

for nn, far in ((16, 0), (16, 48)):
    scene = synth.make_scene(1920, 1080, nn, seed=0, device='cuda', far_views=far)
    views = engine.device_views_from_scene(scene, 'cuda')
    r = engine.Restoration(1080, 1920, len(views))
    for rep in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r.match(views[scene.target], views)
        e1.record()
        torch.cuda.synchronize()
    print(f'{nn} neighbours + {far} far views: match + finalize {e0.elapsed_time(e1):.3f} ms, n_obs {r.n_obs()}, '
          f'views with matches {(r.view_counts() > 0).sum().item()} of {len(views)}', flush=True)
