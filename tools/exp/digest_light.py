#!/usr/bin/env python3
"""md5 digests of (trace, J, parameters) of light-model fits -- run with two builds (SUCRE_HIP_LIB) and diff."""
import hashlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import engine, synth  # noqa: E402

for (W, H, nn, seed, T) in ((800, 600, 8, 5, 12), (1920, 1080, 64, 0, 4), (333, 207, 13, 7, 15)):
    scene = synth.make_scene(W, H, nn, seed=seed, device='cuda')
    views = engine.device_views_from_scene(scene, 'cuda')
    r = engine.Restoration(H, W, len(views), light=True)
    r.match(views[scene.target], views)
    for closed in (False, True):
        r.fit_init(views[scene.target])
        t = r.fit(T, use_closed_form=closed)
        torch.cuda.synchronize()
        d = hashlib.md5(t.cpu().numpy().tobytes() + r.J().cpu().numpy().tobytes() + r.params().cpu().numpy().tobytes()).hexdigest()
        print(f'DIGEST light {W}x{H}x{nn + 1} closed={closed} n_obs={r.n_obs()} {d}', flush=True)
    del r, views, scene
    torch.cuda.empty_cache()
