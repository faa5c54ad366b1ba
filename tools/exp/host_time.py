#!/usr/bin/env python3
"""Host time of every call of one restoration (match, fit_init, fit, J) next to the image's wall time, one image at a time:
what the GPU waits for between two images.  python tools/exp/host_time.py [path of another tree to import sucre_amd from]"""
import sys
import time
from pathlib import Path

import torch

root = Path(sys.argv[1]).resolve() if len(sys.argv) > 1 else Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
from sucre_amd import engine, synth  # noqa: E402

scene = synth.make_scene(1920, 1080, 64, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(1080, 1920, len(views))
tgt = views[scene.target]
names = ['match', 'fit_init', 'fit', 'J']
for rep in range(6):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    r.match(tgt, views, min_cover=1e-6); t.append(time.perf_counter())
    r.fit_init(tgt); t.append(time.perf_counter())
    r.fit(200, record_trace=True); t.append(time.perf_counter())
    J = r.J(); t.append(time.perf_counter())
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    print(f'{root.name:8s} image {rep}: ' + ' '.join(f'{n} {1e3 * (b - a):.3f}' for n, a, b in zip(names, t, t[1:])) + f' | host total {1e3 * (t[4] - t[0]):.3f} ms, wall {1e3 * (t[5] - t[0]):.3f} ms')
