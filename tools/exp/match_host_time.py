#!/usr/bin/env python3
"""Host time vs GPU span of Restoration.match on an idle GPU (is the match stage's event time its kernels or the host?)."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from sucre_amd import engine, synth  # noqa: E402

scene = synth.make_scene(1920, 1080, 64, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(1080, 1920, len(views))
tgt = views[scene.target]
for i in range(8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    r.match(tgt, views)
    t1 = time.perf_counter()
    e1.record()
    torch.cuda.synchronize()
    print(f'match {i}: host {1e3 * (t1 - t0):.3f} ms, GPU span {e0.elapsed_time(e1):.3f} ms')
import cProfile
import pstats
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for i in range(5):
    r.match(tgt, views)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
