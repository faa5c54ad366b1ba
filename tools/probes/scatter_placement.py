#!/usr/bin/env python3
"""Does scatter_kernel's duration depend on WHERE its workspace lies?  Six workspaces of one geometry in one process, the
same scene matched + finalized into each of them three times; run under rocprofv3 --kernel-trace and read the
per-dispatch durations in order (tools/probes/scatter_placement.sh).  Prints the workspaces' addresses."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import engine, synth  # noqa: E402

scene = synth.make_scene(1920, 1080, 64, seed=0)
views = engine.device_views_from_scene(scene, 'cuda')
rs = [engine.Restoration(1080, 1920, len(views)) for _ in range(6)]
for r in rs:
    print('workspace at 0x%x  (mod 1 GiB: 0x%x, mod 2 MiB: 0x%x)' % (r.ws.data_ptr(), r.ws.data_ptr() % (1 << 30), r.ws.data_ptr() % (1 << 21)))
for rep in range(3):
    for r in rs:
        r.match(views[scene.target], views)
        torch.cuda.synchronize()
