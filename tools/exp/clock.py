#!/usr/bin/env python3
"""Wave schedule of the fit kernel (SUCRE_EXP_CLOCK builds): s_memrealtime stamps of every wave."""
import ctypes as C
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
from sucre_amd import engine, synth, _lib
scene = synth.make_scene(1920, 1080, 64, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(1080, 1920, len(views))
r.match(views[scene.target], views)
r.fit_init(views[scene.target])
r.fit(50)
torch.cuda.synchronize()
ws, H, W, n = r._geom
big = torch.zeros(16 + 2 * 8192, dtype=torch.float64, device='cuda')
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    _lib.check(r.lib.sucre_fit_run(ws, H, W, n, 50 + rep, 1, 0.05, 0.9, 0.999, 1e-8, 0, C.c_void_p(big.data_ptr()), st))
torch.cuda.synchronize()
b = big.cpu().numpy()[16:].reshape(-1, 2)
b = b[b[:, 0] > 0]
t0 = b[:, 0].min()
start, end = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0   # us
life = end - start
print(os.path.basename(os.environ.get('SUCRE_HIP_LIB', 'default')), 'waves', len(b), 'kernel span %.1f us' % end.max(),
      'start: median %.1f max %.1f' % (np.median(start), start.max()),
      'life: min %.1f median %.1f p90 %.1f max %.1f' % (life.min(), np.median(life), np.percentile(life, 90), life.max()),
      'end: median %.1f p10 %.1f' % (np.median(end), np.percentile(end, 10)))
np.save(f"gpurun_out/exp_clock_{os.path.basename(os.environ.get('SUCRE_HIP_LIB', 'default'))}.npy", np.stack([start, end], 1))
