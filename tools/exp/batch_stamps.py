#!/usr/bin/env python3
"""Where ONE wave of a batch launch spends an image.  Needs the stamps of tools/exp/batch_stamps.patch in batch_iter_kernel /
stream_strips (git apply it; the stamped J-parameter kernel spills 20 bytes per lane, which the Makefile refuses: build fit.hip by
hand with the Makefile's flags + -DSUCRE_EXP_WAVE_TIMES and link the six objects into sucre_amd/libsucre_hip_stamps.so):
    SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_stamps.so python3 tools/exp/batch_stamps.py
Stamps: 1 image begins, 10 step begins, 13 the step's item has landed (after both waits), 4 pass done, 5 sums in LDS."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import _lib, engine, synth  # noqa: E402

scene = synth.make_scene(640, 480, 4, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
rs = []
for k in range(32):
    r = engine.Restoration(480, 640, len(views))
    r.match(views[scene.target], views)
    r.fit_init(views[scene.target])
    rs.append(r)
engine.fit_batch(rs, 6)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * (5 * 8192))()
lib.sucre_exp_wave_times.argtypes = [ctypes.c_void_p]
assert lib.sucre_exp_wave_times(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64)[:256]
a = a[a != 0]
ids = (a >> np.uint64(56)).astype(int)
t = (a & np.uint64((1 << 56) - 1)).astype(np.int64)
print(len(a), 'stamps; ticks are', 'shader clock (clock64)')
# split by image
starts = [i for i, d in enumerate(ids) if d == 1] + [len(ids)]
rows = []
for s, e in zip(starts[:-1], starts[1:]):
    seg_ids, seg_t = ids[s:e], t[s:e]
    line = []
    for j in range(1, len(seg_ids)):
        line.append(f'{seg_ids[j-1]}->{seg_ids[j]}:{seg_t[j]-seg_t[j-1]}')
    rows.append((seg_t[-1] - seg_t[0], ' '.join(line)))
for k, (tot, line) in enumerate(rows):
    print(f'image {k}: {tot} ticks | {line}')
if len(starts) > 2:
    per = np.diff(t[np.array(starts[:-1])])
    print('ticks between image starts:', per.tolist())
