#!/usr/bin/env python3
"""How ragged is the end of a fit launch?  Needs the experiment build that records, per wave, when it entered and left its
strips (100 MHz wall clock):

    make -C sucre_amd/csrc VARIANT=wavetimes EXTRA=-DSUCRE_EXP_WAVE_TIMES          (+ -DSUCRE_DEAL_FIT=... to try other shares)
    SUCRE_HIP_LIB=$PWD/sucre_amd/libsucre_hip_wavetimes.so python3 tools/exp/wave_times.py [closed] [W H NN]

Runs one image (default: the headline one, 1920x1080, 64 neighbours + self) alone on the GPU, reads the LAST launch's times and
prints: the launch's span, when the waves ended (percentiles), the mean busy time of a wave as a share of the span (what a
perfectly even end would make of the launch), and the same per workgroup GENERATION (workgroup b is the (b / 256)-th arrival
on its CU; the SIMD issues the oldest ready wave first -- layout.h, "The deal") and per XCD (workgroup b runs on XCD b % 8)."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import _lib, engine, synth  # noqa: E402

args = sys.argv[1:]
closed = 'closed' in args
nums = [int(a) for a in args if a.isdigit()]
W, H, nn = (nums + [1920, 1080, 64][len(nums):])[:3]
scene = synth.make_scene(W, H, nn, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(H, W, len(views))
r.match(views[scene.target], views)
r.fit_init(views[scene.target])
lib = _lib.load()
fn = lib.sucre_exp_wave_times
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
G = 4 if closed else 5
n = 1024 * G
pct = lambda x: ' '.join(f'{np.percentile(x, p):6.1f}' for p in (0, 10, 50, 90, 99, 100))
iters = [int(a[6:]) for a in args if a.startswith('iters=')]
for rep in range(3):
    r.fit(iters[0] if iters else 6, use_closed_form=closed)
    torch.cuda.synchronize()
    buf = np.zeros((5, 8192), np.uint64)
    assert fn(buf.ctypes.data) == 0
    t0, t1 = buf[0, :n].astype(np.int64), buf[1, :n].astype(np.int64)
    base = t0.min()
    s, e = (t0 - base) / 100.0, (t1 - base) / 100.0     # microseconds
    span = e.max()
    busy = e - s
    if rep < 2:
        continue
    print(f'{"closed form" if closed else "J parameter"}, {W}x{H}x{nn + 1}: span {span:.1f} us (first wave in -> last wave out), starts within {s.max():.1f} us')
    print(f'   ends    p0/10/50/90/99/100: {pct(e)}')
    print(f'   busy    mean {busy.mean():.1f} us = {busy.mean() / span:.3f} of the span')
    gen = np.arange(n) // 1024
    print('   per generation  mean end: ' + ' '.join(f'{e[gen == g].mean():6.1f}' for g in range(G)) + '   last end: ' + ' '.join(f'{e[gen == g].max():6.1f}' for g in range(G)))
    if not closed:
        tin, tout = (buf[2, :n].astype(np.int64) - base) / 100.0, (buf[3, :n].astype(np.int64) - base) / 100.0
        mhz = buf[4, :n].astype(np.float64) / (buf[3, :n].astype(np.float64) - buf[2, :n].astype(np.float64)) * 100.0
        print(f'   shader clock {np.median(mhz):.0f} MHz (median over the waves)')
        print(f'   kernel entered p0/50/100: {np.percentile(tin, 0):6.1f} {np.percentile(tin, 50):6.1f} {tin.max():6.1f};  left p0/10/50/90/99/100: {pct(tout)};  '
              f'first in -> last out {tout.max() - tin.min():.1f} us;  after the last wave\'s strips: {tout.max() - span:.1f} us')
    xcd = (np.arange(n) // 4) % 8
    print('   per XCD         mean end: ' + ' '.join(f'{e[xcd == x].mean():6.1f}' for x in range(8)), flush=True)
