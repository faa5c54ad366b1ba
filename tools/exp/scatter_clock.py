#!/usr/bin/env python3
"""Why does scatter_kernel take 294 us in one process and 340 us in another?  With the experiment build
(make -C sucre_amd/csrc VARIANT=wavetimes EXTRA=-DSUCRE_EXP_WAVE_TIMES; SUCRE_HIP_LIB=...) the kernel reports the shader clock it
ran at; this script times the compaction of the headline image (HIP events around Restoration.match's second half are not
exposed, so: the whole match + finalize) a few times and prints the clock next to it.  Run it in several processes / on several
boxes: duration x clock is what stays the same."""
import ctypes
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import _lib, engine, synth  # noqa: E402

W, H, nn = 1920, 1080, 64
scene = synth.make_scene(W, H, nn, seed=0, device='cuda')
views = engine.device_views_from_scene(scene, 'cuda')
r = engine.Restoration(H, W, len(views))
lib = _lib.load()
fn = lib.sucre_exp_scatter_clock
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
warm = 'busy' in sys.argv[1:]
for rep in range(6):
    if warm:   # a burst of fit launches first: the power management's state when the compaction starts
        r.match(views[scene.target], views); r.fit_init(views[scene.target]); r.fit(100)
    torch.cuda.synchronize()
    buf = np.zeros(2, np.uint64)
    fn(buf.ctypes.data)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r.match(views[scene.target], views)
    e1.record(); torch.cuda.synchronize()
    assert fn(buf.ctypes.data) == 0
    cyc, tick = float(buf[0]), float(buf[1])
    print(f'match + finalize {e0.elapsed_time(e1) * 1e3:7.1f} us;  scatter_kernel: {tick / 8100 / 100:6.1f} us of workgroup time on average at {cyc / tick * 100:.0f} MHz', flush=True)
