#!/usr/bin/env python3
"""CPU-baseline calibration: the reference itself and the CPU oracle timed on the SAME inputs,
in the SAME container, with the SAME number of threads.

Build container only (the reference is imported through tests/golden/ref_harness.py and never travels):

    python tools/time_reference.py [--width 1920 --height 1080 --neighbours 4 --iters 3 --threads 8]

Times, on one synthetic image (synth.make_scene(W, H, n, seed 0): the target + its n nearest views):
  * the reference's ``match_two_way`` + depth gather over every view (sfm.py:121-125,137) and its ``sucre.adam`` for
    ``--iters`` iterations at ``batch_size=5`` (sucre.py:124-157), J-parameter mode and --use-closed-form;
  * the oracle (oracle/sucre_oracle.c) on exactly those inputs: ``match_view`` per view and ``fit`` for the same number
    of iterations.
Writes profiles/reference_cpu_calibration.json.  bench.py reads the ratios from it: its ``cpu_baseline`` leg times the
oracle (kind "port") on the GPU box's host cores and reports next to it ``reference_equivalent`` = that time multiplied
by the ratio measured here (labelled as such: the reference itself cannot be timed on the GPU box).
"""
from __future__ import annotations

import argparse
import contextlib
import datetime
import io
import json
import platform
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests'), str(ROOT / 'tests' / 'golden')]


def quiet(fn, *a, **k):
    with contextlib.redirect_stderr(io.StringIO()), contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main() -> None:
    p = argparse.ArgumentParser()
    p.add_argument('--width', type=int, default=1920)
    p.add_argument('--height', type=int, default=1080)
    p.add_argument('--neighbours', type=int, default=4)
    p.add_argument('--iters', type=int, default=3)
    p.add_argument('--threads', type=int, default=8)
    p.add_argument('--repeats', type=int, default=2, help='each leg is timed this many times; the best is kept')
    p.add_argument('--out', default=str(ROOT / 'profiles' / 'reference_cpu_calibration.json'))
    a = p.parse_args()

    import helpers
    import ref_harness as rh
    from oracle import oracle
    from sucre_amd import synth

    torch.set_num_threads(a.threads)
    oracle.set_num_threads(a.threads)
    scene = synth.make_scene(a.width, a.height, a.neighbours, seed=0)
    sfm, loader, _, sucre_mod = rh.import_reference()

    # --- the reference -------------------------------------------------------------------------------------------------
    best = {}

    def keep(key, dt):
        best[key] = min(best.get(key, float('inf')), dt)

    for _ in range(a.repeats):
        t0 = time.perf_counter()
        per_view, md, target = rh.reference_matches(scene)
        keep('ref_match_s', time.perf_counter() - t0)
    n_obs = len(md)
    for closed in (False, True):
        for _ in range(a.repeats):
            model = sucre_mod.SUCRe(image=target, use_closed_form=closed)
            t0 = time.perf_counter()
            quiet(sucre_mod.adam, model, md, lr=0.05, num_iter=a.iters, batch_size=5, device='cpu')
            keep('ref_fit_closed_s' if closed else 'ref_fit_param_s', time.perf_counter() - t0)

    # --- the oracle on the same inputs ------------------------------------------------------------------------------------
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        opv, samples = helpers.oracle_scene_samples(scene)
        keep('oracle_match_s', time.perf_counter() - t0)
    assert sum(len(s[0]) for s in samples) == n_obs, 'the oracle and the reference disagree on the observation count'
    tgt = scene.views[scene.target]
    J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    for closed in (False, True):
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            oracle.fit(scene.height, scene.width, samples, None if closed else J0, num_iter=a.iters, use_closed_form=closed)
            keep('oracle_fit_closed_s' if closed else 'oracle_fit_param_s', time.perf_counter() - t0)

    n_views = len(scene.views)
    per = lambda s: s / a.iters / n_obs * 1e9   # noqa: E731
    rec = {
        'what': "the reference's own sfm.Image.match_two_way + sucre.adam (torch CPU, batch_size=5) and the CPU oracle "
                '(oracle/sucre_oracle.c, OpenMP) timed on identical inputs in the build container; best of '
                f'{a.repeats} runs per leg',
        'workload': f'{a.width}x{a.height} target, {n_views} views ({a.neighbours} neighbours + self), {n_obs} observations, '
                    f'{a.iters} Adam iterations',
        'n_obs': int(n_obs), 'n_views': n_views, 'iters': a.iters, 'cores': a.threads,
        'torch_version': torch.__version__, 'python': platform.python_version(), 'machine': platform.processor() or platform.machine(),
        'date': datetime.date.today().isoformat(),
        'seconds': best,
        'ref_ns_per_obs_iter': per(best['ref_fit_param_s']),
        'oracle_ns_per_obs_iter': per(best['oracle_fit_param_s']),
        'fit_ratio': best['ref_fit_param_s'] / best['oracle_fit_param_s'],
        'ref_ns_per_obs_iter_closed': per(best['ref_fit_closed_s']),
        'oracle_ns_per_obs_iter_closed': per(best['oracle_fit_closed_s']),
        'fit_ratio_closed': best['ref_fit_closed_s'] / best['oracle_fit_closed_s'],
        'ref_match_s_per_view': best['ref_match_s'] / n_views,
        'oracle_match_s_per_view': best['oracle_match_s'] / n_views,
        'match_ratio': best['ref_match_s'] / best['oracle_match_s'],
        'note': 'ratio = reference time / oracle time on the same inputs and thread count; bench.py multiplies the '
                "oracle's time on the GPU box's host cores by it to state a reference-equivalent CPU baseline",
    }
    Path(a.out).write_text(json.dumps(rec, indent=1) + '\n')
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    main()
