#!/usr/bin/env python3
"""Randomised sweep of the batch launches (sucre_fit_run_batch) on the GPU box: random image sizes (ragged tiles), 1 .. 40 images
per batch (more than one launch holds), random view counts per image, both J modes, both store formats, random split of the
call sequence -- every image's trace, parameters and J must be BITWISE what Restoration.fit gives it alone.
    python3 tools/batch_sweep.py [n_batches] [seed]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from sucre_amd import engine, synth  # noqa: E402


def main():
    n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7000)
    t0 = time.perf_counter()
    images = 0
    for case in range(n_batches):
        W, H = int(rng.integers(33, 420)), int(rng.integers(33, 300))
        gx, gy = int(rng.integers(4, 9)), int(rng.integers(3, 7))
        n_img = int(rng.integers(1, min(41, gx * gy + 1)))
        closed = bool(rng.integers(0, 2))
        fmt = 'u16mm' if rng.integers(0, 4) == 0 else 'f32'
        T = int(rng.integers(1, 10))
        split = int(rng.integers(0, T))
        survey = synth.make_survey(W, H, gx, gy, seed=int(rng.integers(0, 100000)), device='cuda')
        views = engine.device_views_from_scene(survey, 'cuda')
        idxs = [int(i) for i in rng.choice(gx * gy, size=n_img, replace=False)]
        rs, tgts = [], []
        for idx in idxs:
            sel = survey.neighbours(idx, int(rng.integers(0, 12)))
            r = engine.Restoration(H, W, len(sel), obs_format=fmt)
            r.match(views[idx], [views[q] for q in sel])
            rs.append(r); tgts.append(views[idx])
        want = []
        for r, tgt in zip(rs, tgts):
            r.fit_init(tgt)
            tr = torch.cat([r.fit(split, use_closed_form=closed), r.fit(T - split, use_closed_form=closed)]) if split else r.fit(T, use_closed_form=closed)
            want.append((tr.cpu().numpy(), r.params().cpu().numpy().copy(), r.J().cpu().numpy()))
        for r, tgt in zip(rs, tgts):
            r.fit_init(tgt)
        if split:
            a, b = engine.fit_batch(rs, split, use_closed_form=closed), engine.fit_batch(rs, T - split, use_closed_form=closed)
            got = [torch.cat([x, y]) for x, y in zip(a, b)]
        else:
            got = engine.fit_batch(rs, T, use_closed_form=closed)
        torch.cuda.synchronize()
        for i, (r, (tr, p, J)) in enumerate(zip(rs, want)):
            tag = f'batch {case}: {W}x{H}, {n_img} images, closed={closed}, {fmt}, T={T}, split={split}, image {i} ({r.n_views} views, {r.n_obs()} obs)'
            assert np.array_equal(got[i].cpu().numpy(), tr, equal_nan=True), tag + ': trace'
            assert np.array_equal(r.params().cpu().numpy(), p, equal_nan=True), tag + ': parameters'
            assert np.array_equal(r.J().cpu().numpy(), J, equal_nan=True), tag + ': J'
        images += n_img
        del rs, views, survey
    print(f'{n_batches} batches, {images} images: every trace, parameter set and J bitwise the one-by-one fit ({time.perf_counter() - t0:.0f} s)')


if __name__ == '__main__':
    main()
