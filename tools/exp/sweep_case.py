#!/usr/bin/env python3
"""Re-creates scene `s` of tools/parity_sweep.py's sequence for `seed0` and looks at the closed-form fit pixel by pixel.
usage: python3 tools/exp/sweep_case.py seed0 s"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import helpers
from oracle import oracle
from sucre_amd import engine, synth

seed0, want = int(sys.argv[1]), int(sys.argv[2])
closed = (sys.argv[3] if len(sys.argv) > 3 else 'closed') == 'closed'
rng = np.random.default_rng(seed0)
for s in range(want + 1):
    W, H = int(rng.integers(33, 260)), int(rng.integers(33, 200))
    nn = int(rng.integers(1, 14))
    kw = dict(relief=float(rng.choice([0.0, 0.15, 0.6])), spacing=float(rng.choice([0.05, 0.1, 0.25, 0.5])),
              invalid_frac=float(rng.choice([0.0, 0.01, 0.3])), rot_sigma=float(rng.choice([0.0, 0.03, 0.15])),
              pos_sigma=float(rng.choice([0.0, 0.1, 0.4])), far_views=int(rng.integers(0, 3)))
    T = int(rng.choice([3, 20, 60]))
print(W, H, nn, kw, T)
sc = synth.make_scene(W, H, nn, seed=seed0 + want, **kw)
per_view, samples = helpers.oracle_scene_samples(sc)
views = engine.device_views_from_scene(sc, 'cuda')
counts = np.zeros((H, W), int)
for u, v, cP, I in samples:
    np.add.at(counts, (v.astype(int), u.astype(int)), 1)
print('obs per pixel histogram', np.bincount(counts.ravel()))
for T in (1, 2, 3):
    r = engine.Restoration(H, W, len(views))
    r.match(views[sc.target], views)
    r.fit_init(views[sc.target])
    tr = r.fit(T, use_closed_form=closed).cpu().numpy()
    J = r.J().cpu().numpy()
    tgt = sc.views[sc.target]
    J0 = None if closed else oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    Jo, po, to = oracle.fit(H, W, samples, J0, num_iter=T, use_closed_form=closed)
    if not closed and T == 1:
        step = np.abs(Jo - J0) / 0.05
        print('   first Adam step of J relative to lr: fraction of pixels-channels below 0.99:', float(np.nanmean(step < 0.99)), 'below 0.5:', float(np.nanmean(step < 0.5)))
    d = np.abs(J - Jo)
    print(f'T={T}: rms {helpers.rms_per_channel(J, Jo)}  max |dJ| {np.nanmax(d):.3e}  params diff {np.abs(tr[:, 1:] - to[:, 1:]).max():.3e} cost rel {np.abs(tr[:, 0] / to[:, 0] - 1).max():.3e}')
    idx = np.argsort(np.nan_to_num(d).max(axis=2).ravel())[::-1][:6]
    for i in idx:
        y, x = divmod(int(i), W)
        zs = [float(np.linalg.norm(cP[:, (u == x) & (v == y)].astype(np.float64), axis=0)[0]) for u, v, cP, I in samples if ((u == x) & (v == y)).any()]
        print(f'   pixel ({x},{y}) obs {counts[y, x]} z {np.round(zs, 3)}  J hip {J[y, x]} oracle {Jo[y, x]}')
    print('   params hip', tr[-1, 1:], '\n   params ora', to[-1, 1:])
