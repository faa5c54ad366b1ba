import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np, torch, helpers
from oracle import oracle
from sucre_amd import engine, synth
seed0, target_s = 61000, 157
rng = np.random.default_rng(seed0)
for s in range(target_s + 1):
    W, H = int(rng.integers(33, 260)), int(rng.integers(33, 200))
    nn = int(rng.integers(1, 14))
    kw = dict(relief=float(rng.choice([0.0, 0.15, 0.6])), spacing=float(rng.choice([0.05, 0.1, 0.25, 0.5])),
              invalid_frac=float(rng.choice([0.0, 0.01, 0.3])), rot_sigma=float(rng.choice([0.0, 0.03, 0.15])),
              pos_sigma=float(rng.choice([0.0, 0.1, 0.4])), far_views=int(rng.integers(0, 3)))
    T = int(rng.choice([3, 20, 60]))
print('scene', W, H, nn, kw, 'T', T)
sc = synth.make_scene(W, H, nn, seed=seed0 + target_s, **kw)
per_view, samples = helpers.oracle_scene_samples(sc)
views = engine.device_views_from_scene(sc, 'cuda')
tgt = sc.views[sc.target]
for fmt in ('f32', 'u16mm'):
    r = engine.Restoration(H, W, len(views), obs_format=fmt)
    r.match(views[sc.target], views)
    smp = samples if fmt == 'f32' else oracle.quantize_ranges_u16mm(samples)
    r.fit_init(views[sc.target])
    tr = r.fit(T).cpu().numpy()
    J = r.J().cpu().numpy()
    J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    Jo, po, to = oracle.fit(H, W, smp, J0, num_iter=T)
    err = np.nan_to_num(np.abs(J - Jo)).max(axis=2)
    top = np.sort(err.ravel())[-6:]
    print(fmt, 'rms', helpers.rms_per_channel(J, Jo), 'top errors', top, 'max dparam', np.abs(tr[:, 1:] - to[:, 1:]).max(), 'n_obs', r.n_obs())
    v, u = np.unravel_index(err.argmax(), err.shape)
    cnt = sum(int(((s_[0] == u) & (s_[1] == v)).sum()) for s_ in smp)
    print('  worst pixel', u, v, 'observations', cnt, 'J engine', J[v, u], 'oracle', Jo[v, u])
