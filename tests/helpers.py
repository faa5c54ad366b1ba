"""Shared test plumbing: scenes -> oracle inputs, fixtures, comparison metrics."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from oracle import oracle  # noqa: E402  (tests are allowed to use the oracle)

GOLDEN_DIR = Path(__file__).resolve().parent / 'golden'


def cam_matrices(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor):
    """float32 matrices exactly as the reference derives them: K.inverse() (sfm.py:92), Pose.inverse()
    = (R.T, -R.T @ t) (sfm.py:42-47)."""
    Kinv = K.inverse()
    Rinv = R.T
    tinv = -R.T @ t
    return dict(K=K.numpy(), Kinv=Kinv.numpy(), R=R.numpy(), t=t.numpy().ravel(),
                Rinv=Rinv.contiguous().numpy(), tinv=tinv.numpy().ravel())


def oracle_cam(scene, view):
    m = cam_matrices(scene.K, view.R, view.t)
    return oracle.make_cam(scene.height, scene.width, **m)


def oracle_scene_samples(scene, min_cover: float = 1e-6):
    """Oracle restatement of match_images + prepare/load_matches for the scene's target.

    Returns (per_view list of (name, kept, ViewMatches), samples list of (u1, v1, cP, I) for kept views in
    name order)."""
    tgt = scene.views[scene.target]
    cam1 = oracle_cam(scene, tgt)
    d1 = tgt.depth_f32().numpy()
    per_view, samples = [], []
    for view in scene.views:
        cam2 = oracle_cam(scene, view)
        m = oracle.match_view(d1, cam1, view.depth_f32().numpy(), cam2)
        kept = len(m) / (scene.width * scene.height) > min_cover
        per_view.append((view.name, kept, m))
    for (name, kept, m), view in sorted(zip(per_view, scene.views), key=lambda p: p[0][0]):
        if not kept:
            continue
        cP = oracle.unproject(oracle_cam(scene, view), m.u2, m.v2, m.d)
        I = oracle.gather_rgb(view.rgb_u8.numpy(), m.u2, m.v2)
        samples.append((m.u1, m.v1, cP, I))
    return per_view, samples


def rms_per_channel(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Per-channel RMS over pixels finite in both; NaN masks must agree (asserted by callers)."""
    ok = np.isfinite(a).all(axis=-1) & np.isfinite(b).all(axis=-1)
    d = (a[ok].astype(np.float64) - b[ok].astype(np.float64))
    return np.sqrt((d * d).mean(axis=0))


class Fixture:
    """A committed golden vector: synthetic inputs + what the reference computed from them
    (tests/golden/gen_golden.py)."""

    def __init__(self, name: str):
        from sucre_amd import synth
        self.name = name
        z = np.load(GOLDEN_DIR / f'{name}.npz')
        self.z = z
        W, H = int(z['width']), int(z['height'])
        views = []
        for i, nm in enumerate(z['names']):
            views.append(synth.SynthView(name=str(nm), R=torch.tensor(z['R'][i]), t=torch.tensor(z['t'][i]),
                                         depth_u16=torch.tensor(z['depth_u16'][i].astype(np.int32)),
                                         rgb_u8=torch.tensor(z['rgb_u8'][i])))
        self.scene = synth.SynthScene(width=W, height=H, K=torch.tensor(z['K']), views=views,
                                      target=int(z['target']), seed=int(z['seed']))

    def __getitem__(self, key):
        return self.z[key]

    def match_lists(self, k: int):
        """(u1, v1, u2, v2) of view k in torch.where order, from the dense match map."""
        m = self.z['match_map'][k]
        v1, u1 = np.nonzero(m >= 0)
        q = m[v1, u1]
        W = self.scene.width
        return u1.astype(np.int16), v1.astype(np.int16), (q % W).astype(np.int16), (q // W).astype(np.int16)


_FIXTURES = {}


def load_fixture(name: str) -> Fixture:
    if name not in _FIXTURES:
        _FIXTURES[name] = Fixture(name)
    return _FIXTURES[name]


# ---- closed-form "knee" scenes (tests/golden/gen_golden_extras.py) --------------------------------------------------------

KNEE_FIXTURES = ['knee_190x51_n1', 'knee_215x74_n1', 'knee_225x87_n1', 'knee_115x67_n5']


def knee_bars(fx):
    """The reference against ITSELF on a knee scene (batch_size 1 vs 5: only the summation order differs): the spread
    of the water parameters over the trajectory and of the final J.  A restatement is 'as good as the reference' when
    it lies within a small multiple of that spread; the cost of iteration 0 has no step behind it and stays tight."""
    t1, t5 = fx['trace_closed_bs1'], fx['trace_closed_bs5']
    par = float(np.abs(t1[:, 1:] - t5[:, 1:]).max())
    Jsp = float(rms_per_channel(fx['J_closed_bs1'], fx['J_closed_bs5']).max())
    return par, Jsp


def check_knee(fx, J, trace, label):
    t1, t5 = fx['trace_closed_bs1'], fx['trace_closed_bs5']
    par, Jsp = knee_bars(fx)
    # this IS a knee scene: some water parameter's first Adam step is visibly shorter than lr (|g| ~ eps)
    assert np.any(np.abs(t5[0, 1:] - 0.1) / 0.05 < 0.99)
    assert abs(trace[0, 0] / t5[0, 0] - 1) < 1e-6, (label, 'cost of iteration 0')
    d = min(float(np.abs(trace[:, 1:] - t5[:, 1:]).max()), float(np.abs(trace[:, 1:] - t1[:, 1:]).max()))
    assert d < 3 * par, (label, 'parameters', d, par)
    assert np.array_equal(np.isnan(J), np.isnan(fx['J_closed_bs5'])), label
    dJ = min(float(rms_per_channel(J, fx['J_closed_bs5']).max()), float(rms_per_channel(J, fx['J_closed_bs1']).max()))
    assert dJ < 3 * Jsp, (label, 'J', dJ, Jsp)
    return d / par, dJ / Jsp
