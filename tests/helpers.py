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


def cam_matrices(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor, Kinv=None, tinv=None):
    """float32 matrices exactly as the reference derives them: K.inverse() (sfm.py:92), Pose.inverse()
    = (R.T, -R.T @ t) (sfm.py:42-47) -- on THIS host, unless the caller brings the ones another host derived (MKL's
    float32 products differ in the last bit between CPU models; helpers.Baseline)."""
    Kinv = K.inverse() if Kinv is None else Kinv
    Rinv = R.T
    tinv = -R.T @ t if tinv is None else tinv
    return dict(K=K.numpy(), Kinv=Kinv.numpy(), R=R.numpy(), t=t.numpy().ravel(),
                Rinv=Rinv.contiguous().numpy(), tinv=tinv.numpy().ravel())


def oracle_cam(scene, view):
    m = cam_matrices(scene.K, view.R, view.t, Kinv=getattr(scene, 'Kinv_given', None), tinv=getattr(view, 'tinv_given', None))
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
    it lies within a small multiple of that spread; the cost of iteration 0 has no step behind it and stays tight.
    The multiple (KNEE_FACTOR) is 1.5 x the spread, measured from the NEARER of the two reference runs: the oracle sits at
    0.67 .. 1.38 x and the engine at 0.33 .. 1.19 x on the four scenes (3 x until round 4, when the reference-made goldens
    at BASELINE sizes became the anchor for everything that is not a knee)."""
    t1, t5 = fx['trace_closed_bs1'], fx['trace_closed_bs5']
    par = float(np.abs(t1[:, 1:] - t5[:, 1:]).max())
    Jsp = float(rms_per_channel(fx['J_closed_bs1'], fx['J_closed_bs5']).max())
    return par, Jsp


KNEE_FACTOR = 1.5


def check_knee(fx, J, trace, label):
    t1, t5 = fx['trace_closed_bs1'], fx['trace_closed_bs5']
    par, Jsp = knee_bars(fx)
    # this IS a knee scene: some water parameter's first Adam step is visibly shorter than lr (|g| ~ eps)
    assert np.any(np.abs(t5[0, 1:] - 0.1) / 0.05 < 0.99)
    assert abs(trace[0, 0] / t5[0, 0] - 1) < 1e-6, (label, 'cost of iteration 0')
    d = min(float(np.abs(trace[:, 1:] - t5[:, 1:]).max()), float(np.abs(trace[:, 1:] - t1[:, 1:]).max()))
    assert d < KNEE_FACTOR * par, (label, 'parameters', d, par)
    assert np.array_equal(np.isnan(J), np.isnan(fx['J_closed_bs5'])), label
    dJ = min(float(rms_per_channel(J, fx['J_closed_bs5']).max()), float(rms_per_channel(J, fx['J_closed_bs1']).max()))
    assert dJ < KNEE_FACTOR * Jsp, (label, 'J', dJ, Jsp)
    return d / par, dJ / Jsp


# ---- reference-made goldens at BASELINE.json's own sizes (tests/golden/gen_golden_baseline.py) ---------------------------

BASELINE_C1 = 'baseline_c1_640x480_n4'
BASELINE_C2 = 'baseline_c2_1920x1080_n64'
BASELINE_C2FULL = 'baseline_c2full_1920x1080_n64'
BASELINE_C5VIEWS = 'baseline_c5views_480x360_n256'
BASELINE_ODD = 'baseline_odd_333x207_n8'
BASELINE_DEEP = 'baseline_deep_640x480_n8'   # ranges 0.72 .. 8.03 m (synth.make_deep_scene)


def scene_digests(scene):
    """SHA-256 per view over (uint16 depth plane, uint8 colour plane) + one over all views, poses and K -- the
    restatement of ``input_digests`` of tests/golden/gen_golden_baseline.py (which cannot be imported on the GPU box's
    side of things without the reference harness)."""
    import hashlib
    per_view, h_all = [], hashlib.sha256()
    for v in scene.views:
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(v.depth_u16.cpu().numpy().astype(np.uint16)).tobytes())
        h.update(np.ascontiguousarray(v.rgb_u8.cpu().numpy()).tobytes())
        per_view.append(h.hexdigest())
        h_all.update(h.digest())
        h_all.update(np.ascontiguousarray(v.R.numpy()).tobytes())
        h_all.update(np.ascontiguousarray(v.t.numpy()).tobytes())
    h_all.update(np.ascontiguousarray(scene.K.numpy()).tobytes())
    return per_view, h_all.hexdigest()


def match_map_digest(match_map: np.ndarray) -> str:
    """SHA-256 of a dense (H,W) int32 match map (q = v2*W + u2 at (v1,u1), -1 elsewhere)."""
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(match_map, dtype=np.int32).tobytes()).hexdigest()


_SCENES = {}


class SceneMismatch(AssertionError):
    """The regenerated synthetic scene is not the one the reference ran on."""


def _make_scene(width, height, n_neighbours, seed, kind='plain'):
    """The seeded scene of a baseline fixture.  ``SUCRE_TEST_PERTURB_SCENE=<view index>`` flips ONE colour byte of ONE view
    after rendering -- the smallest way a host could render another scene: every reference-pinned test must then be RED
    (``SUCRE_TEST_PERTURB_SCENE=0 pytest -m gpu`` shows it; tests/test_host_logic.py asserts it on the CPU tier)."""
    import os

    from sucre_amd import synth
    scene = (synth.make_deep_scene if kind == 'deep' else synth.make_scene)(width, height, n_neighbours, seed=seed)
    k = os.environ.get('SUCRE_TEST_PERTURB_SCENE')
    if k not in (None, ''):
        scene.views[int(k) % len(scene.views)].rgb_u8[height // 2, width // 2, 1] ^= 1
    return scene


class Baseline:
    """A reference-made golden at a BASELINE size: OUTPUTS only; the inputs are regenerated here (on the CPU, like the
    generator did) and checked against the stored SHA-256 digests.  A host that renders another scene (a libm that rounds
    one pixel of one view the other way) gets a RED test that names the differing views -- never a looser bar: every
    bar below is held under bit-identical inputs or not at all (VERDICT round 4, weak point 1)."""

    def __init__(self, name: str):
        from sucre_amd import synth
        self.name = name
        self.z = np.load(GOLDEN_DIR / f'{name}.npz')
        z = self.z
        skey = (int(z['width']), int(z['height']), int(z['n_neighbours']), int(z['seed']), str(z['scene_kind']) if 'scene_kind' in z else 'plain')
        if skey not in _SCENES:   # several fixtures hold results on the same scene (config 2: short, in full, light model)
            _SCENES.clear()       # ... one at a time: a config-2 scene is 0.7 GB
            _SCENES[skey] = _make_scene(*skey)
        self.scene = _SCENES[skey]
        assert self.scene.names == [str(n) for n in z['names']] and self.scene.target == int(z['target'])
        # The float32 matrices the reference derived from K, R, t on ITS host (torch CPU: K.inverse(), -R.T @ t).  They are
        # handed to the oracle and to the engine, so the per-pixel arithmetic is compared under the reference's own
        # matrices on whatever CPU this runs (the build container's and the GPU box's MKL differ in the last bit of
        # -R.T @ t, which moves ~7 of a million matches of a view).  ``derived_identical``: would this host's have been?
        own_Kinv = self.scene.K.inverse().numpy()
        own_tinv = np.stack([(-v.R.T @ v.t).numpy().ravel() for v in self.scene.views])
        self.derived_identical = bool(np.array_equal(own_Kinv, z['Kinv']) and np.array_equal(own_tinv, z['tinv']))
        self.scene.Kinv_given = torch.tensor(z['Kinv'])
        for v, ti in zip(self.scene.views, z['tinv']):
            v.tinv_given = torch.tensor(ti).view(3, 1)
        if not self.derived_identical:
            print(f'NOTE {name}: this host derives other float32 camera matrices than the reference\'s host did '
                  f'({int((own_tinv != z["tinv"]).any(axis=1).sum())} of {len(own_tinv)} views); the stored ones are used')
        per_view, total = scene_digests(self.scene)
        self.views_identical = [a == str(b) for a, b in zip(per_view, z['input_digest_per_view'])]
        self.inputs_identical = total == str(z['input_digest'])
        if not self.inputs_identical:
            _SCENES.pop(skey, None)
            bad = [self.scene.views[k].name for k, same in enumerate(self.views_identical) if not same]
            raise SceneMismatch(
                f'{name}: the scene regenerated on this host is NOT the one the reference ran on: '
                f'{len(bad)} of {len(per_view)} views differ in their uint16 depth / uint8 colour planes '
                f'({", ".join(bad[:8])}{", ..." if len(bad) > 8 else ""})'
                + ('' if bad else '; the pixel planes agree, so the poses or K differ')
                + ' -- no reference-pinned bar can be held on other inputs')

    def __getitem__(self, key):
        return self.z[key]

    def j_sums(self, J: np.ndarray):
        """(nan_count, per-channel float64 sum, sum of squares) of a full (H,W,3) image, as the generator stored them."""
        ok = np.isfinite(J).all(axis=-1)
        Jd = J[ok].astype(np.float64)
        return int((~ok).sum()), Jd.sum(axis=0), (Jd * Jd).sum(axis=0)


_BASELINES = {}


def load_baseline(name: str) -> Baseline:
    if name not in _BASELINES:
        _BASELINES[name] = Baseline(name)
    return _BASELINES[name]


# ---- what the reference-pinned checks MEASURED, for the run's terminal summary (tests/conftest.py) ------------------------
# One row per check: label, key, per-channel RMS(J), max |d params|, max rel d cost, whether the regenerated inputs and
# this host's derived camera matrices were the reference's, how many match maps were compared.  pytest -q swallows the
# prints; the summary hook does not, so the driver's GPUTEST record carries numbers instead of dots.
PARITY_ROWS: list = []


def record_parity(label, key, b=None, **kw):
    row = dict(label=label, key=key, **kw)
    if b is not None:
        row.update(fixture=b.name, inputs_identical=b.inputs_identical, derived_identical=b.derived_identical)
    PARITY_ROWS.append(row)


def parity_summary_lines(max_lines: int = 25):
    """<= max_lines lines: a header, then one line per (who, fixture, mode) -- rows of the same mode (the two images and the two
    paths of a shared-water check, repeated runs) merged by their WORST value; the match-map verdict of the fixture rides on
    each of its lines."""
    if not PARITY_ROWS:
        return []

    def worst(a, b):
        if a is None:
            return b
        if b is None:
            return a
        return np.maximum(a, b)

    def fmt(x):
        return '    -   ' if x is None else f'{float(x):8.1e}'
    groups, maps = {}, {}
    for r in PARITY_ROWS:
        who = 'engine' if 'engine' in r['label'].lower() else 'oracle'
        fx = r.get('fixture', '?').replace('baseline_', '')
        key = r['key']
        if key == 'matches':
            maps[(who, fx)] = f"n {r['counts']} maps {r['maps']}"
            continue
        key = 'shared' if key.startswith('shared') else 'param@1' if key == 'param_1' else key
        g = groups.setdefault((who, fx, key), dict(rms=None, dpar=None, dcost=None, dlight=None, T=None, n=0,
                                                   inputs=True, derived=True))
        for f in ('rms', 'dpar', 'dcost', 'dlight'):
            g[f] = worst(g[f], r.get(f))
        g['T'] = r.get('T') or g['T']
        g['n'] += 1
        g['inputs'] &= bool(r.get('inputs_identical', True))
        g['derived'] &= bool(r.get('derived_identical', True))
    lines = [f"{'who':6s} {'reference-made fixture':22s} {'mode':12s} {'its':>3s} {'RMS(J) r':>8s} {'g':>8s} {'b':>8s} "
             f"{'|dparam|':>8s} {'rel dcost':>9s} {'|dlight|':>8s} inputs matrices match sets identical"]
    for (who, fx, key), g in groups.items():
        rms = np.atleast_1d(g['rms']) if g['rms'] is not None else [None] * 3
        lines.append(f"{who:6s} {fx:22s} {key:12s} {str(g['T'] or '-'):>3s} " + ' '.join(fmt(x) for x in rms)
                     + f" {fmt(g['dpar'])} {fmt(g['dcost']):>9s} {fmt(g['dlight'])} "
                     f"{'same' if g['inputs'] else 'DIFF':6s} {'own' if g['derived'] else 'stored':8s} {maps.get((who, fx), '-')}")
    if len(lines) > max_lines:
        body = lines[1:]
        body.sort(key=lambda ln: ('param@1' in ln, 'oracle' in ln[:6]))   # the engine's full runs first
        lines = [lines[0]] + body[:max_lines - 2] + [f'... {len(body) - (max_lines - 2)} more rows not shown']
    return lines


def dense_map(m, H, W):
    """Dense (H,W) int32 match map of an oracle ViewMatches."""
    mm = np.full((H, W), -1, np.int32)
    mm[m.v1.astype(np.int64), m.u1.astype(np.int64)] = m.v2.astype(np.int32) * W + m.u2.astype(np.int32)
    return mm


def check_baseline_matches(b, counts, maps, label):
    """Per-view match counts and dense match maps against the reference's: bit-exact (count + SHA-256 of the map), every
    view (the inputs are the reference's: ``Baseline`` refuses to exist otherwise)."""
    assert b.inputs_identical and all(b.views_identical)
    ref_counts = b['n_matches'].tolist()
    assert len(counts) == len(ref_counts), (label, 'number of views')
    n_maps = 0
    for k, n in enumerate(counts):
        assert n == ref_counts[k], (label, 'count of view', k, n, ref_counts[k])
        if maps is not None and maps[k] is not None:
            assert match_map_digest(maps[k]) == str(b['match_digest'][k]), (label, 'match map of view', k)
            n_maps += 1
    record_parity(label, 'matches', b, counts=f'{len(counts)}/{len(ref_counts)}',
                  maps=f'{n_maps}/{len(ref_counts)}' if maps is not None else '-')


def check_baseline_fit(b, key, J, trace, rms_bar, param_bar, cost_bar, label, trace_key=None, light_bar=None, mode=None):
    """A full (H,W,3) J and a (T, >=10) trace against what the REFERENCE stored: J[::stride, ::stride], the NaN count
    and the per-channel sums of J and J^2 over the whole image, the (T,10) cost / B / beta / gamma trajectory."""
    st = max(int(b['stride']), 4) if key == 'param_1' else int(b['stride'])
    assert b.inputs_identical
    ref = b[f'J_{key}']
    sub = J[::st, ::st]
    nan_count, s, sq = b.j_sums(J)
    assert np.array_equal(np.isnan(sub), np.isnan(ref)), (label, 'NaN mask')
    assert nan_count == int(b[f'J_{key}_nan_count']), (label, 'NaN count of the whole image')
    rms = rms_per_channel(sub, ref)
    n_ok = J.shape[0] * J.shape[1] - nan_count
    dmean = np.abs(s - b[f'J_{key}_sum']) / n_ok         # every pixel of the image enters these two
    dsq = np.abs(sq - b[f'J_{key}_sqsum']) / n_ok
    out = dict(rms=rms, dmean=dmean, dsq=dsq)
    assert rms.max() < rms_bar, (label, key, 'RMS(J) vs the reference', rms)
    assert dmean.max() < rms_bar and dsq.max() < rms_bar, (label, key, 'whole-image mean / mean square', dmean, dsq)
    if trace is not None:
        rt = b[trace_key or ('trace_closed' if key == 'closed' else 'trace_param')][:trace.shape[0]]
        out['dpar'] = float(np.abs(trace[:, 1:10] - rt[:, 1:10]).max())
        out['dcost'] = float(np.abs(trace[:, 0] / rt[:, 0] - 1).max())
        if light_bar is not None:   # cam2light, sigma: their gradients sit at Adam's eps (the reference's own batch-order noise: 1e-3)
            out['dlight'] = float(np.abs(trace[:, 10:] - rt[:, 10:]).max())
            assert out['dlight'] < light_bar, (label, key, 'light parameters', out['dlight'])
        assert out['dpar'] < param_bar, (label, key, 'B, beta, gamma trajectory', out['dpar'])
        assert out['dcost'] < cost_bar, (label, key, 'cost trajectory', out['dcost'])
    print(f'{label} [{key}] vs the REFERENCE: rms(J)={rms} |dmean|={dmean.max():.2e} '
          f'max|dparams|={out.get("dpar", float("nan")):.2e} max rel dcost={out.get("dcost", float("nan")):.2e}')
    record_parity(label, mode or key, b, rms=rms, dpar=out.get('dpar'), dcost=out.get('dcost'), dlight=out.get('dlight'),
                  T=None if trace is None else int(trace.shape[0]), bars=(rms_bar, param_bar, cost_bar))
    return out


def synth_image(idx, view, K, W, H):
    """sfm.Image whose pixels come from a synthetic view instead of files (what tests/golden/ref_harness.py does to the
    reference's class)."""
    from sucre_amd import sfm

    class _Image(sfm.Image):
        def get_rgb(self):
            return view.rgb_f32()

        def get_depth_map(self):
            return view.depth_f32()
    return _Image(idx, Path(view.name), Path('depth_' + view.name), sfm.Pose(view.R, view.t), sfm.Camera(1, W, H, K))
