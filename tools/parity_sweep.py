#!/usr/bin/env python3
"""Randomised parity sweep on the GPU box: many small scenes (odd sizes, few/many views, steep relief, large twist
noise, many invalid pixels, far views) -- matching must be bit-identical to the CPU oracle and the fit must stay
within the test tolerances, in every mode.  Not a test (too long for the suite); run by hand:
    python3 tools/parity_sweep.py [n_scenes] [seed0] [max_width max_height]
environment: SWEEP_MAX_NEIGHBOURS (13), SWEEP_LIGHT (1), SWEEP_FCOLOUR (1: float32 colours off the 1/255 grid), SWEEP_IMPORT (1: the kept views' lists through sucre_import_view), SWEEP_ONLY=<scene index> (one scene of the sequence, with traces)"""
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import helpers  # noqa: E402
from oracle import oracle  # noqa: E402
from sucre_amd import engine, synth  # noqa: E402


def overflow_run(tr, to, tag, worst, zmax=0.0, closed=True):
    """True when the run is outside the domain where float32 trajectories are comparable: ranges so long that
    a^2 = exp(-2 beta z) leaves the normal float32 numbers (beta z > 40: hundreds of metres).  A pixel whose sum(a^2) is a
    DENORMAL number has a closed-form J with a few bits of precision -- rounding noise in the reference itself (its
    float32 operation order decides; the engine's one-pass algebra has another) -- and once every a^2
    of an observed pixel is zero, J is +-inf, the cost is inf and the channel's parameters are NaN from the next step on.
    Such a run is held to its first iteration and, if it was still on the oracle's trajectory when the oracle died, to dying
    in the same iteration and the same columns with an infinite cost."""
    finite = np.isfinite(to).all()
    beta_max = float(np.nanmax(np.abs(np.where(np.isfinite(to[:, 4:7]), to[:, 4:7], 0.0)))) if to.size else 0.0
    if finite and (not closed or zmax * beta_max <= 40.0):   # (with J as a parameter nothing is divided by sum(a^2))
        return False
    first = int((~np.isfinite(to)).any(axis=1).argmax()) if not finite else len(to)
    if os.environ.get('SWEEP_ONLY'):
        fe = int((~np.isfinite(tr)).any(axis=1).argmax()) if not np.isfinite(tr).all() else -1
        print(tag, 'far-range run: zmax', zmax, 'beta max', beta_max, 'oracle first non-finite iteration', first, 'engine', fe,
              'costs: oracle', np.array2string(to[:first + 1, 0], precision=3), 'engine', np.array2string(tr[:first + 1, 0], precision=3), sep='\n')
    assert abs(tr[0, 0] - to[0, 0]) < 1e-4 * abs(to[0, 0]) + 1e-9 and np.abs(tr[0, 1:] - to[0, 1:]).max() < 1e-5, (tag, 'first iteration of a far-range run')
    on_track = first > 0 and np.abs(tr[:first, 1:] - to[:first, 1:]).max() < 1e-5
    if not finite and on_track:
        worst['overflow_runs'] = worst.get('overflow_runs', 0) + 1
        assert int((~np.isfinite(tr)).any(axis=1).argmax()) == first and not np.isfinite(tr).all(), (tag, 'first non-finite iteration')
        assert np.array_equal(np.isfinite(tr[first]), np.isfinite(to[first])) and np.array_equal(np.isinf(tr[first]), np.isinf(to[first])), \
            (tag, 'what died first', tr[first], to[first])
    else:
        worst['far_runs'] = worst.get('far_runs', 0) + 1
    return True


def check_light(rl, views, target, H, W, samples, make_J0, tag, worst):
    """The artificial-light model of one scene (J-parameter and closed-form) against the oracle: the first iteration's cost
    has no step behind it and is tight; the trajectory is ill-conditioned by construction (the cam2light gradients sit at
    Adam's eps), so it is held to the bounds of tests/test_gpu_parity.py over a few steps."""
    rl.match(views[target], views)
    assert rl.n_obs() == sum(len(x[0]) for x in samples), (tag, 'light n_obs')
    Tl = 4
    for closed in (False, True):
        rl.fit_init(views[target])
        trl = rl.fit(Tl, use_closed_form=closed).cpu().numpy()
        Jl = rl.J().cpu().numpy()
        Jo, po, to = oracle.fit_light(H, W, samples, None if closed else make_J0(), num_iter=Tl, use_closed_form=closed)
        if os.environ.get('SWEEP_ONLY') and not np.array_equal(np.isnan(Jl), np.isnan(Jo)):
            d = np.isnan(Jl) != np.isnan(Jo)
            print('NaN masks differ at', int(d.sum()), 'of', d.size, 'engine NaNs', int(np.isnan(Jl).sum()), 'oracle NaNs', int(np.isnan(Jo).sum()),
                  'engine there', Jl[d][:6], 'oracle there', Jo[d][:6], 'traces', trl[:, :4], to[:, :4], sep='\n')
        assert np.array_equal(np.isnan(Jl), np.isnan(Jo)), (tag, 'light nan mask', closed)
        if not np.isfinite(to[0, 0]):   # the re-solved J of a pixel lit by almost nothing overflows float32: in both
            assert trl[0, 0] == to[0, 0] or (np.isnan(trl[0, 0]) and np.isnan(to[0, 0])), (tag, 'light cost 0', closed, trl[0, 0], to[0, 0])
            worst['light_overflow'] = worst.get('light_overflow', 0) + 1
            continue
        assert abs(trl[0, 0] - to[0, 0]) < 1e-5 * to[0, 0] + 1e-9, (tag, 'light cost 0', closed, trl[0, 0], to[0, 0])
        worst['light_cost0'] = max(worst.get('light_cost0', 0.0), abs(trl[0, 0] - to[0, 0]) / max(to[0, 0], 1e-12))
        knee = bool(np.any(np.abs(to[0, 1:10] - 0.1) / 0.05 < 0.99))
        if os.environ.get('SWEEP_ONLY'):
            print('light closed' if closed else 'light', 'engine trace', trl[:, :10], 'oracle trace', to[:, :10], sep='\n')
        if not knee:
            # (a re-solved J that overflows float32 turns the whole trajectory into NaN from the next iteration on -- in
            # the oracle and in the engine alike: same iterations, and the finite ones are compared)
            assert np.array_equal(np.isnan(trl[:, 1:10]), np.isnan(to[:, 1:10])), (tag, 'light NaN iterations', closed)
            worst['light_nan_runs'] = worst.get('light_nan_runs', 0) + int(np.isnan(to[:, 1:10]).any())
            dpw = float(np.nan_to_num(np.abs(trl[:, 1:10] - to[:, 1:10])).max())
            worst['light_water'] = max(worst.get('light_water', 0.0), dpw)
            assert dpw < (2e-3 if closed else 2e-4), (tag, 'light water params', closed, dpw)


def main():
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    max_w, max_h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (260, 200)
    rng = np.random.default_rng(seed0)
    worst = {'J': 0.0, 'Jc': 0.0, 'Ju': 0.0, 'p': 0.0, 'pc': 0.0}
    t0 = time.time()
    for s in range(n_scenes):
        W, H = int(rng.integers(33, max_w)), int(rng.integers(33, max_h))
        nn = int(rng.integers(1, int(os.environ.get('SWEEP_MAX_NEIGHBOURS', '13')) + 1))   # > 254: the quantised counting-sort bins
        kw = dict(relief=float(rng.choice([0.0, 0.15, 0.6])), spacing=float(rng.choice([0.05, 0.1, 0.25, 0.5])),
                  invalid_frac=float(rng.choice([0.0, 0.01, 0.3])), rot_sigma=float(rng.choice([0.0, 0.03, 0.15])),
                  pos_sigma=float(rng.choice([0.0, 0.1, 0.4])), far_views=int(rng.integers(0, 3)))
        if os.environ.get('SWEEP_ONLY') and int(os.environ['SWEEP_ONLY']) != s:   # keep the generator's sequence, skip the work
            rng.choice([3, 20, 60])
            continue
        sc = synth.make_scene(W, H, nn, seed=seed0 + s, **kw)
        per_view, samples = helpers.oracle_scene_samples(sc)
        zmax = max([float(np.linalg.norm(x[2], axis=0).max()) for x in samples if len(x[0])] + [0.0])
        views = engine.device_views_from_scene(sc, 'cuda')
        tgt = sc.views[sc.target]
        T = int(rng.choice([3, 20, 60]))
        for fmt in ('f32', 'u16mm'):
            r = engine.Restoration(H, W, len(views), obs_format=fmt)
            r.match(views[sc.target], views)
            assert r.view_counts().cpu().numpy().tolist() == [len(m) for _, _, m in per_view], (s, 'counts')
            kept = [k for _, k, _ in per_view]
            assert r.view_keep().cpu().numpy().astype(bool).tolist() == kept, (s, 'kept')
            if fmt == 'f32':
                for k, (_, _, m) in enumerate(per_view):
                    ref = np.full((H, W), -1, np.int32)
                    ref[m.v1.astype(np.int64), m.u1.astype(np.int64)] = m.v2.astype(np.int32) * W + m.u2.astype(np.int32)
                    assert np.array_equal(r.match_map(k).cpu().numpy(), ref), (s, 'map', k)
            if r.n_obs() == 0:
                continue
            smp = samples if fmt == 'f32' else oracle.quantize_ranges_u16mm(samples)
            for closed in (False, True):
                r.fit_init(views[sc.target])
                tr = r.fit(T, use_closed_form=closed).cpu().numpy()
                J = r.J().cpu().numpy()
                J0 = None if closed else oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
                Jo, po, to = oracle.fit(H, W, smp, J0, num_iter=T, use_closed_form=closed)
                if overflow_run(tr, to, (s, fmt, closed), worst, zmax, closed):
                    continue
                if os.environ.get('SWEEP_ONLY'):
                    d = np.isnan(J) != np.isnan(Jo)
                    dtr = np.abs(tr[:, 1:] - to[:, 1:]).max(axis=1)
                    print(fmt, 'closed' if closed else 'J-param', 'NaN masks differ at', int(d.sum()), 'engine there', J[d][:6], 'oracle there', Jo[d][:6],
                          'params', tr[-1, 1:], to[-1, 1:], 'max |dparam| by iteration', np.array2string(dtr, precision=2), sep='\n')
                    if d.any():
                        idx = np.argwhere(d.any(axis=2))[:3]
                        for (vv, uu) in idx:
                            zs = [np.linalg.norm(smp_[2][:, (smp_[0] == uu) & (smp_[1] == vv)], axis=0) for smp_ in smp]
                            print('pixel', uu, vv, 'ranges of its observations', [z.tolist() for z in zs if z.size])
                assert np.array_equal(np.isnan(J), np.isnan(Jo)), (s, fmt, closed, 'nan mask')
                rms_all = float(np.nan_to_num(helpers.rms_per_channel(J, Jo)).max())
                # Adam on J has the same eps knee per pixel (sucre.py:148 steps J by lr m / (sqrt(v) + eps), and a pixel
                # whose gradient is ~1e-8 takes a step whose length depends on the gradient's last digits): on these
                # small images a single such pixel (|dJ| ~ 1e-3, everything else < 1e-6) is the whole RMS.  The tight bar
                # is therefore taken without the five largest pixel errors, the north-star bar (1e-4) with them.
                err = np.nan_to_num(np.abs(J - Jo)).reshape(-1, 3)
                trimmed = err.copy()
                trimmed[np.argsort(err.max(axis=1))[-5:]] = 0.0
                n_valid = max(1, int((~np.isnan(Jo).any(axis=2)).sum()))
                rms = float(np.sqrt((trimmed.astype(np.float64) ** 2).sum(axis=0) / n_valid).max())
                worst['J_untrimmed'] = max(worst.get('J_untrimmed', 0.0), rms_all if not closed else 0.0)
                # (round 6, scene 157 of seed 61000: ONE pixel of a 165x36 image -- five observations, red gradient at Adam's eps knee --
                # ends 8e-3 from the oracle's with the 5-byte store and is the whole RMS of 1.25e-4; the same with one strip per wave
                # and with four, tools/exp/sweep_scene_61000_157.py.  Such an image is counted, and held to the trimmed bar below.)
                if not closed and rms_all >= 1e-4:
                    assert rms < 1e-5 and n_valid < 20000, (s, fmt, closed, 'untrimmed', rms_all, 'trimmed', rms)
                    worst['one_pixel_over_1e-4'] = worst.get('one_pixel_over_1e-4', 0) + 1
                # (a channel whose closed-form J overflows -- an observed pixel whose every a^2 underflows, ranges of
                # hundreds of metres -- has an infinite cost and NaN parameters from the next step on, in the reference, the
                # oracle and the engine alike: same iterations, same channels; the other channels are compared)
                if os.environ.get('SWEEP_ONLY') and not np.array_equal(np.isnan(tr[:, 1:]), np.isnan(to[:, 1:])):
                    print('engine trace (first 4 rows)', tr[:4], 'oracle trace', to[:4], 'first NaN iteration per column: engine',
                          np.isnan(tr).argmax(axis=0), 'oracle', np.isnan(to).argmax(axis=0), sep='\n')
                dp = float(np.abs(tr[:, 1:] - to[:, 1:]).max())
                key = ('Jc' if closed else 'J') if fmt == 'f32' else 'Ju'
                bar_J, bar_p = (1e-4, 1e-3) if closed else (1e-5, 1e-4)
                # closed-form J = sum(y a) / sum(a^2) is unbounded where a pixel has one or two far observations
                # (a = exp(-beta z) small): the bar is relative to the largest |J| then
                scale = max(1.0, float(np.nanmax(np.abs(Jo)))) if closed and np.isfinite(Jo).any() else 1.0
                # Adam's first step is lr g / (|g| + eps): where it comes out visibly shorter than lr, a gradient sits at
                # the eps knee (|g| ~ 1e-7) and float32 summation noise of 1e-10 in it already moves the parameters by
                # 1e-5 -- seen on 2-view scenes in closed-form mode, whose re-solved J makes the gradients nearly cancel
                knee = bool(np.any(np.abs(to[0, 1:] - 0.1) / 0.05 < 0.99))   # (rare with J as a parameter, but it happens: scene 451 of seed 13000)
                if knee:   # the trajectory is then not comparable digit for digit (nor is the reference's with itself):
                    worst['knee'] = worst.get('knee', 0) + 1   # held to the first iteration's cost, which has no step behind it
                    # (absolute floor: with one observation per pixel the re-solved J fits exactly and the cost is rounding
                    # noise -- 3e-11 in the two-pass form of the oracle, 1e-20 in the engine's one-pass form)
                    assert abs(tr[0, 0] - to[0, 0]) < 1e-5 * to[0, 0] + 1e-9, (s, fmt, closed, 'cost of iteration 0', tr[0, 0], to[0, 0])
                    continue
                worst[key] = max(worst[key], rms / scale)
                worst['pc' if closed else 'p'] = max(worst['pc' if closed else 'p'], dp)
                assert rms < bar_J * scale and dp < bar_p, (s, fmt, closed, rms, dp, scale, W, H, nn, kw, T)
        # the import path (a kept matches file / a hand-built MatchesData: sucre_import_view instead of matching): the kept
        # views' lists go in, the fit must land where the oracle's does
        if os.environ.get('SWEEP_IMPORT', '1') != '0' and sum(len(x[0]) for x in samples) > 0:
            lists = []
            for (u1_, v1_, cP_, I_) in samples:
                c = torch.from_numpy(np.ascontiguousarray(cP_, np.float32))
                z_ = torch.sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2])            # sucre.py:53 in the match kernel's order
                rgb_ = torch.from_numpy(np.rint(np.ascontiguousarray(I_, np.float64).T * 255).astype(np.uint8))
                lists.append((torch.from_numpy(u1_.astype(np.int16)), torch.from_numpy(v1_.astype(np.int16)), z_, rgb_))
            ri = engine.Restoration(H, W, len(lists))
            ri.import_matches(views[sc.target], lists)
            assert ri.n_obs() == sum(len(x[0]) for x in samples), (s, 'import n_obs')
            for closed in (False, True):
                ri.fit_init(views[sc.target])
                tri = ri.fit(T, use_closed_form=closed).cpu().numpy()
                Ji = ri.J().cpu().numpy()
                J0 = None if closed else oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
                Jo, po, to = oracle.fit(H, W, samples, J0, num_iter=T, use_closed_form=closed)
                if overflow_run(tri, to, (s, 'import', closed), worst, zmax, closed):
                    continue
                assert np.array_equal(np.isnan(Ji), np.isnan(Jo)), (s, 'import nan mask', closed)
                assert abs(tri[0, 0] - to[0, 0]) < 1e-5 * to[0, 0] + 1e-9, (s, 'import cost 0', closed, tri[0, 0], to[0, 0])
                if bool(np.any(np.abs(to[0, 1:] - 0.1) / 0.05 < 0.99)):
                    continue
                err = np.nan_to_num(np.abs(Ji - Jo)).reshape(-1, 3)
                err[np.argsort(err.max(axis=1))[-5:]] = 0.0
                n_valid = max(1, int((~np.isnan(Jo).any(axis=2)).sum()))
                scale = max(1.0, float(np.nanmax(np.abs(Jo)))) if closed and np.isfinite(Jo).any() else 1.0
                rmsi = float(np.sqrt((err.astype(np.float64) ** 2).sum(axis=0) / n_valid).max()) / scale
                dpi = float(np.abs(tri[:, 1:] - to[:, 1:]).max())
                worst['Ji'] = max(worst.get('Ji', 0.0), rmsi)
                worst['pi'] = max(worst.get('pi', 0.0), dpi)
                assert rmsi < (1e-4 if closed else 1e-5) and dpi < (1e-3 if closed else 1e-4), (s, 'import', closed, rmsi, dpi, W, H, nn, kw, T)
        # artificial-light model on the same scene (J-parameter and closed-form)
        if os.environ.get('SWEEP_LIGHT', '1') != '0' and sum(len(x[0]) for x in samples) > 0:
            rl = engine.Restoration(H, W, len(views), light=True)
            check_light(rl, views, sc.target, H, W, samples, lambda: oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()),
                        (s, 'light', W, H, nn, kw), worst)
        # float32 colours (--image-scale inputs are not k/255): the colours ride in extension planes (SUCRE_EXT_COLOUR)
        if os.environ.get('SWEEP_FCOLOUR', '1') != '0' and sum(len(x[0]) for x in samples) > 0:
            gen = torch.Generator().manual_seed(seed0 + s)
            frgb, fviews = [], []
            for v in sc.views:
                f = (v.rgb_u8.to(torch.float64) / 255).to(torch.float32)
                f = (f + (torch.rand(f.shape, generator=gen) - 0.5) * 0.003).clamp(0, 1).contiguous()
                frgb.append(f)
                fviews.append(engine.DeviceView(depth=v.depth_f32().cuda().contiguous(), rgb=f.cuda(), K=sc.K, R=v.R, t=v.t, name=v.name))
            rf = engine.Restoration(H, W, len(fviews), float_colour=True)
            rf.match(fviews[sc.target], fviews)
            assert rf.view_counts().cpu().numpy().tolist() == [len(m) for _, _, m in per_view], (s, 'float colour counts')
            fsamples = []
            for ((name, kept, m), f), smp in zip([p for p in sorted(zip(per_view, frgb), key=lambda p: p[0][0]) if p[0][1]], samples):
                fsamples.append((smp[0], smp[1], smp[2], f.numpy()[m.v2.astype(np.int64), m.u2.astype(np.int64)].T.copy()))
            Tf = int(T)
            for closed in (False, True):
                rf.fit_init(fviews[sc.target])
                trf = rf.fit(Tf, use_closed_form=closed).cpu().numpy()
                Jf = rf.J().cpu().numpy()
                J0 = None
                if not closed:
                    J0 = frgb[sc.target].numpy().copy()
                    J0[tgt.depth_f32().numpy() <= 0] = np.nan
                Jo, po, to = oracle.fit(H, W, fsamples, J0, num_iter=Tf, use_closed_form=closed)
                if overflow_run(trf, to, (s, 'float colour', closed), worst, zmax, closed):
                    continue
                assert np.array_equal(np.isnan(Jf), np.isnan(Jo)), (s, 'float colour nan mask', closed)
                knee = bool(np.any(np.abs(to[0, 1:] - 0.1) / 0.05 < 0.99))
                assert abs(trf[0, 0] - to[0, 0]) < 1e-5 * to[0, 0] + 1e-9, (s, 'float colour cost 0', closed, trf[0, 0], to[0, 0])
                if knee:
                    continue
                err = np.nan_to_num(np.abs(Jf - Jo)).reshape(-1, 3)
                err[np.argsort(err.max(axis=1))[-5:]] = 0.0
                n_valid = max(1, int((~np.isnan(Jo).any(axis=2)).sum()))
                scale = max(1.0, float(np.nanmax(np.abs(Jo)))) if closed else 1.0
                rmsf = float(np.sqrt((err.astype(np.float64) ** 2).sum(axis=0) / n_valid).max()) / scale
                dpf = float(np.abs(trf[:, 1:] - to[:, 1:]).max())
                worst['Jf'] = max(worst.get('Jf', 0.0), rmsf)
                worst['pf'] = max(worst.get('pf', 0.0), dpf)
                assert rmsf < (1e-4 if closed else 1e-5) and dpf < (1e-3 if closed else 1e-4), (s, 'float colour', closed, rmsf, dpf, W, H, nn, kw, Tf)
            if os.environ.get('SWEEP_LIGHT', '1') != '0':   # the light model on float32 colours: points AND colours ride along
                def float_J0():
                    J0f = frgb[sc.target].numpy().copy()
                    J0f[tgt.depth_f32().numpy() <= 0] = np.nan
                    return J0f
                rlf = engine.Restoration(H, W, len(fviews), light=True, float_colour=True)
                check_light(rlf, fviews, sc.target, H, W, fsamples, float_J0, (s, 'light + float colour', W, H, nn, kw), worst)
        if (s + 1) % 10 == 0:
            print(f'{s + 1} scenes ok, worst so far {worst}, {time.time() - t0:.0f}s', flush=True)
    print('sweep ok', worst)


if __name__ == '__main__':
    main()
