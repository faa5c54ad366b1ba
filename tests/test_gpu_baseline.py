"""The HIP engine against the REFERENCE ITSELF at BASELINE.json's sizes -- directly, not through the oracle.

tests/golden/baseline_c1_640x480_n4.npz and baseline_c2_1920x1080_n64.npz hold what the reference computed
(tests/golden/gen_golden_baseline.py: its own match_two_way over every view, its own sucre.adam) on scenes the seeded
generator reproduces; the regenerated inputs are digest-checked before anything is compared (helpers.Baseline).

  config 1 IN FULL: 640x480, 5 views, the reference's 200 iterations, J as a parameter and closed form: full J.
  config 2, SHORT : 1920x1080 x 65 views (the bench's own image), the reference's matching of all 65 views and its
                    first 10 (J parameter) / 5 (closed form) iterations at 79 M observations: J[::4, ::4], the NaN count
                    and the per-channel sums of J and J^2 over the WHOLE image, the cost / B / beta / gamma trajectory.

Bars: match sets bit for bit (per-view counts + SHA-256 of the dense match map); RMS(J) <= 1e-6 in J-parameter mode and
<= 2e-5 in closed-form mode (north star: 1e-4; the reference's own batch-order noise is 5e-8 resp. 5e-6).
"""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def engine_run(b, T_param, T_closed, with_maps=True):
    from sucre_amd import engine
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views))
    r.match(views[sc.target], views)
    counts = r.view_counts().cpu().numpy().tolist()
    keep = r.view_keep().cpu().numpy().astype(bool).tolist()
    assert keep == b['kept'].tolist()
    maps = [r.match_map(k).cpu().numpy() for k in range(len(views))] if with_maps else None
    helpers.check_baseline_matches(b, counts, maps, b.name + ' engine')
    assert r.n_obs() == int(b['n_obs'])
    out = {}
    r.fit_init(views[sc.target])
    t1 = r.fit(1)
    out['J1'] = r.J().cpu().numpy()
    t2 = r.fit(T_param - 1)
    torch.cuda.synchronize()
    out['J'] = r.J().cpu().numpy()
    out['trace'] = np.concatenate([t1.cpu().numpy(), t2.cpu().numpy()])
    r.fit_init(views[sc.target])
    tc = r.fit(T_closed, use_closed_form=True)
    torch.cuda.synchronize()
    out['Jc'] = r.J().cpu().numpy()
    out['trace_c'] = tc.cpu().numpy()
    out['n_obs'] = r.n_obs()
    del r, views
    torch.cuda.empty_cache()
    return out


@pytest.mark.timeout(900)
def test_config1_in_full_engine_vs_reference():
    b = helpers.load_baseline(helpers.BASELINE_C1)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    assert (T_param, T_closed) == (200, 200) and int(b['stride']) == 1
    e = engine_run(b, T_param, T_closed)
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, config 1, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, 'ENGINE, config 1 in full, 200 iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, 'ENGINE, config 1 in full, 200 iterations')


@pytest.mark.timeout(900)
def test_config1_in_full_through_the_batch_launch_vs_reference():
    """BASELINE config 1 against the reference itself once more, through ``sucre_fit_run_batch`` (what ``bench.py --config 1``
    times): three workspaces holding the config-1 image, all 200 iterations of both J modes in one launch per iteration --
    every one of the three must be the reference's J and trajectory (and they are bit-identical to each other)."""
    from sucre_amd import engine
    b = helpers.load_baseline(helpers.BASELINE_C1)
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    rs = [engine.Restoration(sc.height, sc.width, len(views)) for _ in range(3)]
    for r in rs:
        r.match(views[sc.target], views)
    assert rs[0].n_obs() == int(b['n_obs'])
    for key, closed, T, bars in (('param', False, int(b['T_param']), (1e-6, 2e-6, 2e-5)), ('closed', True, int(b['T_closed']), (2e-5, 2e-5, 2e-5))):
        for r in rs:
            r.fit_init(views[sc.target])
        traces = engine.fit_batch(rs, T, use_closed_form=closed)
        torch.cuda.synchronize()
        J0 = rs[0].J().cpu().numpy()
        for i, r in enumerate(rs):
            J = r.J().cpu().numpy()
            assert np.array_equal(np.nan_to_num(J), np.nan_to_num(J0)) and np.array_equal(traces[i].cpu().numpy(), traces[0].cpu().numpy())
        helpers.check_baseline_fit(b, key, J0, traces[0].cpu().numpy(), *bars, f'ENGINE (batch launch), config 1 in full, {T} iterations')


@pytest.mark.timeout(900)
def test_odd_image_size_engine_vs_reference():
    """333x207 (ragged last tile column and row), 8 neighbours + self, against the reference itself: match maps bit for bit, the
    whole J after its 60 J-parameter and 30 closed-form iterations."""
    b = helpers.load_baseline(helpers.BASELINE_ODD)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    e = engine_run(b, T_param, T_closed)
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, 333x207, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, f'ENGINE, 333x207, {T_param} iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, f'ENGINE, 333x207, {T_closed} iterations')


@pytest.mark.timeout(900)
def test_deep_scene_engine_vs_reference():
    """A scene whose ranges span a factor of eleven (0.72 .. 8.03 m; tests/golden/baseline_deep_640x480_n8.npz) against the reference
    itself, in full: match maps of all nine views bit for bit, 200 J-parameter and 200 closed-form iterations -- on the store the
    device chooses for such an image: the float32 words (its ranges do not fit the 24-bit codes of the synthetic surveys; the
    26-bit codes 'f32z26' give the same bits: test_gpu_parity.py)."""
    from sucre_amd import _lib, engine
    b = helpers.load_baseline(helpers.BASELINE_DEEP)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    assert (T_param, T_closed) == (200, 200)
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views))
    r.match(views[sc.target], views)
    word = r.store_format().cpu().numpy().astype(np.uint32)
    assert int(word[0]) == _lib.STORE_F32 and int(word[3]) - int(word[2]) > 0xfffffd, word
    del r
    e = engine_run(b, T_param, T_closed)
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, deep scene, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, f'ENGINE, deep scene, {T_param} iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, f'ENGINE, deep scene, {T_closed} iterations')


@pytest.mark.timeout(900)
def test_config5_view_count_engine_vs_reference():
    """BASELINE config 5's view count (256 neighbours + self: five mask words per pixel, strips of up to 257 levels = 65 chunks)
    on a 480x360 image against the reference itself: all 257 match maps bit for bit, 8 J-parameter and 4 closed-form iterations."""
    b = helpers.load_baseline(helpers.BASELINE_C5VIEWS)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    e = engine_run(b, T_param, T_closed)
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, 257 views, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, f'ENGINE, 257 views, {T_param} iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, f'ENGINE, 257 views, {T_closed} iterations')


@pytest.mark.timeout(900)
def test_config1_extensions_engine_vs_reference():
    """The extensions at config-1 size against the reference itself (tests/golden/baseline_c1_extensions.npz): its own
    --light-model run of 100 iterations (19 parameters, autograd) and its two-module shared-water composition (40
    iterations, tied B, beta, gamma) -- the latter through the single-launch group kernel AND through the split
    grad / all-reduce / step path that N ranks run."""
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    b = helpers.load_baseline('baseline_c1_extensions')
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    # light model
    r = engine.Restoration(sc.height, sc.width, len(views), light=True)
    r.match(views[sc.target], views)
    helpers.check_baseline_matches(b, r.view_counts().cpu().numpy().tolist(), [r.match_map(k).cpu().numpy() for k in range(len(views))], 'extensions engine')
    r.fit_init(views[sc.target])
    T = b['trace_light'].shape[0]
    trace = r.fit(T).cpu().numpy()
    assert abs(trace[0, 0] / b['trace_light'][0, 0] - 1) < 1e-6
    helpers.check_baseline_fit(b, 'light', r.J().cpu().numpy(), trace, 1e-4, 2e-4, 5e-3, f'ENGINE, config 1 light model, {T} iterations',
                               trace_key='trace_light', light_bar=3e-3)
    del r
    # shared water: two images of the scene, tied parameters
    targets = [int(t) for t in b['shared_targets']]
    rt = b['shared_trace']
    for path in ('group', 'split'):
        rs = []
        for t in targets:
            x = engine.Restoration(sc.height, sc.width, len(views))
            x.match(views[t], views)
            x.fit_init(views[t])
            rs.append(x)
        assert sum(x.n_obs() for x in rs) == int(b['shared_n_total'])
        trace = torch.zeros((rt.shape[0], 10), dtype=torch.float64, device='cuda')
        if path == 'group':
            sdist.fit_shared_water(engine.HipWaterGroup(rs, trace=trace), rt.shape[0])
        else:   # what two ranks do, in one process: every image its own backend, the sums added by hand
            bes = [engine.HipWaterBackend(x, trace=trace if i == 0 else None) for i, x in enumerate(rs)]
            total = sum(be.n_obs() for be in bes)
            for be in bes:
                be.set_n_obs_total(total)
            for it in range(1, rt.shape[0] + 1):
                sums = [be.grad(it).clone() for be in bes]
                tot = sums[0] + sums[1]
                for be in bes:
                    be._sums.copy_(tot)
                    be.step(it)
        torch.cuda.synchronize()
        tr = trace.cpu().numpy()
        dpar, dcost = np.abs(tr[:, 1:] - rt[:, 1:]).max(), np.abs(tr[:, 0] / rt[:, 0] - 1).max()
        print(f'ENGINE, config 1 shared water ({path}): max|dparams|={dpar:.2e} max rel dcost={dcost:.2e}')
        helpers.record_parity(f'ENGINE, config 1 shared water ({path})', 'shared', b, dpar=dpar, dcost=dcost, T=rt.shape[0])
        assert dpar < 2e-6 and dcost < 2e-5
        for i, x in enumerate(rs):
            helpers.check_baseline_fit(b, f'shared{i}', x.J().cpu().numpy(), None, 1e-6, 0, 0, f'ENGINE, config 1 shared water ({path}), image {i}')
        del rs


@pytest.mark.timeout(1500)
def test_config2_short_engine_vs_reference():
    b = helpers.load_baseline(helpers.BASELINE_C2)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    e = engine_run(b, T_param, T_closed)
    assert e['n_obs'] > 70_000_000
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, config 2, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, f'ENGINE, config 2, {T_param} iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, f'ENGINE, config 2, {T_closed} iterations')


@pytest.mark.timeout(1500)
def test_config2_in_full_engine_vs_reference():
    """BASELINE config 2 start to end against the reference itself: the bench's own image (1920x1080, 64 neighbours + self,
    79 M observations), the reference's WHOLE J-parameter run -- all 200 Adam iterations, an hour of its CPU path -- and 60
    closed-form iterations (tests/golden/baseline_c2full_1920x1080_n64.npz): J[::4, ::4], the NaN count and the sums of J and
    J^2 over the whole image, and every row of the cost / B / beta / gamma trajectory."""
    b = helpers.load_baseline(helpers.BASELINE_C2FULL)
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    assert T_param == 200 and T_closed >= 50
    e = engine_run(b, T_param, T_closed, with_maps=False)   # (the match maps are pinned by the short fixture of the same scene)
    helpers.check_baseline_fit(b, 'param_1', e['J1'], None, 1e-7, 0, 0, 'ENGINE, config 2 in full, 1 iteration')
    helpers.check_baseline_fit(b, 'param', e['J'], e['trace'], 1e-6, 2e-6, 2e-5, 'ENGINE, config 2 IN FULL, 200 iterations')
    helpers.check_baseline_fit(b, 'closed', e['Jc'], e['trace_c'], 2e-5, 2e-5, 2e-5, f'ENGINE, config 2, {T_closed} closed-form iterations')


@pytest.mark.timeout(1500)
def test_config2_light_model_engine_vs_reference():
    """--light-model at BASELINE config-2 size against the reference itself (tests/golden/baseline_c2_light.npz: the bench's own
    image, 79 M observations; the reference's autograd through se3.exp and the light cone for its first iterations, J as a
    parameter and --use-closed-form): cost of iteration 0, the water trajectory, cam2light / sigma, J[::4, ::4] and the
    whole-image sums.  The light parameters' gradients are sums of 79 M terms of both signs that the reference forms in
    float32 batch by batch: its own batch-order noise on them is 1e-3, hence light_bar."""
    from sucre_amd import engine
    b = helpers.load_baseline('baseline_c2_light')
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views), light=True)
    r.match(views[sc.target], views)
    helpers.check_baseline_matches(b, r.view_counts().cpu().numpy().tolist(), None, 'config 2 light engine')
    for key, closed in (('light', False), ('light_closed', True)):
        rt = b[f'trace_{key}']
        r.fit_init(views[sc.target])
        trace = r.fit(rt.shape[0], use_closed_form=closed).cpu().numpy()
        torch.cuda.synchronize()
        assert abs(trace[0, 0] / rt[0, 0] - 1) < 2e-6, (key, trace[0, 0], rt[0, 0])
        # (the cost follows the light parameters: the ORACLE's is 7.4e-5 from the reference's after six iterations, 4.2e-6 in closed form)
        helpers.check_baseline_fit(b, key, r.J().cpu().numpy(), trace, 1e-5 if closed else 2e-6, 2e-5, 2e-5 if closed else 2e-4,
                                   f'ENGINE, config 2 light model, {rt.shape[0]} iterations', trace_key=f'trace_{key}', light_bar=3e-3)


@pytest.mark.timeout(1500)
def test_config2_shared_water_engine_vs_reference():
    """Shared water parameters at BASELINE config-2 size -- a rank's share of config 4 in small: two 1080p images of the 65-view
    scene (158 M observations) stepping B, beta, gamma together -- against two reference modules with tied parameters
    (tests/golden/baseline_c2_shared.npz), through the single-launch group kernel and through the split grad / sum / step path
    that N ranks run."""
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    b = helpers.load_baseline('baseline_c2_shared')
    sc = b.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    targets = [int(t) for t in b['shared_targets']]
    rt = b['shared_trace']
    for path in ('group', 'split'):
        rs = []
        for t in targets:
            x = engine.Restoration(sc.height, sc.width, len(views))
            x.match(views[t], views)
            x.fit_init(views[t])
            rs.append(x)
        assert sum(x.n_obs() for x in rs) == int(b['shared_n_total'])
        trace = torch.zeros((rt.shape[0], 10), dtype=torch.float64, device='cuda')
        if path == 'group':
            sdist.fit_shared_water(engine.HipWaterGroup(rs, trace=trace), rt.shape[0])
        else:   # what two ranks do, in one process: every image its own backend, the sums added by hand
            bes = [engine.HipWaterBackend(x, trace=trace if i == 0 else None) for i, x in enumerate(rs)]
            total = sum(be.n_obs() for be in bes)
            for be in bes:
                be.set_n_obs_total(total)
            for it in range(1, rt.shape[0] + 1):
                sums = [be.grad(it).clone() for be in bes]
                tot = sums[0] + sums[1]
                for be in bes:
                    be._sums.copy_(tot)
                    be.step(it)
        torch.cuda.synchronize()
        tr = trace.cpu().numpy()
        dpar, dcost = np.abs(tr[:, 1:] - rt[:, 1:]).max(), np.abs(tr[:, 0] / rt[:, 0] - 1).max()
        print(f'ENGINE, config 2 shared water ({path}), {rt.shape[0]} iterations: max|dparams|={dpar:.2e} max rel dcost={dcost:.2e}')
        helpers.record_parity(f'ENGINE, config 2 shared water ({path})', 'shared', b, dpar=dpar, dcost=dcost, T=rt.shape[0])
        assert dpar < 2e-6 and dcost < 2e-5
        for i, x in enumerate(rs):
            helpers.check_baseline_fit(b, f'shared{i}', x.J().cpu().numpy(), None, 1e-6, 0, 0, f'ENGINE, config 2 shared water ({path}), image {i}')
        del rs
        torch.cuda.empty_cache()


# ---- round 6: BASELINE config 5's LOSSY store (uint16 millimetres, 5 B/observation) against the REFERENCE ITSELF -------------
# Until round 5 every `u16mm` check at size compared the engine with the oracle fed the same quantised ranges: that shows the
# kernel implements the format, not that the format meets the north star's 1e-4 against the reference, which never rounds a
# range (VERDICT round 5, weak point 1).  Here the engine's u16mm store runs on the reference-made goldens -- unquantised
# reference outputs -- at config 5's view count, at config 2's size in full (200 + 60 iterations) and on config 1 in full.
# The J bar is the north star's 1e-4; parameter / cost bars are a few times what tools/exp/u16mm_vs_reference.py measured with
# the oracle on the CPU (the engine's own figures are in the terminal summary, mode 'u16mm' / 'u16mm closed').
U16MM_CASES = {
    # fixture: (RMS(J) bar, |dparams| bar, rel dcost bar) J-parameter mode; the same for closed form
    helpers.BASELINE_C5VIEWS: ((1e-4, 1e-6, 5e-6), (1e-4, 2e-5, 1e-4)),
    helpers.BASELINE_C1: ((1e-4, 2e-6, 5e-5), (1e-4, 3e-4, 2e-4)),
    helpers.BASELINE_C2FULL: ((1e-4, 2e-6, 5e-5), (1e-4, 3e-4, 2e-4)),
}


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('name', list(U16MM_CASES))
def test_u16mm_store_engine_vs_unquantised_reference(name):
    from sucre_amd import _lib, engine
    b = helpers.load_baseline(name)
    sc = b.scene
    T_param, T_closed = int(b['T_param']), int(b['T_closed'])
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views), obs_format='u16mm')
    r.match(views[sc.target], views)
    # the store is lossy, the matching is not: same match sets, same counts
    helpers.check_baseline_matches(b, r.view_counts().cpu().numpy().tolist(), None, f'{name} engine u16mm')
    assert r.n_obs() == int(b['n_obs']) and int(r.store_format()[0].item()) == _lib.STORE_U16MM
    bars_p, bars_c = U16MM_CASES[name]
    r.fit_init(views[sc.target])
    trace = r.fit(T_param).cpu().numpy()
    torch.cuda.synchronize()
    helpers.check_baseline_fit(b, 'param', r.J().cpu().numpy(), trace, *bars_p, f'ENGINE u16mm store, {name}, {T_param} iterations', mode='u16mm')
    r.fit_init(views[sc.target])
    trace_c = r.fit(T_closed, use_closed_form=True).cpu().numpy()
    torch.cuda.synchronize()
    helpers.check_baseline_fit(b, 'closed', r.J().cpu().numpy(), trace_c, *bars_c, f'ENGINE u16mm store, {name}, {T_closed} closed-form iterations',
                               mode='u16mm closed')
    del r, views
    torch.cuda.empty_cache()
