"""GPU tests of the drop-in surface: the sucre.py CLI on files, sfm/loader mirrors, and the shared-water path."""
import copy
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers
from oracle import oracle
from sucre_amd import synth

pytestmark = pytest.mark.gpu


def _same_points(a, b):
    """Camera points K^-1 (d [u+.5, v+.5, 1]) formed by a torch matmul: which FMA order the (3,3)@(3,n) product takes
    depends on the BLAS path of the machine (host CPU features, GPU library), so they agree to a rounding, not to
    the bit, across machines (the engine's own ranges are formed in a fixed order and are compared bit for bit)."""
    return a.shape == b.shape and np.allclose(a, b, rtol=1e-6, atol=1e-6)


def write_scene(scene, root: Path):
    """Synthetic scene -> image files + COLMAP text model, as a user of the reference would have them."""
    from PIL import Image as PILImage
    from sucre_amd import sfm
    (root / 'images').mkdir(parents=True); (root / 'depth').mkdir()
    for v in scene.views:
        PILImage.fromarray(v.rgb_u8.numpy()).save(root / 'images' / v.name)
        PILImage.fromarray(v.depth_u16.numpy().astype(np.uint16)).save(root / 'depth' / ('depth_' + Path(v.name).stem + '.png'))
    sfm.write_colmap_text(root / 'model', scene.K, scene.width, scene.height, scene.names,
                          [sfm.Pose(v.R, v.t) for v in scene.views])


def scene_as_loaded(scene, model):
    """The scene with the poses the COLMAP round trip produced (quaternion text is not bit-preserving)."""
    sc = copy.copy(scene)
    sc.views = []
    for v in scene.views:
        im = model[v.name]
        sc.views.append(synth.SynthView(name=v.name, R=im.pose.R.contiguous(), t=im.pose.t.contiguous(), depth_u16=v.depth_u16, rgb_u8=v.rgb_u8))
    return sc


@pytest.fixture(scope='module')
def disk_scene(tmp_path_factory):
    from sucre_amd import sfm
    root = tmp_path_factory.mktemp('scene')
    scene = synth.make_scene(96, 64, 4, seed=21, far_views=1)
    write_scene(scene, root)
    model = sfm.COLMAPModel(root / 'model', root / 'images', root / 'depth')
    return root, scene, model, scene_as_loaded(scene, model)


def test_cli_end_to_end(disk_scene, tmp_path, capsys):
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    name = scene.names[scene.target]
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(tmp_path), '--image-name', name, '--num-iter', '30', '--keep-matches', '--save-interval', '20'])
    out = capsys.readouterr().out
    assert 'Solve least squares with Adam optimizer (30 iterations).' in out and 'iter: 0029, cost:' in out
    stem = Path(name).stem
    for f in (f'{stem}_rgb.png', f'{stem}_reconstruction.png', f'{stem}_rgb_0000.png', f'{stem}_rgb_0020.png', f'{stem}.pt'):
        assert (tmp_path / f).exists(), f
    state = torch.load(tmp_path / f'{stem}.pt')
    assert set(state) == {'B', 'beta', 'gamma', 'J'} and state['B'].shape == (3, 1) and state['J'].shape == (64, 96, 3)
    # oracle on the same (as-loaded) inputs
    _, samples = helpers.oracle_scene_samples(loaded)
    tgt = loaded.views[loaded.target]
    J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    Jo, po, to = oracle.fit(64, 96, samples, J0, num_iter=30)
    J = state['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo))
    assert helpers.rms_per_channel(J, Jo).max() < 1e-5
    got = np.concatenate([state['B'].numpy().ravel(), state['beta'].numpy().ravel(), state['gamma'].numpy().ravel()])
    assert np.abs(got - po).max() < 1e-5
    # kept matches: a real HDF5 file in the reference's layout holding the oracle's match lists
    from sucre_amd import h5bridge
    assert h5bridge.available() and (tmp_path / f'{stem}.h5').exists()
    kept = h5bridge.read_groups(tmp_path / f'{stem}.h5')
    per_view, samples = helpers.oracle_scene_samples(loaded)
    assert list(kept) == sorted(n for n, k, _ in per_view if k)
    for (vname, keep, m), view in zip(per_view, loaded.views):
        if not keep:
            continue
        g = kept[vname]
        for key in ('u1', 'v1', 'u2', 'v2', 'd'):
            assert g[key].dtype == getattr(m, key).dtype and np.array_equal(g[key], getattr(m, key)), (vname, key)
        assert g['I'].dtype == np.float32 and np.array_equal(g['I'], oracle.gather_rgb(view.rgb_u8.numpy(), m.u2, m.v2))
    # second run: the kept file is consumed instead of matching (sucre.py:185) and gives the same restoration
    out2 = tmp_path / 'again'
    import shutil
    out2.mkdir(); shutil.copy(tmp_path / f'{stem}.h5', out2 / f'{stem}.h5')
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(out2), '--image-name', name, '--num-iter', '30', '--keep-matches'])
    txt = capsys.readouterr().out
    assert 'Compute' not in txt and 'Total of' in txt
    state2 = torch.load(out2 / f'{stem}.pt')
    assert helpers.rms_per_channel(state2['J'].numpy(), J).max() < 1e-6
    assert torch.allclose(state2['B'], state['B'], atol=1e-6) and (out2 / f'{stem}.h5').exists()


def test_match_two_way_and_matches_data_compat(disk_scene):
    from sucre_amd import loader
    root, scene, model, loaded = disk_scene
    target = model[scene.names[scene.target]]
    other = model[scene.names[0]]
    m = target.match_two_way(other)
    per_view, samples = helpers.oracle_scene_samples(loaded)
    ref = per_view[0][2]
    assert len(m) == len(ref)
    assert np.array_equal(m.u1.cpu().numpy(), ref.u1) and np.array_equal(m.v1.cpu().numpy(), ref.v1)
    assert np.array_equal(m.u2.cpu().numpy(), ref.u2) and np.array_equal(m.v2.cpu().numpy(), ref.v2)
    # the reference-format iterator over the HBM store
    image_list = list(model.images.values())
    mf = loader.MatchesFile(Path('/tmp/unused.h5'), colmap_model=model)
    with pytest.raises(RuntimeError, match='no CPU path'):
        target.match_images(image_list, mf)          # the reference's default device='cpu' is refused, not replaced
    target.match_images(image_list, mf, device='cuda')
    mf.prepare_matches(); mf.check_integrity()
    md = mf.load_matches()
    assert len(md) == sum(len(s[0]) for s in samples) and len(mf) == len(md)
    assert [im.name for im in mf.get_image_list()] == sorted(n for n, k, _ in per_view if k)
    got = list(md.iter(batch_size=1, device='cpu'))
    assert len(got) == len(samples)
    for (u, v, cP, I), (su, sv, scP, sI) in zip(got, samples):
        assert np.array_equal(u.numpy(), su) and np.array_equal(v.numpy(), sv) and np.array_equal(I.numpy(), sI)
        assert _same_points(cP.numpy(), scP)     # the camera points of loader.py:113 themselves


def test_shared_water_group_vs_oracle():
    """Two images share B, beta, gamma (the N>1 exchange, emulated in one process: the all-reduce is a sum): the
    single-launch group path (engine.HipWaterGroup: one launch per iteration over both images, parameter step in the
    next launch's prologue) against the oracle, J-parameter and closed-form."""
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    T = 15
    for closed in (False, True):
        rs, oimgs, scenes = [], [], []
        for seed in (10, 11):
            scene = synth.make_scene(48, 32, 3, seed=seed)
            views = engine.device_views_from_scene(scene, 'cuda')
            r = engine.Restoration(32, 48, len(views))
            r.match(views[scene.target], views)
            r.fit_init(views[scene.target])
            rs.append(r)
            _, samples = helpers.oracle_scene_samples(scene)
            tgt = scene.views[scene.target]
            oimgs.append(oracle.SharedWaterImage(32, 48, samples, None if closed else oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()),
                                                 use_closed_form=closed))
        total = sum(r.n_obs() for r in rs)
        assert total == sum(o.n_obs for o in oimgs)
        trace = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
        group = engine.HipWaterGroup(rs, use_closed_form=closed, trace=trace)
        sdist.fit_shared_water(group, T)
        p0, p1 = rs[0].params().cpu().numpy(), rs[1].params().cpu().numpy()
        assert np.array_equal(p0, p1) and np.array_equal(p0, trace[-1, 1:].cpu().numpy().astype(np.float32))
        assert np.all(np.isfinite(trace.cpu().numpy())) and float(trace[-1, 0]) < float(trace[0, 0])
        from sucre_amd import _lib
        with pytest.raises(_lib.SucreError, match='iterations run in order'):
            group.grad(1)          # a group runs its iterations once (its water state is double-buffered by step parity)
        pstate = np.zeros(27, np.float32); pstate[:9] = 0.1
        for it in range(1, T + 1):
            acc = sum(o.grad(pstate[:9], it, total) for o in oimgs)
            oracle.shared_step(pstate, acc, it, total)
        if closed:
            for o in oimgs:
                o.final_update_J(pstate[:9])
        assert np.abs(p0 - pstate[:9]).max() < (1e-4 if closed else 1e-5)
        for r, o in zip(rs, oimgs):
            assert helpers.rms_per_channel(r.J().cpu().numpy(), o.J).max() < (1e-4 if closed else 1e-5)


@pytest.mark.parametrize('closed', [False, True])
def test_shared_water_group_of_one_equals_the_fused_fit(golden, closed):
    """A group of one image on one rank is the reference algorithm: the single-launch group path must reproduce
    sucre_fit_run bit for bit (same kernels' arithmetic, same reduction tree; only where the parameter step is taken
    differs)."""
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    sc = golden.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    T = 12
    r = engine.Restoration(sc.height, sc.width, len(views))
    r.match(views[sc.target], views)
    r.fit_init(views[sc.target])
    t1 = r.fit(T, use_closed_form=closed).cpu().numpy()
    J1, p1 = r.J().cpu().numpy(), r.params().cpu().numpy()
    r.fit_init(views[sc.target])
    trace = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
    sdist.fit_shared_water(engine.HipWaterGroup([r], use_closed_form=closed, trace=trace), T)
    assert np.array_equal(trace.cpu().numpy(), t1)
    assert np.array_equal(r.J().cpu().numpy(), J1, equal_nan=True) and np.array_equal(r.params().cpu().numpy(), p1)


def test_shared_water_vs_tied_reference_modules(golden):
    """HIP split path (sucre_fit_grad / all-reduce / sucre_fit_step) against the golden produced by two reference
    SUCRe modules with tied B, beta, gamma Parameters."""
    import copy
    from sucre_amd import engine
    t0, t1 = (int(x) for x in golden['shared_targets'])
    rt = golden['shared_trace']
    T = rt.shape[0]
    backends, traces = [], []
    views = engine.device_views_from_scene(golden.scene, 'cuda')
    for tgt in (t0, t1):
        r = engine.Restoration(golden.scene.height, golden.scene.width, len(views))
        r.match(views[tgt], views)
        r.fit_init(views[tgt])
        tr = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
        backends.append(engine.HipWaterBackend(r, trace=tr)); traces.append(tr)
    total = sum(b.n_obs() for b in backends)
    assert total == int(golden['shared_n_total'])
    for b in backends:
        b.set_n_obs_total(total)
    for it in range(1, T + 1):
        sums = [b.grad(it) for b in backends]
        red = sums[0] + sums[1]
        for s in sums:
            s.copy_(red)
        for b in backends:
            b.step(it)
    tr = traces[0].cpu().numpy()
    assert np.array_equal(tr, traces[1].cpu().numpy())
    assert np.abs(tr[:, 1:] - rt[:, 1:]).max() < 1e-5
    assert np.abs(tr[:, 0] / rt[:, 0] - 1).max() < 1e-4
    for b, key in zip(backends, ('shared_J0', 'shared_J1')):
        J = b.r.J().cpu().numpy()
        assert np.array_equal(np.isnan(J), np.isnan(golden[key]))
        assert helpers.rms_per_channel(J, golden[key]).max() < 1e-4


def test_survey_of_images_reuses_one_workspace():
    """Several targets of one survey restored back to back through the same workspace (BASELINE config 3 shape):
    every image must equal its own oracle run -- nothing may leak from the previous image."""
    from sucre_amd import engine
    survey = synth.make_survey(80, 48, 5, 4, seed=9)
    dev_views = engine.device_views_from_scene(survey, 'cuda')
    resto = engine.acquire_restoration(48, 80, 5, 'cuda')
    for idx in (6, 7, 13, 0, 19):
        sel = survey.neighbours(idx, 4)
        scene = survey.scene_for(idx, 4)
        resto.match(dev_views[idx], [dev_views[q] for q in sel])
        resto.fit_init(dev_views[idx])
        trace = resto.fit(10)
        J = resto.J().cpu().numpy()
        per_view, samples = helpers.oracle_scene_samples(scene)
        assert resto.view_counts().cpu().numpy().tolist() == [len(m) for _, _, m in per_view]
        tgt = scene.views[scene.target]
        Jo, po, to = oracle.fit(48, 80, samples, oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()), num_iter=10)
        assert np.array_equal(np.isnan(J), np.isnan(Jo))
        assert helpers.rms_per_channel(J, Jo).max() < 1e-5
        assert np.abs(trace.cpu().numpy()[:, 1:] - to[:, 1:]).max() < 1e-5
    assert engine.acquire_restoration(48, 80, 5, 'cuda') is resto


def test_cli_light_model(disk_scene, tmp_path, capsys):
    """--light-model end to end: vignetting plot, cam2light / sigma in the .pt, J equal to the oracle's light fit."""
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    name = scene.names[scene.target]
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(tmp_path), '--image-name', name, '--num-iter', '12', '--light-model'])
    stem = Path(name).stem
    for f in (f'{stem}_rgb.png', f'{stem}_reconstruction.png', f'{stem}_vignetting.png', f'{stem}.pt'):
        assert (tmp_path / f).exists(), f
    state = torch.load(tmp_path / f'{stem}.pt')
    assert set(state) == {'B', 'beta', 'gamma', 'cam2light', 'sigma', 'J'}
    assert state['cam2light'].shape == (6,) and state['sigma'].shape == (2, 2)
    _, samples = helpers.oracle_scene_samples(loaded)
    tgt = loaded.views[loaded.target]
    Jo, po, to = oracle.fit_light(64, 96, samples, oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()), num_iter=12)
    J = state['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo)) and helpers.rms_per_channel(J, Jo).max() < 2e-5
    got = np.concatenate([state[k].numpy().ravel() for k in ('B', 'beta', 'gamma', 'cam2light', 'sigma')])
    assert np.abs(got[:9] - po[:9]).max() < 2e-5 and np.abs(got[9:] - po[9:]).max() < 1e-3


def test_cli_light_model_closed_form(disk_scene, tmp_path):
    """--light-model --use-closed-form end to end, with intermediate plots (the stand-alone light update_J)."""
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    name = scene.names[scene.target]
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(tmp_path), '--image-name', name, '--num-iter', '12', '--light-model',
                '--use-closed-form', '--save-interval', '5'])
    stem = Path(name).stem
    state = torch.load(tmp_path / f'{stem}.pt')
    assert set(state) == {'B', 'beta', 'gamma', 'cam2light', 'sigma', 'J'}     # sucre.py:213-215 always stores J
    _, samples = helpers.oracle_scene_samples(loaded)
    Jo, po, to = oracle.fit_light(64, 96, samples, None, num_iter=12, use_closed_form=True)
    J = state['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo)) and helpers.rms_per_channel(J, Jo).max() < 1e-4
    got = np.concatenate([state[k].numpy().ravel() for k in ('B', 'beta', 'gamma', 'cam2light', 'sigma')])
    assert np.abs(got[:9] - po[:9]).max() < 5e-5 and np.abs(got[9:] - po[9:]).max() < 1e-3
    assert len(list(tmp_path.glob('**/*.png'))) >= 3


@pytest.mark.parametrize('extra,reused', [(['--light-model'], True), (['--image-scale', '0.5'], True),
                                          (['--light-model', '--use-closed-form'], True),
                                          (['--light-model', '--image-scale', '0.5'], True)],
                         ids=['light-model', 'image-scale', 'light-closed-form', 'light-image-scale'])
def test_cli_kept_matches_are_reused_in_every_mode(disk_scene, tmp_path, capsys, extra, reused):
    """A matches file kept by one run is consumed by the next instead of re-matching (sucre.py:185) -- also with
    --light-model (the camera points are rebuilt from the file's u2, v2, d like loader.py:113) and for resized images
    (the kept colours are then float32, not k/255) -- and gives the same restoration.  (A file the engine cannot import
    would be reported as not reused and matched again: `reused` says which behaviour a mode must show.)"""
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    name = scene.names[scene.target]
    stem = Path(name).stem
    base = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
            '--image-name', name, '--num-iter', '12', '--keep-matches'] + extra
    first, again = tmp_path / 'first', tmp_path / 'again'
    sucre.main(base + ['--output-dir', str(first)])
    assert 'Compute' in capsys.readouterr().out
    kept = [f for f in first.iterdir() if f.suffix in ('.h5', '.npz')]
    assert len(kept) == 1
    again.mkdir()
    import shutil
    shutil.copy(kept[0], again / kept[0].name)
    sucre.main(base + ['--output-dir', str(again)])
    txt = capsys.readouterr().out
    assert ('Compute' not in txt) == reused and 'Total of' in txt and ('is not reused' in txt) == (not reused)
    a, b = torch.load(first / f'{stem}.pt'), torch.load(again / f'{stem}.pt')
    assert set(a) == set(b)
    assert np.array_equal(np.isnan(a['J'].numpy()), np.isnan(b['J'].numpy()))
    # the second run rebuilds cP with torch on the host (loader.py:113); a host BLAS may round the 3x3 product
    # differently from the kernel's FMA chain (1 ulp on a third of the components on the MI355X boxes' CPUs), which the
    # plain model does not feel (J to 1e-6) and the ill-conditioned light trajectory amplifies (DESIGN.md section 4.5:
    # the reference's own batch-order noise there is 1.5e-5 RMS in J, 1.1e-3 in the light parameters)
    light = '--light-model' in extra
    assert helpers.rms_per_channel(b['J'].numpy(), a['J'].numpy()).max() < (3e-5 if light else 1e-6)
    for k in a:
        if k != 'J':
            assert torch.allclose(a[k], b[k], atol=(2e-3 if k in ('cam2light', 'sigma') else 1e-4) if light else 2e-6), k


def test_cli_images_in_flight_equal_one_by_one(disk_scene, tmp_path, monkeypatch):
    """A survey through the CLI with two images in flight (engine.in_flight_slot: own stream + own workspace per
    slot) must write the same bits as strictly sequential restoration -- images are independent problems
    (sucre.py:204, 243) and the engine's reductions have a fixed order."""
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    base = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
            '--image-ids', '1', '6', '--num-iter', '25']
    outs = {}
    for mode in ('2', '1', '3'):
        monkeypatch.setenv('SUCRE_IMAGES_IN_FLIGHT', mode)
        out = tmp_path / f'inflight{mode}'
        sucre.main(base + ['--output-dir', str(out)])
        outs[mode] = {p.name: torch.load(p) for p in sorted(out.glob('*.pt'))}
        assert len(outs[mode]) == 5 and len(list(out.glob('*_rgb.png'))) == 5
    for mode in ('2', '3'):
        for name, state in outs['1'].items():
            for k, v in state.items():
                assert torch.equal(torch.nan_to_num(v, nan=-7.0), torch.nan_to_num(outs[mode][name][k], nan=-7.0)), (mode, name, k)
    # and against the oracle for one of them, so "equal" is not "equally wrong"
    name = scene.names[3]
    sc = copy.copy(loaded); sc.target = 3
    _, samples = helpers.oracle_scene_samples(sc)
    tgt = sc.views[3]
    Jo, po, to = oracle.fit(64, 96, samples, oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()), num_iter=25)
    J = outs['2'][Path(name).with_suffix('.pt').name]['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo)) and helpers.rms_per_channel(J, Jo).max() < 1e-5


@pytest.mark.parametrize('closed', [False, True], ids=['J-parameter', 'closed-form'])
def test_cli_fit_batch_equals_one_by_one(disk_scene, tmp_path, monkeypatch, closed, capsys):
    """A survey of small images through the CLI with several images per fit launch (SUCRE_FIT_BATCH; engine.fit_batch, every
    image of a chunk in its own workspace on the slot's stream) must write the same bits as one launch per image; ``auto``
    batches images of this size."""
    from sucre_amd import sucre
    root, scene, model, loaded = disk_scene
    base = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
            '--image-ids', '1', '6', '--num-iter', '20'] + (['--use-closed-form'] if closed else [])
    outs = {}
    for mode in ('1', '3', 'auto'):
        monkeypatch.setenv('SUCRE_FIT_BATCH', mode)
        out = tmp_path / f'batch{mode}'
        capsys.readouterr()
        sucre.main(base + ['--output-dir', str(out)])
        said = capsys.readouterr().out
        assert ('images per launch' in said) == (mode != '1'), mode   # the batch path is the one that ran
        outs[mode] = {p.name: torch.load(p) for p in sorted(out.glob('*.pt'))}
        assert len(outs[mode]) == 5 and len(list(out.glob('*_rgb.png'))) == 5
    for mode in ('3', 'auto'):
        for name, state in outs['1'].items():
            for k, v in state.items():
                assert torch.equal(torch.nan_to_num(v, nan=-7.0), torch.nan_to_num(outs[mode][name][k], nan=-7.0)), (mode, name, k)


def test_plot_J_on_the_device_equals_the_host_path():
    """SUCRe.plot_J with J on the GPU (order statistics by radix select, one stretch kernel: sucre_select_ranks,
    sucre_plot_stretch) must give the same image as the host path, which tests/test_host_logic.py pins to
    reference-made images -- also when a NaN sits in one channel only, when values repeat, and for negative values."""
    from sucre_amd import sucre
    g = torch.Generator().manual_seed(5)
    for H, W in ((97, 131), (480, 640), (33, 1)):
        J = torch.rand((H, W, 3), generator=g) ** 2 * 1.3 - 0.1
        J[torch.rand((H, W), generator=g) < 0.03] = float('nan')
        J[0, :7] = float('nan')
        J[H // 2, 0, 1] = float('nan')                      # one channel only: the whole pixel is invalid
        if W > 100:
            J[5:9, 10:90] = J[5, 10].clone()                # runs of equal values around the order statistics
            J[20:30, :, 2] = torch.round(J[20:30, :, 2] * 8) / 8
        m = sucre.SUCRe.__new__(sucre.SUCRe)
        torch.nn.Module.__init__(m)
        m.J = J.clone()
        host = np.asarray(m.plot_J())
        m.J = J.cuda()
        dev = np.asarray(m.plot_J())
        assert host.shape == (H, W, 3) and np.array_equal(host, dev)


def test_check_store_flags_every_kind_of_damage(golden):
    """sucre_check_store = MatchesFile.check_integrity (loader.py:89-101) in one launch: sound store -> all zero; a
    NaN, a negative range or a lost observation planted in the store -> the right bit of the right view."""
    from sucre_amd import engine
    sc = golden.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views))
    r.match(views[sc.target], views)
    assert r.check_store().cpu().tolist() == [0] * len(views)
    lists = []
    for k in range(len(views)):
        z, rgb = r.export_view(k)
        v1, u1 = torch.nonzero(z > 0, as_tuple=True)
        lists.append([u1.to(torch.int16), v1.to(torch.int16), z[v1, u1].clone(), rgb[v1, u1].clone()])
    lists[0][2][3] = float('nan')
    lists[1][2][5] = -1.0
    r.import_matches(views[sc.target], [tuple(l) for l in lists])
    got = r.check_store().cpu().tolist()
    assert got[0] == 1 and got[1] == 2 and all(g == 0 for g in got[2:])
    with pytest.raises(AssertionError, match='NaN'):
        from sucre_amd import loader
        mf = loader.MatchesFile.__new__(loader.MatchesFile)
        mf.path, mf.restoration = 'planted', r
        mf.check_integrity()


def test_cli_image_scale_half(disk_scene, tmp_path):
    """--image-scale 0.5 (sfm.py:193-199, loader.py:156-170): intrinsics rescaled, colours area-averaged in float64
    (no longer k/255 -> float32 observations), depth nearest-neighbour.  The oracle is fed the same resized inputs."""
    from sucre_amd import sfm, sucre
    root, scene, _, _ = disk_scene
    name = scene.names[scene.target]
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(tmp_path), '--image-name', name, '--num-iter', '15', '--image-scale', '0.5'])
    state = torch.load(tmp_path / (Path(name).stem + '.pt'))
    assert state['J'].shape == (32, 48, 3)
    for f in ('_rgb.png', '_reconstruction.png'):
        assert (tmp_path / (Path(name).stem + f)).exists()
    model = sfm.COLMAPModel(root / 'model', root / 'images', root / 'depth', image_scale=0.5)
    tgt = model[name]
    H, W = tgt.camera.height, tgt.camera.width
    assert (H, W) == (32, 48)
    cam = lambda im: oracle.make_cam(H, W, **helpers.cam_matrices(im.camera.K, im.pose.R.contiguous(), im.pose.t.contiguous()))
    d1 = tgt.get_depth_map().numpy()
    samples = []
    for im in sorted(model.images.values(), key=lambda i: i.name):
        m = oracle.match_view(d1, cam(tgt), im.get_depth_map().numpy(), cam(im))
        if len(m) / (W * H) > 1e-6:
            cP = oracle.unproject(cam(im), m.u2, m.v2, m.d)
            samples.append((m.u1, m.v1, cP, im.get_rgb().numpy()[m.v2.astype(np.int64), m.u2.astype(np.int64)].T.copy()))
    J0 = tgt.get_rgb().numpy().copy()
    assert np.abs(J0 * 255 - np.rint(J0 * 255)).max() > 0.05          # really off the 1/255 grid
    J0[d1 <= 0] = np.nan
    Jo, po, to = oracle.fit(H, W, samples, J0, num_iter=15)
    J = state['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo)) and helpers.rms_per_channel(J, Jo).max() < 1e-6
    got = np.concatenate([state[k].numpy().ravel() for k in ('B', 'beta', 'gamma')])
    assert np.abs(got - po).max() < 1e-5


def test_cli_light_model_on_resized_images(disk_scene, tmp_path):
    """--light-model --image-scale 0.5, a combination the reference accepts (sfm.py:193-199, sucre.py:54-61): the
    observations carry float32 colours AND camera points.  The oracle's light fit is fed the same resized inputs."""
    from sucre_amd import sfm, sucre
    root, scene, _, _ = disk_scene
    name = scene.names[scene.target]
    T = 10
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(tmp_path), '--image-name', name, '--num-iter', str(T), '--image-scale', '0.5', '--light-model'])
    stem = Path(name).stem
    state = torch.load(tmp_path / (stem + '.pt'))
    assert set(state) == {'B', 'beta', 'gamma', 'cam2light', 'sigma', 'J'} and state['J'].shape == (32, 48, 3)
    for f in ('_rgb.png', '_reconstruction.png', '_vignetting.png'):
        assert (tmp_path / (stem + f)).exists()
    model = sfm.COLMAPModel(root / 'model', root / 'images', root / 'depth', image_scale=0.5)
    tgt = model[name]
    H, W = tgt.camera.height, tgt.camera.width
    cam = lambda im: oracle.make_cam(H, W, **helpers.cam_matrices(im.camera.K, im.pose.R.contiguous(), im.pose.t.contiguous()))
    d1 = tgt.get_depth_map().numpy()
    samples = []
    for im in sorted(model.images.values(), key=lambda i: i.name):
        m = oracle.match_view(d1, cam(tgt), im.get_depth_map().numpy(), cam(im))
        if len(m) / (W * H) > 1e-6:
            cP = oracle.unproject(cam(im), m.u2, m.v2, m.d)
            samples.append((m.u1, m.v1, cP, im.get_rgb().numpy()[m.v2.astype(np.int64), m.u2.astype(np.int64)].T.copy()))
    J0 = tgt.get_rgb().numpy().copy()
    J0[d1 <= 0] = np.nan
    Jo, po, to = oracle.fit_light(H, W, samples, J0, num_iter=T)
    J = state['J'].numpy()
    assert np.array_equal(np.isnan(J), np.isnan(Jo)) and helpers.rms_per_channel(J, Jo).max() < 2e-5
    got = np.concatenate([state[k].numpy().ravel() for k in ('B', 'beta', 'gamma', 'cam2light', 'sigma')])
    assert np.abs(got[:9] - po[:9]).max() < 2e-5 and np.abs(got[9:] - po[9:]).max() < 1e-3


def test_overlap_cull_sizes_the_workspace_and_changes_nothing(tmp_path, monkeypatch):
    """The whole COLMAP model is every target's neighbour list (sucre.py:182-183,238-239).  Image.match_images drops
    the images that cannot overlap the target before anything is decoded or allocated; the restoration must be
    the same bits as without the cull, on a workspace sized by the survivors."""
    from sucre_amd import engine, loader, sfm, sucre
    survey = synth.make_survey(96, 64, 12, 10, seed=5, spacing=0.5)          # 120 images, most of them far away
    synth.write_to_disk(survey, tmp_path / 's')
    results = {}
    for cull in ('0', '1'):
        monkeypatch.setenv('SUCRE_CULL_VIEWS', cull)
        engine.release_pool()
        model = sfm.COLMAPModel(tmp_path / 's' / 'model', tmp_path / 's' / 'images', tmp_path / 's' / 'depth')
        image_list = list(model.images.values())
        target = model.images[5 * 12 + 6]
        mf = loader.MatchesFile(tmp_path / f'm{cull}.h5', colmap_model=model)
        target.match_images(image_list, mf, device='cuda')
        mf.check_integrity()
        md = mf.load_matches()
        model_ = sucre.SUCRe(target).to('cuda')
        sucre.adam(model_, md, num_iter=15, device='cuda', verbose=False)
        decoded = sum(im._device_view is not None for im in image_list)
        results[cull] = (mf.restoration.n_views, [im.name for im in mf.get_image_list()], len(md),
                         model_.J.detach().cpu().numpy(), model_.water_vector().cpu().numpy(), decoded)
    (n0, names0, obs0, J0, p0, dec0), (n1, names1, obs1, J1, p1, dec1) = results['0'], results['1']
    assert n0 == 120 and n1 < 40 and dec0 == 120 and dec1 == n1          # workspace and decoding follow the survivors
    assert names0 == names1 and obs0 == obs1 and len(names1) >= 9
    assert np.array_equal(J0, J1, equal_nan=True) and np.array_equal(p0, p1)


class SynthImage:
    """sfm.Image whose pixels come from a synthetic view instead of files (what tests/golden/ref_harness.py does to
    the reference's class)."""

    def __new__(cls, idx, view, K, W, H):
        from sucre_amd import sfm

        class _Image(sfm.Image):
            def get_rgb(self):
                return view.rgb_f32()

            def get_depth_map(self):
                return view.depth_f32()
        return _Image(idx, Path(view.name), Path('depth_' + view.name), sfm.Pose(view.R, view.t), sfm.Camera(1, W, H, K))


def test_reference_call_sequence_with_hand_built_matches_data(golden, tmp_path):
    """restore_image's own statements (sucre.py:179-215) with the matches appended one view at a time the way the
    reference's match_images does (sfm.py:127-138: match_two_way, min_cover rule, save_matches), then
    prepare / check / load_matches -> a list-backed MatchesData (loader.py:103-118) -> SUCRe -> adam.  Must land on
    the reference's own result for that sequence (golden J_param_5) and on its update_J (golden J_closed_init)."""
    from sucre_amd import loader, sucre
    sc = golden.scene
    images = [SynthImage(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)]
    target = images[sc.target]
    matches_file = loader.MatchesFile(tmp_path / 'm.h5', colmap_model=None)
    u1, v1, wP1 = target.unproject_depth_map(target.get_depth_map().cuda(), to_world=True)
    for other in images:
        other_depth = other.get_depth_map().cuda()
        u2, v2, wP2 = other.unproject_depth_map(other_depth, to_world=True)
        m = target.match_two_way(other, u1=u1, v1=v1, wP1=wP1, u2=u2, v2=v2, wP2=wP2)
        if len(m) / (sc.width * sc.height) > 1e-6:
            matches_file.save_matches(matches=m, d=other_depth[m.v2, m.u2])
    matches_file.prepare_matches()
    matches_file.check_integrity()
    md = matches_file.load_matches()
    assert md.restoration is None and len(md) == int(golden['n_obs']) == len(matches_file)
    # the samples are the reference's: same order (name order), same camera points and colours
    _, samples = helpers.oracle_scene_samples(sc)
    assert len(md.data) == len(samples)
    for s, (su, sv, scP, sI) in zip(md.data, samples):
        assert np.array_equal(s.u.numpy(), su) and np.array_equal(s.v.numpy(), sv)
        assert _same_points(s.cP.numpy(), scP) and np.array_equal(s.I.numpy(), sI)
    model = sucre.SUCRe(image=target).to('cuda')
    sucre.adam(sucre=model, matches_data=md, lr=0.05, num_iter=5, batch_size=5, device='cuda')
    J = model.J.detach().cpu().numpy()
    assert np.array_equal(np.isnan(J), np.isnan(golden['J_param_5']))
    assert helpers.rms_per_channel(J, golden['J_param_5']).max() < 1e-6
    got = model.water_vector().cpu().numpy()
    assert np.abs(got - golden['trace_param'][4, 1:]).max() < 1e-6
    assert md.restoration is not None                       # imported once, reused
    closed = sucre.SUCRe(image=target, use_closed_form=True).to('cuda')
    closed.update_J(md)
    assert helpers.rms_per_channel(closed.J.cpu().numpy(), golden['J_closed_init']).max() < 1e-6
    # and the reference-format iterator gives the true camera points back
    for (u, v, cP, I), (su, sv, scP, sI) in zip(md.iter(batch_size=1, device='cpu'), samples):
        assert _same_points(cP.numpy(), scP) and np.array_equal(I.numpy(), sI)


def test_engine_backed_matches_data_iterates_true_camera_points(golden):
    """MatchesData.iter on the HBM store yields cP = unproject_depth(u2, v2, d) (loader.py:113), not a stand-in, with
    and without the light model's extension planes."""
    from sucre_amd import loader
    sc = golden.scene
    images = [SynthImage(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)]
    _, samples = helpers.oracle_scene_samples(sc)
    for light in (False, True):
        mf = loader.MatchesFile(Path('/tmp/unused.h5'), colmap_model=None)
        images[sc.target].match_images(images, mf, device='cuda', light_model=light)
        got = list(mf.load_matches().iter(batch_size=1, device='cpu'))
        assert len(got) == len(samples)
        for (u, v, cP, I), (su, sv, scP, sI) in zip(got, samples):
            assert np.array_equal(u.numpy(), su) and _same_points(cP.numpy(), scP) and np.array_equal(I.numpy(), sI)


def test_select_ranks_is_an_exact_order_statistic():
    """csrc/plot.hip: radix select on the float bit pattern against a full sort, with NaN pixels, negative values,
    zeros of both signs, duplicates and the extreme ranks."""
    from sucre_amd import engine
    g = torch.Generator().manual_seed(5)
    for H, W in ((7, 5), (64, 48), (333, 517)):
        J = torch.randn((H, W, 3), generator=g) * torch.tensor([1.0, 1e-3, 40.0])
        J[torch.rand((H, W), generator=g) < 0.1] = float('nan')            # invalid pixels: all channels
        J[0, 0, 1] = float('nan')                                          # ... or one channel only (sucre.py:87)
        J[1, 1] = torch.tensor([0.0, -0.0, 0.0]); J[2, 2] = J[2, 1]       # signed zeros, duplicates
        ok = ~torch.isnan(J).any(dim=2)
        n = int(ok.sum())
        ranks = sorted({0, 1, n // 100, n // 2, n - 2, n - 1} & set(range(n)))
        got = engine.select_ranks(J.cuda().contiguous(), ranks).cpu().numpy()
        ref = np.sort(J[ok].numpy(), axis=0)[ranks].T
        assert np.array_equal(got, ref)


def test_plot_J_on_the_device_is_the_reference_picture(golden):
    """SUCRe.plot_J with J resident on the GPU (percentiles by device-side select, nothing but 12 floats comes back)
    against the picture the reference itself made from the same J (golden plot_J_200), not against our host path."""
    from sucre_amd import sucre
    sc = golden.scene
    image = SynthImage(1, sc.views[sc.target], sc.K, sc.width, sc.height)
    model = sucre.SUCRe(image=image).to('cuda')
    with torch.no_grad():
        model.J.copy_(torch.tensor(golden['J_param_200']))
    assert model.J.is_cuda
    assert np.array_equal(np.asarray(model.plot_J()), golden['plot_J_200'])
    assert np.array_equal(np.asarray(model.cpu().plot_J()), golden['plot_J_200'])    # and the host path


def test_depth_range_from_the_stored_integers_is_the_range_of_the_device_depth(disk_scene):
    """Image.depth_range (the overlap cull's input) is taken from the decoded uint16 millimetres on the host; it must be
    exactly the extremes of the float32 depth map the engine holds (loader.py:167-170 conversion is monotone)."""
    root, scene, model, loaded = disk_scene
    for im in list(model.images.values())[:3]:
        im.release_device(); im.__dict__.pop('_depth_range', None)
        view = im.device_view('cuda')
        d = view.depth
        got = im.depth_range('cuda')
        assert got == (float(d[d > 0].min()), float(d.max()))
        im.__dict__.pop('_depth_range', None)      # and the device-side evaluation agrees
        assert im.depth_range('cuda') == got


def _cli(root, out, *extra):
    from sucre_amd import sucre
    sucre.main(['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(out)] + [str(x) for x in extra])


def test_cli_flags_that_shape_the_problem(disk_scene, tmp_path, capsys):
    """The reference's remaining command-line flags end to end (sucre.py:222-261, 264-305): --filter-images-path removes
    neighbour views, --min-cover drops thin ones, --learning-rate and --params-path reach the optimiser, --image-list
    selects the targets, --force-compute-matches ignores a kept file, --batch-size / --num-workers change nothing."""
    root, scene, model, loaded = disk_scene
    name = scene.names[scene.target]
    stem = Path(name).stem
    tgt = loaded.views[loaded.target]
    J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    H, W = loaded.height, loaded.width

    # --filter-images-path: the listed image is no neighbour any more (sucre.py:238-239)
    dropped = scene.names[0] if scene.target != 0 else scene.names[1]
    (tmp_path / 'skip.txt').write_text(dropped + '\n')
    _cli(root, tmp_path / 'f', '--image-name', name, '--num-iter', 6, '--filter-images-path', tmp_path / 'skip.txt')
    txt = capsys.readouterr().out
    per_view, samples = helpers.oracle_scene_samples(loaded)
    keep = [i for i, v in enumerate(loaded.views) if v.name != dropped]
    kept_names = [vname for vname, kept, _ in per_view if kept]            # `samples` holds the kept views, in this order
    sub = [s for vname, s in zip(kept_names, samples) if vname != dropped]
    assert f'Total of {sum(len(s[0]) for s in sub)} observations.' in txt
    Jo, po, to = oracle.fit(H, W, sub, J0, num_iter=6)
    st = torch.load(tmp_path / 'f' / f'{stem}.pt')
    assert helpers.rms_per_channel(st['J'].numpy(), Jo).max() < 1e-5
    assert len(keep) == len(loaded.views) - 1

    # --min-cover 0.8 (strict >, sfm.py:136) and --learning-rate
    _, samples80 = helpers.oracle_scene_samples(loaded, 0.8)
    _cli(root, tmp_path / 'm', '--image-name', name, '--num-iter', 6, '--min-cover', 0.8, '--learning-rate', 0.01)
    txt = capsys.readouterr().out
    assert f'Total of {sum(len(s[0]) for s in samples80)} observations.' in txt
    Jo, po, to = oracle.fit(H, W, samples80, J0, num_iter=6, lr=0.01)
    st = torch.load(tmp_path / 'm' / f'{stem}.pt')
    got = np.concatenate([st[k].numpy().ravel() for k in ('B', 'beta', 'gamma')])
    assert helpers.rms_per_channel(st['J'].numpy(), Jo).max() < 1e-5 and np.abs(got - po).max() < 1e-5

    # --params-path: the water parameters start from a saved state (sucre.py:206-207)
    start = {'B': torch.tensor([[0.2], [0.3], [0.4]]), 'beta': torch.tensor([[0.15], [0.12], [0.11]]),
             'gamma': torch.tensor([[0.05], [0.25], [0.3]])}
    torch.save(start, tmp_path / 'start.pt')
    _cli(root, tmp_path / 'p', '--image-name', name, '--num-iter', 5, '--params-path', tmp_path / 'start.pt',
         '--batch-size', 3, '--num-workers', 2)
    capsys.readouterr()
    p0 = np.concatenate([start[k].numpy().ravel() for k in ('B', 'beta', 'gamma')])
    Jo, po, to = oracle.fit(H, W, samples, J0, params0=p0, num_iter=5)
    st = torch.load(tmp_path / 'p' / f'{stem}.pt')
    got = np.concatenate([st[k].numpy().ravel() for k in ('B', 'beta', 'gamma')])
    assert np.abs(got - po).max() < 1e-5 and helpers.rms_per_channel(st['J'].numpy(), Jo).max() < 1e-5

    # --image-list selects the targets; --keep-matches then --force-compute-matches
    names = [scene.names[scene.target], scene.names[(scene.target + 1) % len(scene.names)]]
    (tmp_path / 'targets.txt').write_text('\n'.join(names) + '\n')
    _cli(root, tmp_path / 'l', '--image-list', tmp_path / 'targets.txt', '--num-iter', 4, '--keep-matches')
    txt = capsys.readouterr().out
    assert txt.count('Compute') == 2
    for n in names:
        assert (tmp_path / 'l' / (Path(n).stem + '.pt')).exists() and (tmp_path / 'l' / (Path(n).stem + '_rgb.png')).exists()
    _cli(root, tmp_path / 'l', '--image-list', tmp_path / 'targets.txt', '--num-iter', 4, '--keep-matches')
    assert 'Compute' not in capsys.readouterr().out                       # kept files are consumed ...
    _cli(root, tmp_path / 'l', '--image-list', tmp_path / 'targets.txt', '--num-iter', 4, '--force-compute-matches')
    assert capsys.readouterr().out.count('Compute') == 2                  # ... unless matching is forced


def _golden_lists(golden, k):
    """Match lists of view k from the reference-made golden, as (H1*W1,) linear target pixel -> linear other pixel."""
    u1, v1, u2, v2 = golden.match_lists(k)
    W = golden.scene.width
    return v1.astype(np.int64) * W + u1, v2.astype(np.int64) * W + u2


def test_match_one_way_and_two_way_take_the_reference_arguments(golden):
    """sfm.Image.match_one_way (sfm.py:115-119) and match_two_way (sfm.py:121-125) with the reference's own arguments,
    positionally: the full valid sets give the golden match lists (made by the reference); a caller that passes a
    SUBSET of the pixels gets the matches of that subset -- the reference's composition on the caller's points, with the
    projection on the GPU (sucre_project_points).  Host tensors in, host tensors out."""
    sc = golden.scene
    images = [SynthImage(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)]
    target = images[sc.target]
    W = sc.width
    u1, v1, wP1 = target.unproject_depth_map(target.get_depth_map(), to_world=True)      # host tensors, like device='cpu'
    for k, other in enumerate(images):
        u2, v2, wP2 = other.unproject_depth_map(other.get_depth_map(), to_world=True)
        p1, p2 = _golden_lists(golden, k)
        # one-way: truncation towards zero, bound test on the other sensor, no in-front-of-camera test
        m1 = target.match_one_way(other, u1, v1, wP1)
        px = other.project_to_view(wP1)
        ref = (px[0] > -1) & (px[0] < other.camera.width) & (px[1] > -1) & (px[1] < other.camera.height)
        assert m1.u1.device.type == 'cpu' and len(m1) == int(ref.sum())
        assert torch.equal(m1.u1, u1[ref]) and torch.equal(m1.v1, v1[ref])
        assert torch.equal(m1.u2, px[0][ref].long()) and torch.equal(m1.v2, px[1][ref].long())
        # two-way, positional, full sets: the golden lists
        m = target.match_two_way(other, u1, v1, wP1, u2, v2, wP2)
        assert np.array_equal((m.v1 * W + m.u1).numpy(), p1) and np.array_equal((m.v2 * W + m.u2).numpy(), p2)
        # the same on device tensors
        md = target.match_two_way(other, u1.cuda(), v1.cuda(), wP1.cuda(), u2.cuda(), v2.cuda(), wP2.cuda())
        assert md.u1.is_cuda and np.array_equal((md.v1 * W + md.u1).cpu().numpy(), p1)
        # and the fused no-argument form agrees
        mf = target.match_two_way(other)
        assert np.array_equal((mf.v1 * W + mf.u1).cpu().numpy(), p1) and np.array_equal((mf.v2 * W + mf.u2).cpu().numpy(), p2)
        # a subset of the target's pixels (every third) -> the golden lists restricted to it
        sub = torch.arange(0, u1.numel(), 3)
        ms = target.match_two_way(other, u1[sub], v1[sub], wP1[:, sub], u2, v2, wP2)
        keep = np.isin(p1, (v1[sub] * W + u1[sub]).numpy())
        assert np.array_equal((ms.v1 * W + ms.u1).numpy(), p1[keep]) and np.array_equal((ms.v2 * W + ms.u2).numpy(), p2[keep])
        # a subset of the OTHER image's pixels: only matches whose p2 is in it survive (Matches.__and__, sfm.py:171-175)
        sub2 = torch.arange(0, u2.numel(), 2)
        ms2 = target.match_two_way(other, u1, v1, wP1, u2[sub2], v2[sub2], wP2[:, sub2])
        keep2 = np.isin(p2, (v2[sub2] * W + u2[sub2]).numpy())
        assert np.array_equal((ms2.v1 * W + ms2.u1).numpy(), p1[keep2])
    with pytest.raises(TypeError):
        target.match_two_way(images[0], u1, v1, wP1)


def test_list_backed_matches_data_objects_are_independent(golden):
    """Two hand-built MatchesData of the same image size (loader.py:36-53): importing the second into the engine must
    not change what a fit of the first one sees (advisor r02: a pooled workspace handed A the observations of B)."""
    from sucre_amd import loader, sucre
    sc = golden.scene
    images = [SynthImage(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)]
    _, samples = helpers.oracle_scene_samples(sc)
    target = images[sc.target]

    def build(smps):
        md = loader.MatchesData()
        for u, v, cP, I in smps:
            md.append(u=torch.tensor(u), v=torch.tensor(v), cP=torch.tensor(cP), I=torch.tensor(I))
        return md
    md_a, md_b = build(samples), build(samples[:2])            # B: fewer views -> different observations
    a = sucre.SUCRe(image=target, use_closed_form=True).to('cuda')
    a.update_J(md_a)
    J_a = a.J.cpu().numpy().copy()
    b = sucre.SUCRe(image=target, use_closed_form=True).to('cuda')
    b.update_J(md_b)                                            # imports B
    assert md_a.restoration is not md_b.restoration
    a2 = sucre.SUCRe(image=target, use_closed_form=True).to('cuda')
    a2.update_J(md_a)                                           # A again, after B was imported
    assert np.array_equal(a2.J.cpu().numpy(), J_a, equal_nan=True)
    assert helpers.rms_per_channel(J_a, golden['J_closed_init']).max() < 1e-6
    assert not np.array_equal(b.J.cpu().numpy(), J_a, equal_nan=True)


def test_matches_file_written_like_the_reference_is_consumed(golden):
    """(f)1: tests/golden/ref_layout_<fixture>.h5 -- the reference's own matches, written by the h5py calls of
    loader.py:68-87 in their order -- goes through MatchesFile.on_disk / load_file / check_integrity / load_matches and
    the fit lands on the reference's own J for that scene (golden J_param_5, J_closed_init)."""
    from sucre_amd import h5bridge, loader, sucre
    path = helpers.GOLDEN_DIR / f'ref_layout_{golden.name}.h5'
    if not path.exists():
        pytest.skip('fixture made for the plane scene only')
    if not h5bridge.available():
        pytest.skip('no h5py interpreter on this machine (SUCRE_H5PY_PYTHON)')
    sc = golden.scene
    model = {v.name: helpers.synth_image(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)}
    target = model[sc.views[sc.target].name]
    mf = loader.MatchesFile(path, colmap_model=model)
    assert mf.on_disk()
    mf.load_file(target, device='cuda')
    mf.check_integrity()
    assert len(mf) == int(golden['n_obs'])
    assert [im.name for im in mf.get_image_list()] == sorted(str(n) for n, k in zip(golden['names'], golden['kept']) if k)
    md = mf.load_matches()
    m = sucre.SUCRe(image=target).to('cuda')
    sucre.adam(sucre=m, matches_data=md, lr=0.05, num_iter=5, batch_size=5, device='cuda')
    J = m.J.detach().cpu().numpy()
    assert np.array_equal(np.isnan(J), np.isnan(golden['J_param_5']))
    assert helpers.rms_per_channel(J, golden['J_param_5']).max() < 1e-6
    closed = sucre.SUCRe(image=target, use_closed_form=True).to('cuda')
    closed.update_J(md)
    assert helpers.rms_per_channel(closed.J.cpu().numpy(), golden['J_closed_init']).max() < 1e-6


def test_in_flight_slot_is_per_thread_and_a_returned_workspace_knows_its_streams(golden):
    """ ``engine.in_flight_slot`` sets the slot of the CALLING thread only -- the CLI's decode / plan / writer
    pools must not inherit it -- and a leased workspace handed back remembers the streams it was launched on, a user stream
    included, so that its next owner waits for exactly that work (return_restoration may run from a finalizer on any thread)."""
    import threading
    from sucre_amd import engine
    engine.release_pool()   # no idle workspace of an earlier test on the free list
    seen = {}
    gate, done = threading.Event(), threading.Event()

    def other():
        gate.wait(10)
        seen['other'] = engine.current_slot()
        done.set()
    th = threading.Thread(target=other)
    th.start()
    with engine.in_flight_slot(1):
        assert engine.current_slot() == 1
        gate.set()
        done.wait(10)
    th.join()
    assert seen['other'] == 0 and engine.current_slot() == 0
    sc = golden.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    user = torch.cuda.Stream()
    r = engine.lease_restoration(sc.height, sc.width, len(views))
    with torch.cuda.stream(user):
        r.match(views[sc.target], views)
        r.fit_init(views[sc.target])
        r.fit(3)
    assert user.cuda_stream in r._streams_used
    t = threading.Thread(target=engine.return_restoration, args=(r,))   # as a finalizer would: another thread, another current stream
    t.start(); t.join()
    evs = r.__dict__.get('_idle_after')
    assert evs and len(evs) == 1 and not r._streams_used
    r2 = engine.lease_restoration(sc.height, sc.width, len(views))
    assert r2 is r and '_idle_after' not in r2.__dict__
    r2.match(views[sc.target], views)     # on the default stream: ordered behind the user stream's fit by the event
    r2.fit_init(views[sc.target])
    tr = r2.fit(3).cpu().numpy()
    torch.cuda.synchronize()
    ref = engine.Restoration(sc.height, sc.width, len(views))
    ref.match(views[sc.target], views)
    ref.fit_init(views[sc.target])
    assert np.array_equal(tr, ref.fit(3).cpu().numpy())
    engine.return_restoration(r2)
    engine.release_pool()
