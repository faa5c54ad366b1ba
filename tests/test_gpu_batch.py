"""Independent images in one launch per iteration (``sucre_fit_run_batch`` / ``engine.fit_batch``): the reference's loop
over the images of a scene (sucre.py:243-261 -- one SUCRe module, one sucre.adam per image, nothing shared) with all the
images' iterations advancing together.  The bar is the strongest one available: every image's J, parameters and whole
trace are BIT FOR BIT what ``Restoration.fit`` gives that image alone (which is what the reference-pinned tests pin)."""
import numpy as np
import pytest
import torch

import helpers
from oracle import oracle

pytestmark = pytest.mark.gpu


def _targets(survey, engine, idxs, k, obs_format='f32', views_of=None):
    views = engine.device_views_from_scene(survey, 'cuda')
    out = []
    for idx in idxs:
        sel = survey.neighbours(idx, k if views_of is None else views_of[idx])
        vs = [views[q] for q in sel]
        r = engine.Restoration(survey.height, survey.width, len(vs), obs_format=obs_format)
        r.match(views[idx], vs)
        out.append((r, views[idx]))
    return out


def _alone(r, tgt, T, closed, split=None):
    r.fit_init(tgt)
    if split:
        tr = torch.cat([r.fit(split, use_closed_form=closed), r.fit(T - split, use_closed_form=closed)])
    else:
        tr = r.fit(T, use_closed_form=closed)
    torch.cuda.synchronize()
    return r.J().cpu().numpy(), r.params().cpu().numpy().copy(), tr.cpu().numpy()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('closed', [False, True], ids=['J-parameter', 'closed-form'])
@pytest.mark.parametrize('obs_format', ['f32', 'u16mm'])
def test_batch_of_images_is_bitwise_the_images_one_by_one(closed, obs_format):
    """Six targets of one survey with DIFFERENT view counts (5 .. 10 views: six workspace layouts), 160x120 (ragged last tile
    row), 12 iterations, split into two batch calls (resumable, like fit)."""
    from sucre_amd import engine, synth
    survey = synth.make_survey(160, 120, 5, 5, seed=5)
    idxs = [6, 7, 8, 11, 12, 13]
    nv = {i: 4 + j for j, i in enumerate(idxs)}
    T = 12
    pairs = _targets(survey, engine, idxs, 0, obs_format, views_of=nv)
    assert len({r.n_views for r, _ in pairs}) == 6
    want = [_alone(r, tgt, T, closed, split=5) for r, tgt in pairs]
    for r, tgt in pairs:
        r.fit_init(tgt)
    rs = [r for r, _ in pairs]
    t1 = engine.fit_batch(rs, 5, use_closed_form=closed)
    t2 = engine.fit_batch(rs, T - 5, use_closed_form=closed)
    torch.cuda.synchronize()
    assert all(r.steps_done == T for r in rs)
    for i, (r, (J, p, tr)) in enumerate(zip(rs, want)):
        got = np.concatenate([t1[i].cpu().numpy(), t2[i].cpu().numpy()])
        assert np.array_equal(got, tr), (i, 'trace', np.abs(got - tr).max())
        assert np.array_equal(r.params().cpu().numpy(), p), (i, 'parameters')
        Jb = r.J().cpu().numpy()
        assert np.array_equal(np.isnan(Jb), np.isnan(J)) and np.array_equal(Jb[~np.isnan(Jb)], J[~np.isnan(J)]), (i, 'J')
    # the six fits really are six different problems
    assert len({w[2][-1, 1].item() for w in want}) == 6
    # ... and some of them have whole strips of pixels nobody observes: in closed form such a strip is a stream of ONE item,
    # which the chained streams of the batch kernel must not chain (fit.hip, StreamChain; found by this very test)
    assert sum(int(np.isnan(w[0]).all(axis=-1).sum()) >= 64 for w in want) >= 2


@pytest.mark.timeout(900)
def test_batch_of_one_and_the_oracle():
    """A batch of one image equals fit(); and the batch path against the CPU oracle (so the new launcher is pinned to the
    reference's arithmetic directly, not only to the sibling kernel)."""
    from sucre_amd import engine, synth
    scene = synth.make_scene(160, 120, 5, seed=11)
    views = engine.device_views_from_scene(scene, 'cuda')
    r = engine.Restoration(scene.height, scene.width, len(views))
    r.match(views[scene.target], views)
    J, p, tr = _alone(r, views[scene.target], 20, False)
    r.fit_init(views[scene.target])
    tb = engine.fit_batch([r], 20)[0].cpu().numpy()
    assert np.array_equal(tb, tr) and np.array_equal(r.params().cpu().numpy(), p)
    _, samples = helpers.oracle_scene_samples(scene)
    tgt = scene.views[scene.target]
    Jo, po, to = oracle.fit(scene.height, scene.width, samples, oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()), num_iter=20)
    Jb = r.J().cpu().numpy()
    assert np.array_equal(np.isnan(Jb), np.isnan(Jo)) and helpers.rms_per_channel(Jb, Jo).max() < 1e-5
    assert np.abs(tb[:, 1:] - to[:, 1:]).max() < 1e-5


@pytest.mark.timeout(900)
def test_config1_sized_batch_of_sixteen():
    """BASELINE config 1's image (640x480, 4 neighbours + self) sixteen times over -- sixteen different targets of one survey --
    in one launch per iteration: bitwise the sixteen one-by-one fits, 30 iterations; the first image also against the
    reference's own golden trajectory prefix is covered by test_gpu_baseline (same kernels' bits)."""
    from sucre_amd import engine, synth
    survey = synth.make_survey(640, 480, 6, 6, seed=0)
    idxs = [7, 8, 9, 10, 13, 14, 15, 16, 19, 20, 21, 22, 25, 26, 27, 28]
    pairs = _targets(survey, engine, idxs, 4)
    want = [_alone(r, tgt, 30, False) for r, tgt in pairs]
    for r, tgt in pairs:
        r.fit_init(tgt)
    traces = engine.fit_batch([r for r, _ in pairs], 30)
    torch.cuda.synchronize()
    for i, ((r, _), (J, p, tr)) in enumerate(zip(pairs, want)):
        assert np.array_equal(traces[i].cpu().numpy(), tr), i
        Jb = r.J().cpu().numpy()
        assert np.array_equal(np.isnan(Jb), np.isnan(J)) and np.array_equal(Jb[~np.isnan(Jb)], J[~np.isnan(J)]), i


def test_batch_refuses_what_it_cannot_do():
    from sucre_amd import _lib, engine, synth
    scene = synth.make_scene(64, 48, 3, seed=2)
    views = engine.device_views_from_scene(scene, 'cuda')
    a = engine.Restoration(48, 64, len(views))
    b = engine.Restoration(48, 64, len(views), light=True)
    for r in (a, b):
        r.match(views[scene.target], views)
        r.fit_init(views[scene.target])
    with pytest.raises(NotImplementedError):
        engine.fit_batch([a, b], 2)
    with pytest.raises(ValueError, match='own workspace'):
        engine.fit_batch([a, a], 2)
    c = engine.Restoration(48, 64, len(views))
    c.match(views[scene.target], views)
    c.fit_init(views[scene.target])
    c.fit(1)
    with pytest.raises(ValueError, match='Adam step'):   # not at the same step (a ValueError: asserts vanish under python -O)
        engine.fit_batch([a, c], 2)
    d = engine.Restoration(48, 64, len(views), obs_format='u16mm')
    d.match(views[scene.target], views)
    d.fit_init(views[scene.target])
    with pytest.raises(ValueError, match='observation format'):
        engine.fit_batch([a, d], 2)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('closed', [False, True], ids=['J-parameter', 'closed-form'])
def test_more_images_than_one_launch_holds(closed):
    """Forty 96x64 images (a launch holds 32: two launches + two tail launches per iteration), each a different target of one
    survey: bitwise the one-by-one fits."""
    from sucre_amd import engine, synth
    survey = synth.make_survey(96, 64, 8, 7, seed=9)
    idxs = [j * 8 + i for j in range(1, 6) for i in range(0, 8)]
    assert len(idxs) == 40
    pairs = _targets(survey, engine, idxs, 4)
    want = [_alone(r, tgt, 6, closed) for r, tgt in pairs]
    for r, tgt in pairs:
        r.fit_init(tgt)
    traces = engine.fit_batch([r for r, _ in pairs], 6, use_closed_form=closed)
    torch.cuda.synchronize()
    for i, ((r, _), (J, p, tr)) in enumerate(zip(pairs, want)):
        assert np.array_equal(traces[i].cpu().numpy(), tr), i
        assert np.array_equal(r.params().cpu().numpy(), p), i
        Jb = r.J().cpu().numpy()
        assert np.array_equal(np.isnan(Jb), np.isnan(J)) and np.array_equal(Jb[~np.isnan(Jb)], J[~np.isnan(J)]), i


@pytest.mark.timeout(900)
def test_randomised_batches():
    """Ten random batches: image size (ragged tiles included), number of images (1 .. 9), view counts per image, J mode, store
    format, iterations and the split of the call sequence all drawn from a seeded generator -- every image bitwise what it gets
    alone."""
    from sucre_amd import engine, synth
    rng = np.random.default_rng(20251003)
    for case in range(10):
        W, H = int(rng.integers(40, 200)), int(rng.integers(32, 150))
        n_img = int(rng.integers(1, 10))
        closed = bool(rng.integers(0, 2))
        fmt = 'u16mm' if rng.integers(0, 4) == 0 else 'f32'
        T = int(rng.integers(2, 9))
        split = int(rng.integers(0, T))
        survey = synth.make_survey(W, H, 5, 4, seed=int(rng.integers(0, 1000)))
        idxs = [int(i) for i in rng.choice(20, size=n_img, replace=False)]
        nv = {i: int(rng.integers(1, 9)) for i in idxs}
        pairs = _targets(survey, engine, idxs, 0, fmt, views_of=nv)
        want = [_alone(r, tgt, T, closed, split=split or None) for r, tgt in pairs]
        for r, tgt in pairs:
            r.fit_init(tgt)
        rs = [r for r, _ in pairs]
        if split:
            t1 = engine.fit_batch(rs, split, use_closed_form=closed)
            t2 = engine.fit_batch(rs, T - split, use_closed_form=closed)
            got = [torch.cat([a, b]) for a, b in zip(t1, t2)]
        else:
            got = engine.fit_batch(rs, T, use_closed_form=closed)
        torch.cuda.synchronize()
        for i, (r, (J, p, tr)) in enumerate(zip(rs, want)):
            label = (case, W, H, n_img, closed, fmt, T, split, i)
            assert np.array_equal(got[i].cpu().numpy(), tr, equal_nan=True), label
            assert np.array_equal(r.params().cpu().numpy(), p, equal_nan=True), label
            assert np.array_equal(r.J().cpu().numpy(), J, equal_nan=True), label
