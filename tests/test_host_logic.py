"""Host-side logic that needs no GPU: constants used by the kernels, API mirrors, loaders, COLMAP I/O, sharding."""
import inspect
import re
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers
from sucre_amd import dist as sdist
from sucre_amd import loader, se3, sfm, sucre, synth

ROOT = Path(__file__).resolve().parent.parent


def _f32(x):
    return np.float32(x)


def test_u8_to_unit_constants_are_exact_for_all_bytes():
    """fit_math.h: unit_from_u8(k) = fma(k, hi, k*lo) must equal float32(float64(k)/255) (loader.py:157,163)."""
    hi = np.float32(1.0 / 255.0)
    lo = np.float32(1.0 / 255.0 - np.float64(hi))
    k = np.arange(256, dtype=np.float32)
    t = (k * lo).astype(np.float32)
    got = (k.astype(np.float64) * np.float64(hi) + t.astype(np.float64)).astype(np.float32)  # fma: one rounding
    want = (np.arange(256, dtype=np.float64) / 255).astype(np.float32)
    assert np.array_equal(got, want)
    src = (ROOT / 'sucre_amd' / 'csrc' / 'fit_math.h').read_text()
    assert 'kInv255Hi = (float)(1.0 / 255.0)' in src and 'kInv255Lo = (float)(1.0 / 255.0 - (double)kInv255Hi)' in src


def test_depth_quantisation_is_float_division():
    kk = np.arange(65536)
    assert np.array_equal((kk.astype(np.float64) / 1000).astype(np.float32), kk.astype(np.float32) / np.float32(1000))


def test_shard_images_partitions_exactly():
    for n in (0, 1, 7, 8, 64, 513):
        ids = list(range(n))
        for world in (1, 2, 3, 8):
            parts = [sdist.shard_images(ids, r, world) for r in range(world)]
            assert sum(parts, []) == ids                       # order-preserving, disjoint, complete
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


def test_se3_exp_matches_rodrigues():
    torch.manual_seed(0)
    for _ in range(5):
        xi = torch.randn(6) * 0.5
        R, t = se3.exp(xi)
        w = xi[:3].double().numpy(); p = xi[3:].double().numpy()
        th = np.linalg.norm(w)
        Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        Rr = np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * Wx + (th - np.sin(th)) / th ** 3 * Wx @ Wx
        assert np.allclose(R.numpy(), Rr, atol=2e-6) and np.allclose(t.numpy().ravel(), V @ p, atol=5e-6)
        assert t.shape == (3, 1)
    xi = torch.zeros(6, requires_grad=True)
    R, t = se3.exp(xi)
    (R.sum() + t.sum()).backward()                            # differentiable, like the reference's
    assert xi.grad is not None and torch.isfinite(xi.grad).all()


def test_pose_and_image_geometry_roundtrip():
    scene = synth.make_scene(64, 48, 2, seed=3)
    v = scene.views[0]
    pose = sfm.Pose(v.R, v.t)
    P = torch.randn(3, 50)
    assert torch.allclose(pose.inverse().transform(pose.transform(P)), P, atol=1e-5)
    cam = sfm.Camera(1, 64, 48, scene.K)
    im = sfm.Image(1, Path('a.png'), Path('depth_a.png'), pose, cam)
    depth = v.depth_f32()
    u, vv, wP = im.unproject_depth_map(depth, to_world=True)
    px = im.project_to_view(wP)
    assert torch.allclose(px[0], u + 0.5, atol=1e-2) and torch.allclose(px[1], vv + 0.5, atol=1e-2)
    assert int((depth > 0).sum()) == u.numel()


def test_matches_container_semantics():
    cam = sfm.Camera(1, 8, 6, torch.eye(3))
    a = sfm.Image(1, Path('a.png'), Path('d.png'), sfm.Pose(torch.eye(3), torch.zeros(3, 1)), cam)
    b = sfm.Image(2, Path('b.png'), Path('d.png'), sfm.Pose(torch.eye(3), torch.zeros(3, 1)), cam)
    m1 = sfm.Matches(a, b, u1=torch.tensor([0, 1, 2]), v1=torch.tensor([0, 0, 1]), u2=torch.tensor([3, 4, 5]), v2=torch.tensor([1, 1, 2]))
    m2 = sfm.Matches(b, a, u1=torch.tensor([3, 4, 5]), v1=torch.tensor([1, 1, 2]), u2=torch.tensor([0, 7, 2]), v2=torch.tensor([0, 0, 1]))
    both = m1 & m2
    assert len(both) == 2 and both.u1.tolist() == [0, 2] and both.u2.tolist() == [3, 5]
    mp = m1.map()
    assert mp.shape == (6, 8, 2) and mp[0, 1].tolist() == [1, 4] and mp[5, 7].tolist() == [-1, -1]


def test_colmap_text_and_binary_roundtrip(tmp_path):
    import struct
    scene = synth.make_scene(64, 48, 3, seed=5)
    poses = [sfm.Pose(v.R, v.t) for v in scene.views]
    sfm.write_colmap_text(tmp_path / 'txt', scene.K, 64, 48, scene.names, poses)
    model = sfm.COLMAPModel(tmp_path / 'txt', tmp_path / 'img', tmp_path / 'depth')
    assert [im.name for im in model.images.values()] == scene.names
    for im, v in zip(model.images.values(), scene.views):
        assert torch.allclose(im.pose.R, v.R, atol=1e-6) and torch.allclose(im.pose.t, v.t, atol=1e-6)
        assert im.depth_map_path.name == 'depth_' + Path(v.name).stem + '.png'
        assert torch.equal(im.camera.K, scene.K)
    assert model[scene.names[1]].id == 2
    # the same model in COLMAP's binary layout must load identically
    bdir = tmp_path / 'bin'; bdir.mkdir()
    K = scene.K.double()
    (bdir / 'cameras.bin').write_bytes(struct.pack('<Q', 1) + struct.pack('<iiQQ4d', 1, 1, 64, 48, K[0, 0], K[1, 1], K[0, 2], K[1, 2]))
    blob = struct.pack('<Q', len(poses))
    for i, (name, pose) in enumerate(zip(scene.names, poses), start=1):
        cfw = sfm.Pose(pose.R.double(), pose.t.double()).inverse()
        q = sfm.rotmat_to_quat(cfw.R.numpy()); t = cfw.t.numpy().ravel()
        blob += struct.pack('<I4d3dI', i, *q, *t, 1) + name.encode() + b'\x00' + struct.pack('<Q', 1) + struct.pack('<ddq', 1.0, 2.0, -1)
    (bdir / 'images.bin').write_bytes(blob)
    mb = sfm.COLMAPModel(bdir, tmp_path / 'img', tmp_path / 'depth')
    for a, b in zip(model.images.values(), mb.images.values()):
        assert a.name == b.name and torch.allclose(a.pose.R, b.pose.R, atol=1e-6) and torch.allclose(a.pose.t, b.pose.t, atol=1e-6)
    # image_scale rescales intrinsics like sfm.py:193-199
    half = sfm.COLMAPModel(tmp_path / 'txt', tmp_path / 'img', tmp_path / 'depth', image_scale=0.5)
    cam = half.cameras[1]
    assert (cam.width, cam.height) == (32, 24) and torch.allclose(cam.K[:2], scene.K[:2] * 0.5)


def test_quaternion_conversion_is_consistent():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if q[0] < 0: q = -q
        R = sfm.quat_to_rotmat(*q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(R), 1.0)
        assert np.allclose(sfm.rotmat_to_quat(R), q, atol=1e-12)


def test_pixel_loaders_reproduce_reference_quantisation(tmp_path):
    from PIL import Image as PILImage
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, size=(6, 8, 3), dtype=np.uint8)
    depth = rng.integers(0, 65536, size=(6, 8)).astype(np.uint16)
    PILImage.fromarray(rgb).save(tmp_path / 'a.png')
    PILImage.fromarray(depth).save(tmp_path / 'depth_a.png')
    got = loader.load_rgb(tmp_path / 'a.png', width=8, height=6)
    assert got.dtype == torch.float32 and np.array_equal(got.numpy(), (rgb.astype(np.float64) / 255).astype(np.float32))
    assert np.array_equal(loader.load_rgb_u8(tmp_path / 'a.png', 8, 6).numpy(), rgb)
    d = loader.load_depth_map(tmp_path / 'depth_a.png', width=8, height=6)
    assert np.array_equal(d.numpy(), (depth.astype(np.float64) / 1000).astype(np.float32))
    assert loader.load_rgb_u8(tmp_path / 'a.png', 4, 3) is None        # not camera-sized: the float32 path takes over
    half = loader.load_rgb(tmp_path / 'a.png', width=4, height=3)      # --image-scale 0.5: INTER_AREA, factor 2
    x = rgb.astype(np.float64) / 255
    want = (((x[0::2, 0::2] + x[0::2, 1::2]) + x[1::2, 0::2]) + x[1::2, 1::2]) * 0.25
    assert half.dtype == torch.float32 and np.array_equal(half.numpy(), want.astype(np.float32))
    dh = loader.load_depth_map(tmp_path / 'depth_a.png', width=4, height=3)
    assert np.array_equal(dh.numpy(), (depth[0::2, 0::2].astype(np.float64) / 1000).astype(np.float32))
    assert loader.load_rgb(tmp_path / 'a.png', width=16, height=12).shape == (12, 16, 3)   # enlarging: INTER_CUBIC restated
    assert loader.load_rgb(tmp_path / 'a.png', width=5, height=4).shape == (4, 5, 3)


def test_list_backed_matches_data_keeps_reference_semantics():
    md = loader.MatchesData()
    md.append(torch.tensor([1, 2], dtype=torch.int16), torch.tensor([3, 4], dtype=torch.int16), torch.ones(3, 2), torch.zeros(3, 2))
    md.append(torch.tensor([5], dtype=torch.int16), torch.tensor([6], dtype=torch.int16), torch.ones(3, 1), torch.zeros(3, 1))
    assert len(md) == 3
    batches = list(md.iter(batch_size=2))
    assert len(batches) == 1 and batches[0][0].dtype == torch.int64 and batches[0][2].shape == (3, 3)
    assert len(list(md.iter(batch_size=1))) == 2
    img = sfm.Image(1, Path('a.png'), Path('d.png'), sfm.Pose(torch.eye(3), torch.zeros(3, 1)), sfm.Camera(1, 8, 6, torch.eye(3)))
    model = sucre.SUCRe.__new__(sucre.SUCRe)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        sucre._restoration_of(md)
    mf = loader.MatchesFile(Path('/tmp/none.h5'))
    assert len(mf) == 0
    with pytest.raises(RuntimeError, match='match_images'):
        mf.load_matches()


def test_cli_keeps_the_reference_flags_and_defaults():
    """sucre.py:265-305 of the reference."""
    want = {'image_dir': None, 'depth_dir': None, 'model_dir': None, 'output_dir': None, 'image_name': None,
            'image_list': None, 'image_ids': None, 'light_model': False, 'use_closed_form': False, 'min_cover': 1e-6,
            'image_scale': 1.0, 'filter_images_path': None, 'learning_rate': 0.05, 'num_iter': 200, 'batch_size': 5,
            'save_interval': None, 'params_path': None, 'force_compute_matches': False, 'keep_matches': False,
            'num_workers': 0, 'device': 'cuda'}
    p = sucre.build_parser()
    ns = p.parse_args(['--image-dir', 'a', '--depth-dir', 'b', '--model-dir', 'c', '--output-dir', 'd', '--image-ids', '1', '5'])
    got = vars(ns)
    assert set(got) == set(want)
    for k, v in want.items():
        if k not in ('image_dir', 'depth_dir', 'model_dir', 'output_dir', 'image_ids'):
            assert got[k] == v, k
    assert got['image_ids'] == [1, 5] and got['image_dir'] == Path('a')
    with pytest.raises(SystemExit):
        p.parse_args(['--image-dir', 'a', '--depth-dir', 'b', '--model-dir', 'c', '--output-dir', 'd'])
    with pytest.raises(SystemExit):
        p.parse_args(['--image-dir', 'a', '--depth-dir', 'b', '--model-dir', 'c', '--output-dir', 'd', '--image-name', 'x', '--image-ids', '1', '2'])


def test_api_surface_mirrors_reference_signatures():
    assert list(inspect.signature(sucre.adam).parameters)[:8] == ['sucre', 'matches_data', 'lr', 'num_iter', 'batch_size', 'save_dir', 'save_interval', 'device']
    assert list(inspect.signature(sucre.restore_image).parameters) == [
        'image', 'colmap_model', 'output_dir', 'light_model', 'use_closed_form', 'min_cover', 'image_list', 'lr',
        'num_iter', 'batch_size', 'save_interval', 'params_path', 'force_compute_matches', 'keep_matches', 'num_workers', 'device']
    # the reference's positional arguments first; light_model is an extra trailing keyword (default False)
    assert list(inspect.signature(sfm.Image.match_images).parameters)[:6] == ['self', 'image_list', 'matches_file', 'min_cover', 'num_workers', 'device']
    # match_two_way / match_one_way take the reference's arguments positionally (sfm.py:115,121)
    assert list(inspect.signature(sfm.Image.match_two_way).parameters)[:8] == ['self', 'other', 'u1', 'v1', 'wP1', 'u2', 'v2', 'wP2']
    assert list(inspect.signature(sfm.Image.match_one_way).parameters)[:5] == ['self', 'other', 'u1', 'v1', 'wP1']
    # the reference's default device is 'cpu' (sfm.py:128, sucre.py:133,176): kept in the signatures, refused loudly at run time
    for fn in (sfm.Image.match_images, sucre.restore_image, sucre.adam):
        assert inspect.signature(fn).parameters['device'].default == 'cpu', fn
    with pytest.raises(RuntimeError, match='no CPU path'):
        sfm.require_gpu('cpu', 'restore_image')
    sfm.require_gpu('cuda:3', 'x')
    assert list(inspect.signature(sfm.COLMAPModel.__init__).parameters) == ['self', 'model_dir', 'image_dir', 'depth_dir', 'image_scale']
    assert list(inspect.signature(loader.MatchesFile.__init__).parameters) == ['self', 'path', 'colmap_model', 'overwrite']
    for name in ('MatchesSample', 'MatchesData', 'MatchesFile', 'ImageDataset', 'load_rgb', 'load_depth_map', 'load_image_list'):
        assert hasattr(loader, name)
    for name in ('Pose', 'Camera', 'Image', 'Matches', 'COLMAPModel'):
        assert hasattr(sfm, name)


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under sucre_amd/ may import, load or mention it."""
    for f in (ROOT / 'sucre_amd').rglob('*'):
        if f.suffix in ('.py', '.hip', '.h', '.cpp') or f.name == 'Makefile':
            text = f.read_text()
            assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f
            assert 'libsucre_oracle' not in text and 'sucre_oracle' not in text, f
    bench = (ROOT / 'bench.py').read_text()
    assert bench.count('from oracle import oracle') == 1 and 'def cpu_baseline' in bench


def test_light_model_torch_side_matches_reference_formula():
    """SUCRe.compute_l_z with the light model (sucre.py:52-64), used by the output stage."""
    class _Img(sfm.Image):
        def get_rgb(self): return torch.rand(6, 8, 3)
        def get_depth_map(self): return torch.ones(6, 8)
    img = _Img(1, Path('a.png'), Path('d.png'), sfm.Pose(torch.eye(3), torch.zeros(3, 1)), sfm.Camera(1, 8, 6, torch.eye(3)))
    m = sucre.SUCRe(img, light_model=True)
    assert m.cam2light.shape == (6,) and torch.equal(m.sigma.detach(), torch.eye(2)) and m.water_vector().numel() == 19
    with torch.no_grad():
        m.cam2light.copy_(torch.tensor([0.02, -0.01, 0.03, 0.1, -0.2, 0.05])); m.sigma.copy_(torch.tensor([[1.2, 0.1], [-0.2, 0.9]]))
    cP = torch.tensor([[0.3, -0.5, 1.0], [0.2, 0.4, -0.1], [3.0, 2.5, 3.2]])
    l, z = m.compute_l_z(cP)
    R, t = se3.exp(m.cam2light)
    lP = R @ cP + t
    lp = (lP[:2] / lP[2]).T.unsqueeze(2)
    Sigma = m.sigma.T @ m.sigma
    want = torch.exp(-torch.flatten(lp.transpose(1, 2) @ Sigma.inverse() @ lp) / 2)
    assert torch.allclose(l, want, atol=1e-7) and torch.allclose(z, cP.norm(dim=0) + lP.norm(dim=0), atol=1e-6)
    sucre.SUCRe(img, light_model=True, use_closed_form=True)  # allowed: J is solved per iteration by the engine


def test_output_stage_matches_reference_images(golden):
    """plot_J / plot_reconstruction (sucre.py:84-112) on the reference's own fitted state give the reference's
    own 8-bit images (golden arrays produced by the reference, tests/golden/gen_golden.py)."""
    sc = golden.scene
    tgt = sc.views[sc.target]

    class _Img(sfm.Image):
        def get_rgb(self): return tgt.rgb_f32()
        def get_depth_map(self): return tgt.depth_f32()

    img = _Img(1, Path(tgt.name), Path('depth_' + tgt.name), sfm.Pose(tgt.R, tgt.t), sfm.Camera(1, sc.width, sc.height, sc.K))
    model = sucre.SUCRe(img)
    p = torch.tensor(golden['params_200'], dtype=torch.float32)
    with torch.no_grad():
        model.B.copy_(p[0:3].view(3, 1)); model.beta.copy_(p[3:6].view(3, 1)); model.gamma.copy_(p[6:9].view(3, 1))
        model.J.copy_(torch.tensor(golden['J_param_200']))
    assert np.array_equal(np.asarray(model.plot_J()), golden['plot_J_200'])
    rec = np.asarray(model.plot_reconstruction()).astype(np.int16)
    ref = golden['plot_reconstruction_200'].astype(np.int16)
    assert np.abs(rec - ref).max() <= 1 and (rec != ref).mean() < 1e-3   # float32 exp ordering may flip a rounding


def test_h5_bridge_writes_the_reference_layout(tmp_path):
    """loader.py:68-76: one group per neighbour; u1,v1,u2,v2 int16, d float32, I float32 (3,n); groups iterate
    in name order.  Verified by reading the file back with h5py itself (in-process or under the helper python)."""
    from sucre_amd import h5bridge
    if not h5bridge.available():
        pytest.skip('no h5py in this environment')
    rng = np.random.default_rng(0)
    groups = {}
    for name, n in (('img_0002.png', 7), ('img_0000.png', 0), ('img_0001.png', 3)):
        groups[name] = dict(u1=rng.integers(0, 64, n).astype(np.int16), v1=rng.integers(0, 48, n).astype(np.int16),
                            u2=rng.integers(0, 64, n).astype(np.int16), v2=rng.integers(0, 48, n).astype(np.int16),
                            d=rng.random(n).astype(np.float32) + 1, I=rng.random((3, n)).astype(np.float32))
    path = h5bridge.write_groups(tmp_path / 'm.h5', groups)
    assert path.read_bytes()[:8] == b'\x89HDF\r\n\x1a\n'            # a real HDF5 container
    back = h5bridge.read_groups(path)
    assert list(back) == sorted(groups)
    for name, ds in groups.items():
        assert set(back[name]) == set(h5bridge.DATASETS)
        for k, v in ds.items():
            assert back[name][k].dtype == v.dtype and back[name][k].shape == v.shape and np.array_equal(back[name][k], v)


def test_pixel_cache_evicts_least_recently_used(monkeypatch):
    """sfm.PIXEL_CACHE: the resident-pixel budget drops the least recently used image's cache entry (only the
    cache's reference), so a scene larger than the GPU degrades to re-decoding instead of running out of memory."""
    cache = sfm._PixelCache()
    cache._budget = 250

    class Fake:
        def __init__(self):
            self._device_view = ('dev', object())
    a, b, c = Fake(), Fake(), Fake()
    dev = torch.device('cpu')
    cache.insert(a, 100, dev); cache.insert(b, 100, dev)
    assert cache.total == 200 and not cache.full(dev)
    cache.touch(a)                      # b is now the least recently used
    cache.insert(c, 100, dev)
    assert b._device_view is None and a._device_view is not None and c._device_view is not None
    assert cache.total == 200 and list(cache.entries) == [id(a), id(c)]
    cache.forget(a)
    assert cache.total == 100
    cache.insert(a, 1000, dev)          # a single image larger than the budget still stays (it is in use)
    assert a._device_view is not None and c._device_view is None and cache.total == 1000
    # bytes added to an entry later (the float32 colour twin of a view) count too and may evict others, never itself
    cache = sfm._PixelCache(); cache._budget = 250
    a, b = Fake(), Fake()
    cache.insert(a, 100, dev); cache.insert(b, 100, dev)
    cache.grow(b, 100, dev)
    assert cache.total == 200 and a._device_view is None and b._device_view is not None
    cache.grow(b, 500, dev)
    assert cache.total == 700 and b._device_view is not None
    cache.grow(a, 50, dev)              # a is no longer cached: nothing to grow
    assert cache.total == 700


def test_matches_plot_draws_side_by_side(tmp_path):
    from PIL import Image as PILImage
    rng = np.random.default_rng(2)
    for n in ('a', 'b'):
        PILImage.fromarray(rng.integers(0, 256, size=(6, 8, 3), dtype=np.uint8)).save(tmp_path / f'{n}.png')
    cam = sfm.Camera(1, 8, 6, torch.eye(3))
    pose = sfm.Pose(torch.eye(3), torch.zeros(3, 1))
    a = sfm.Image(1, tmp_path / 'a.png', tmp_path / 'd.png', pose, cam)
    b = sfm.Image(2, tmp_path / 'b.png', tmp_path / 'd.png', pose, cam)
    m = sfm.Matches(a, b, u1=torch.tensor([0, 1, 2]), v1=torch.tensor([0, 0, 1]), u2=torch.tensor([3, 4, 5]), v2=torch.tensor([1, 1, 2]))
    pic = m.plot(step=2, color=(255, 0, 0))
    assert pic.size == (16, 6) and (np.asarray(pic) == np.array([255, 0, 0])).all(axis=2).any()


def _images_of(scene, sizes=None):
    """sfm.Image objects (no files) for a synthetic scene; sizes: optional per-view (K, W, H)."""
    from sucre_amd import sfm
    out = []
    for i, v in enumerate(scene.views):
        K, W, H = sizes[i] if sizes else (scene.K, scene.width, scene.height)
        out.append(sfm.Image(i + 1, Path(v.name), Path('depth_' + v.name), sfm.Pose(v.R, v.t), sfm.Camera(i + 1, W, H, K)))
    return out


def test_overlap_cull_never_drops_a_view_that_has_matches():
    """Image.overlapping_views (the pre-pass that sizes the workspace by overlap): over random scenes with far views,
    strong rotation noise, other sensor sizes and a camera looking away from the seabed, every view the oracle finds a
    match in survives; the far views are dropped."""
    import helpers
    from oracle import oracle
    from sucre_amd import synth
    dropped = kept_empty = 0
    for seed in range(10):
        a = synth.make_scene(96, 64, 5, seed=seed, far_views=3, rot_sigma=0.12)
        b = synth.make_scene(128, 80, 5, seed=seed, far_views=0, rot_sigma=0.12)     # same poses, larger sensor
        tgt = a.views[a.target]
        flipped = synth.SynthView(name='flip.png', R=(tgt.R @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))).contiguous(),
                                  t=tgt.t, depth_u16=tgt.depth_u16, rgb_u8=tgt.rgb_u8)
        views = list(a.views) + list(b.views) + [flipped]
        sizes = [(a.K, 96, 64)] * len(a.views) + [(b.K, 128, 80)] * len(b.views) + [(a.K, 96, 64)]
        scene = synth.SynthScene(width=96, height=64, K=a.K, views=views, target=a.target, seed=seed)
        images = _images_of(scene, sizes)
        d = tgt.depth_f32()
        images[a.target].__dict__['_depth_range'] = (float(d[d > 0].min()), float(d[d > 0].max()))
        keep = set(images[a.target].overlapping_views(images, 'cpu'))
        cam1 = oracle.make_cam(64, 96, **helpers.cam_matrices(a.K, tgt.R, tgt.t))
        for k, (v, (K, W, H)) in enumerate(zip(views, sizes)):
            cam2 = oracle.make_cam(H, W, **helpers.cam_matrices(K, v.R, v.t))
            n = len(oracle.match_view(d.numpy(), cam1, v.depth_f32().numpy(), cam2))
            if k not in keep:
                assert n == 0, (seed, k, n)
                dropped += 1
            elif n == 0:
                kept_empty += 1
        assert a.target in keep
    assert dropped >= 25, dropped          # the far views (3 per scene) go
    assert kept_empty <= 3 * 10            # and few empty ones are kept (the test is conservative, not blind)


def test_percentile_plan_and_lerp_are_numpys_percentile():
    """plot_J's percentiles on the device = two exact order statistics (GPU) + numpy's interpolation restated on the
    host: ranks, weight and lerp must reproduce np.percentile of a float32 array bit for bit (and its dtype)."""
    from sucre_amd.sucre import percentile_lerp, percentile_plan
    rng = np.random.default_rng(0)
    for trial in range(120):
        n = int(rng.integers(1, 5000)) if trial < 110 else int(rng.integers(1_000_000, 2_500_000))
        a = (rng.random(n) * float(rng.choice([1, 1e-3, 50])) - 0.2).astype(np.float32)
        s = np.sort(a)
        for q in (1, 99, 50, 0, 100):
            lo, hi, g = percentile_plan(n, q)
            got, ref = percentile_lerp(s[lo], s[hi], g), np.percentile(a, q)
            assert got == ref and got.dtype == ref.dtype == np.float32, (n, q, got, ref)
    b = np.stack([a, a[::-1], a * np.float32(0.5)], 1)          # the reference's call shape: (n, 3), axis=0
    lo, hi, g = percentile_plan(len(a), 99)
    got = np.array([percentile_lerp(np.sort(b[:, c])[lo], np.sort(b[:, c])[hi], g) for c in range(3)])
    assert np.array_equal(got, np.percentile(b, 99, axis=0))


def test_quat_to_rotmat_equals_scipy_also_for_non_unit_input():
    """pycolmap's `cam_from_world.rotation.matrix()` (sfm.py:221) = normalise, then Eigen's toRotationMatrix; scipy's
    Rotation.from_quat (scalar LAST) normalises too and is the independent check available here."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    qs = [rng.normal(size=4) * s for s in (1.0, 1.0, 1e-3, 7.5, 1.0) for _ in range(8)]
    qs += [np.array(q, float) for q in ([1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [-1, 0, 0, 0],
                                         [0.5, 0.5, 0.5, 0.5], [2, 0, 0, 0], [1, 1e-9, 0, 0], [-0.3, 0.1, -0.9, 0.2])]
    for q in qs:
        R = sfm.quat_to_rotmat(*q)
        Rs = Rotation.from_quat([q[1], q[2], q[3], q[0]]).as_matrix()
        assert np.abs(R - Rs).max() < 1e-15, (q, np.abs(R - Rs).max())
        assert np.array_equal(R, sfm.quat_to_rotmat(*(-q)))   # q and -q are the same rotation
    assert np.array_equal(sfm.quat_to_rotmat(0, 0, 0, 0), np.eye(3))   # COLMAP's NormalizeQvec on a zero quaternion


def test_images_txt_as_colmap_writes_it(tmp_path):
    """images.txt cases of a real COLMAP export: populated POINTS2D lines, several cameras, blank lines and comments
    between records, names with blanks and sub-folders; ids need not be contiguous (sfm.py:186-226)."""
    (tmp_path / 'cameras.txt').write_text(
        '# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: 2\n'
        '7 PINHOLE 640 480 500.5 501.25 320.0 240.5\n'
        '2 PINHOLE 1280 720 900 905 640.25 360\n')
    (tmp_path / 'images.txt').write_text(
        '# Image list with two lines of data per image:\n'
        '#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n'
        '# Number of images: 3, mean observations per image: 2\n'
        '12 0.9 0.1 -0.2 0.3 1.5 -2.5 3.25 7 dive 1/frame 0001.png\n'
        '10.5 20.25 -1 300.0 200.0 17 5 6 99\n'
        '\n'
        '3 1 0 0 0 0 0 0 2 b.jpg\n'
        '\n'
        '# a comment between records\n'
        '40 0.0 0.0 2.0 0.0 -1 -2 -3 7 c.JPG\n'
        '1 2 3\n')
    model = sfm.COLMAPModel(tmp_path, tmp_path / 'img', tmp_path / 'dep', image_scale=0.5)
    assert sorted(model.images) == [3, 12, 40] and sorted(model.cameras) == [2, 7]
    a, b, c = model.images[12], model.images[3], model.images[40]
    # Image.name is the file's base name (sfm.py:84), which is also what COLMAPModel[...] looks up (sfm.py:226)
    assert a.name == 'frame 0001.png' and model['frame 0001.png'] is a
    assert a.rgb_path == tmp_path / 'img' / 'dive 1' / 'frame 0001.png'
    assert a.depth_map_path == tmp_path / 'dep' / 'dive 1' / 'depth_frame 0001.png'
    assert c.depth_map_path.name == 'depth_c.png'
    assert a.camera is model.cameras[7] and c.camera is model.cameras[7] and b.camera is model.cameras[2]
    assert (a.camera.width, a.camera.height) == (320, 240) and (b.camera.width, b.camera.height) == (640, 360)
    assert torch.equal(b.camera.K, torch.tensor([[450.0, 0, 320.125], [0, 452.5, 180.0], [0, 0, 1]]))
    # poses: world-from-camera = inverse of the (normalised) cam_from_world of the file
    from scipy.spatial.transform import Rotation
    Rcw = Rotation.from_quat([0.1, -0.2, 0.3, 0.9]).as_matrix()
    assert torch.allclose(a.pose.R, torch.tensor(Rcw.T, dtype=torch.float32), atol=1e-7)
    assert torch.allclose(a.pose.t, torch.tensor(-Rcw.T @ np.array([1.5, -2.5, 3.25]), dtype=torch.float32).view(3, 1), atol=1e-6)
    assert torch.equal(b.pose.R, torch.eye(3)) and torch.equal(b.pose.t, torch.zeros(3, 1))
    assert torch.allclose(c.pose.R, torch.diag(torch.tensor([-1.0, 1.0, -1.0])))   # (0,0,2,0) -> half turn about y


def test_area_resize_without_opencv():
    """INTER_AREA restated (loader._resize_rgb) when cv2 is absent: exact block means for integer factors, the area
    integral of the piecewise-constant image for any other factor, weights that sum to one."""
    try:
        import cv2  # noqa: F401
        pytest.skip('cv2 is installed: the product path uses it')
    except ImportError:
        pass
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, size=(36, 60, 3)).astype(np.float64) / 255
    half = loader._resize_rgb(img, 30, 18)
    assert np.array_equal(half, (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2]) * 0.25)
    for (w, h) in ((45, 27), (42, 25), (17, 11), (59, 35)):
        out = loader._resize_rgb(img, w, h)
        assert out.shape == (h, w, 3)
        # brute force: integrate the source over each destination cell
        sx, sy = 60 / w, 36 / h
        ref = np.zeros_like(out)
        for y in range(h):
            for x in range(w):
                acc = 0.0
                for yy in range(int(np.floor(y * sy)), min(36, int(np.ceil((y + 1) * sy)))):
                    wy = min(yy + 1, (y + 1) * sy) - max(yy, y * sy)
                    for xx in range(int(np.floor(x * sx)), min(60, int(np.ceil((x + 1) * sx)))):
                        wx = min(xx + 1, (x + 1) * sx) - max(xx, x * sx)
                        acc = acc + img[yy, xx] * (wx * wy)
                ref[y, x] = acc / (sx * sy)
        assert np.abs(out - ref).max() < 2e-3   # OpenCV drops slivers below 1e-3 of a pixel and keeps float32 weights
        assert np.abs(loader._resize_rgb(np.ones((36, 60, 3)), w, h) - 1).max() < 1e-6
    # enlarging = INTER_CUBIC (A = -0.75): weights sum to one, constants survive, and a ramp comes out as that kernel's
    # first moment makes it (it does not reproduce linear functions: 0.25 -> 0.296875 for the tap set (-1, 0, 1, 2))
    up = loader._resize_rgb(img, 120, 72)
    assert up.shape == (72, 120, 3) and np.abs(loader._resize_rgb(np.full((36, 60, 3), 0.37), 120, 72) - 0.37).max() < 1e-6
    idx, wgt = loader._cubic_taps(60, 120)
    assert np.abs(wgt.sum(axis=1) - 1).max() < 1e-6 and idx.min() == 0 and idx.max() == 59
    d = 9                                       # samples the source at (9.5 * 0.5 - 0.5) = 4.25
    assert idx[d].tolist() == [3, 4, 5, 6] and abs(float((wgt[d] * np.arange(-1, 3)).sum()) - 0.296875) < 1e-6
    ramp = np.tile(np.arange(60, dtype=np.float64)[None, :, None], (36, 1, 3))
    assert abs(loader._resize_rgb(ramp, 120, 72)[10, d, 0] - (4 + 0.296875)) < 1e-5


def test_cubic_and_area_resize_against_torch():
    """A second, independent implementation of the two published kernels (OpenCV itself is not in this image): torch's
    bicubic interpolation uses the same A = -0.75 cubic, half-pixel centres and border replication as cv2.INTER_CUBIC (weights
    in float64 there, float32 in OpenCV and in the restatement: 1e-6), and its 'area' mode is INTER_AREA for integer factors."""
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    img = rng.random((17, 23, 3))
    t = torch.tensor(img).permute(2, 0, 1)[None]
    for (w, h) in ((46, 34), (37, 41), (24, 18), (92, 51)):
        ours = loader._resize_cubic(img, w, h)
        ref = F.interpolate(t, size=(h, w), mode='bicubic', align_corners=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(ours - ref).max() < 2e-6, (w, h, np.abs(ours - ref).max())
    big = rng.random((36, 60, 3))
    tb = torch.tensor(big).permute(2, 0, 1)[None]
    for (w, h) in ((30, 18), (20, 12), (15, 9)):
        ours = loader._resize_area(big, w, h)
        ref = F.interpolate(tb, size=(h, w), mode='area')[0].permute(1, 2, 0).numpy()
        assert np.abs(ours - ref).max() < 1e-7, (w, h)   # (the block mean's 1 / (fx fy) is a float32 constant in OpenCV: 3e-8 at factor 3)


def test_png_writer_keeps_every_pixel(tmp_path, monkeypatch):
    """sucre._save_png (the CLI's output files, sucre.py:116-121): lossless for RGB through the zlib writer, PIL for
    other modes and on request; a decoder must give back the very pixels."""
    from PIL import Image as PILImage
    rng = np.random.default_rng(7)
    for shape in ((37, 53, 3), (1, 1, 3), (2, 300, 3)):
        a = rng.integers(0, 256, size=shape, dtype=np.uint8)
        sucre._save_png(PILImage.fromarray(a), tmp_path / 'a.png')
        with PILImage.open(tmp_path / 'a.png') as im:
            assert im.mode == 'RGB' and np.array_equal(np.array(im), a)
        monkeypatch.setenv('SUCRE_PNG_WRITER', 'pil')
        sucre._save_png(PILImage.fromarray(a), tmp_path / 'b.png')
        monkeypatch.delenv('SUCRE_PNG_WRITER')
        assert np.array_equal(np.array(PILImage.open(tmp_path / 'b.png')), a)
    g = rng.integers(0, 256, size=(9, 11), dtype=np.uint8)
    sucre._save_png(PILImage.fromarray(g), tmp_path / 'g.png')
    assert np.array_equal(np.array(PILImage.open(tmp_path / 'g.png')), g)


def test_png_writer_processes(tmp_path, monkeypatch):
    """_png.WriterPool (the CLI's PNG encoding, moved out of the process that drives the GPU): files written by the
    child processes decode to the very pixels, errors come back as exceptions, workers start on demand."""
    import threading
    from PIL import Image as PILImage
    from sucre_amd import _pixelio as _png
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, size=(40 + i, 64, 3), dtype=np.uint8) for i in range(6)]
    pool = _png.WorkerPool(2)
    try:
        assert pool._procs == []
        threads = [threading.Thread(target=pool.write, args=(tmp_path / f'{i}.png', a)) for i, a in enumerate(imgs)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        assert 1 <= len(pool._procs) <= 2
        for i, a in enumerate(imgs):
            assert np.array_equal(np.array(PILImage.open(tmp_path / f'{i}.png')), a)
        with pytest.raises(FileNotFoundError, match='No such file'):
            pool.write(tmp_path / 'missing' / 'x.png', imgs[0])
        pool.write(tmp_path / 'again.png', imgs[1])      # the worker survives a failed write
        assert np.array_equal(np.array(PILImage.open(tmp_path / 'again.png')), imgs[1])
        # decoding through the workers = decoding in-process, for colour and 16-bit depth files
        depth = rng.integers(0, 65536, size=(23, 31)).astype(np.uint16)
        PILImage.fromarray(depth).save(tmp_path / 'd.png')
        got = pool.read(tmp_path / 'd.png', depth=True)
        assert got.dtype == np.uint16 and np.array_equal(got, depth)
        assert np.array_equal(pool.read(tmp_path / '0.png'), imgs[0])
        with pytest.raises(FileNotFoundError):
            pool.read(tmp_path / 'nope.png')
        victim = pool._free.get(); pool._free.put(victim)   # a worker that dies is retired, the caller is told
        victim.kill(); victim.wait()
        lost = 0
        for _ in range(3):
            try:
                pool.write(tmp_path / 'after.png', imgs[2])
            except _png.WorkerLost:
                lost += 1
        assert lost <= 1 and np.array_equal(np.array(PILImage.open(tmp_path / 'after.png')), imgs[2])
        # every worker killed at once (the OOM killer): callers neither hang nor see a short read as garbage -- each gets
        # WorkerLost at most once and the pool starts replacements
        for q in list(pool._procs):
            q.kill(); q.wait()
        results = []

        def attempt(i):
            for _ in range(3):
                try:
                    pool.write(tmp_path / f'k{i}.png', imgs[i])
                    results.append(i)
                    return
                except _png.WorkerLost:
                    continue
        threads = [threading.Thread(target=attempt, args=(i,)) for i in range(4)]
        [t.start() for t in threads]
        [t.join(timeout=60) for t in threads]
        assert not any(t.is_alive() for t in threads) and sorted(results) == [0, 1, 2, 3]
        assert all(q.poll() is None for q in pool._procs) and 1 <= len(pool._procs) <= 2
        _png.POOL = pool                                   # and loader's readers go through the pool when asked to
        monkeypatch.setenv('SUCRE_DECODE_IN_WORKERS', '1')
        try:
            assert np.array_equal(loader._imread_rgb_u8(tmp_path / '1.png'), imgs[1])
            assert np.array_equal(loader._imread_depth_u16(tmp_path / 'd.png'), depth)
        finally:
            _png.POOL = None
    finally:
        pool.close()


def test_experiment_variants_compile(tmp_path):
    """Every build-time knob of csrc/experiment.h (the ablations and tunables behind DESIGN.md section 4.2) still compiles
    for gfx950 -- through the product Makefile's own rule, so the no-scratch check applies to them too.  fit.hip reads
    the knobs (match.hip: SUCRE_EXACT_DIV); objects go to a scratch suffix and are removed."""
    import shutil
    import subprocess
    if shutil.which('hipcc') is None and not Path('/opt/rocm/bin/hipcc').exists():
        pytest.skip('no hipcc on this machine')
    csrc = ROOT / 'sucre_amd' / 'csrc'
    header = (csrc / 'experiment.h').read_text()
    # every knob in at least one build; knobs that do not exclude each other share one (a build is ~5 s)
    variants = {
        'tablate1': '-DSUCRE_EXP_NOCOMPUTE -DSUCRE_EXP_WAVE_TIMES -DSUCRE_EXP_PRIO=3 -DSUCRE_EXACT_J_ADAM=1',
        'tablate2': '-DSUCRE_EXP_NOLOAD -DSUCRE_RING=4 -DSUCRE_FIT_WAVES=4 -DSUCRE_DEAL_FIT=64,44,24,14,5 -DSUCRE_DEAL_CLOSED=64,48,32,20',
        'texactdiv': '-DSUCRE_EXACT_DIV=1 -DSUCRE_EXP_WAVE_TIMES -DSUCRE_EXP_MATCH_COUNT_ONLY',
        'tlight2': '-DSUCRE_EXP_NOLOAD -DSUCRE_LIGHT_FOLD1=1 -DSUCRE_LIGHT_FOLD2=1 -DSUCRE_EXP_LIGHT_VECTOR_BASES -DSUCRE_EXP_LIGHT_LOAD_Z -DSUCRE_EXP_LIGHT_GRID=512',
        'tscatter': '-DSUCRE_EXP_WAVE_TIMES',
        'tstorent': '-DSUCRE_STORE_NT=1 -DSUCRE_EXP_STORE_LOCAL -DSUCRE_EXP_SHFL_SUMS',
        'tbatch': '-DSUCRE_EXP_BATCH=2 -DSUCRE_EXP_NOSTORE -DSUCRE_EXP_BATCH_CHAIN=0 -DSUCRE_MIN_STRIPS=1 -DSUCRE_EXP_PLAIN_WAVE_SUMS',
        'thalf': '-DSUCRE_EXP_HALF_EXPS -DSUCRE_CLOSED_WAVES=5 -DSUCRE_EXP_NO_BATCH_CLOSED -DSUCRE_DEAL_CLOSED=64,48,32,20,10',
    }
    objects = {'texactdiv': 'match', 'tlight1': 'light', 'tlight2': 'light', 'tscatter': 'compact'}   # the source a knob lives in (default: fit)
    for macro in re.findall(r'#\s*if(?:n?def)\s+(SUCRE_[A-Z_0-9]+)', header):   # every knob of the header is exercised here
        assert any(macro in flags for flags in variants.values()) or macro in ('SUCRE_CLOSED_WAVES', 'SUCRE_DMA_POLICY'), macro
    # no other build-time switch hides in the kernel sources
    for f in csrc.glob('*.hip'):
        assert not re.search(r'#\s*if(?:n?def)?\s+.*SUCRE_(EXP|STATE|OBS_VIEW)', f.read_text()), f
    try:
        for name, flags in variants.items():
            obj = f"{objects.get(name, 'fit')}_{name}.o"
            out = subprocess.run(['make', '-C', str(csrc), f'VARIANT={name}', f'EXTRA={flags}', obj],
                                 capture_output=True, text=True)
            assert out.returncode == 0, (name, out.stdout[-1500:], out.stderr[-1500:])
            assert (csrc / obj).exists()
    finally:
        for name in variants:
            for f in csrc.glob(f'*_{name}.*'):
                f.unlink()


def test_the_deal_gives_every_strip_to_exactly_one_wave(tmp_path):
    """csrc/layout.h deal_walk / deal_rounds on the host (tests/native/deal_check.cpp, g++): equal shares (the product) and the
    unequal ones of the SUCRE_DEAL_FIT knob, full and partial grids, 1 .. 131072 strips."""
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('no g++ on this machine')
    csrc = ROOT / 'sucre_amd' / 'csrc'
    for name, flags in (('product', []), ('equal', ['-DSUCRE_DEAL_FIT=64,64,64,64,64', '-DSUCRE_DEAL_CLOSED=64,64,64,64']), ('unequal', ['-DSUCRE_DEAL_FIT=64,44,24,14,5', '-DSUCRE_DEAL_CLOSED=64,48,32,20'])):
        exe = tmp_path / f'deal_{name}'
        out = subprocess.run(['g++', '-std=c++17', '-O1', f'-I{csrc}', f'-I{ROOT / "include"}', *flags, str(ROOT / 'tests' / 'native' / 'deal_check.cpp'), '-o', str(exe)],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        run = subprocess.run([str(exe)], capture_output=True, text=True)
        assert run.returncode == 0 and '0 violations' in run.stdout, (name, run.stdout, run.stderr)


def test_persistent_kernels_fit_the_grids_they_are_launched_with():
    """The fit kernels deal their strips statically over a launch's waves and are launched with compile-time grids
    (csrc/layout.h: kFitGrid = 5, kClosedGrid = 4 workgroups per CU): a workgroup that is not resident from the start
    would run its whole share in a second round (seen in the light model, which now asks the runtime).  The build's
    resource reports must therefore show the occupancy the grids assume, and no scratch."""
    csrc = ROOT / 'sucre_amd' / 'csrc'
    rpt = (csrc / 'fit.rpt')
    if not rpt.exists():
        pytest.skip('no resource report next to the objects (csrc/Makefile writes fit.rpt when it compiles fit.hip)')
    layout = (csrc / 'layout.h').read_text() + (csrc / 'experiment.h').read_text()
    fit_waves = int(re.search(r'#define SUCRE_FIT_WAVES (\d+)', layout).group(1))
    closed_waves = int(re.search(r'#define SUCRE_CLOSED_WAVES (\d+)', layout).group(1))
    seen = {}
    for m in re.finditer(r'Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)', rpt.read_text(), re.S):
        name, scratch, occ = m.group(1), int(m.group(2)), int(m.group(3))
        assert scratch == 0, name
        if 'fit_grad_kernel' in name:
            need = fit_waves
        elif 'fit_closed_kernel' in name:
            need = closed_waves
        elif 'group_iter_kernelILi0' in name:
            need = fit_waves
        elif 'group_iter_kernelILi1' in name:
            need = closed_waves
        else:
            continue
        seen[name] = occ
        assert occ >= need, (name, occ, need)
    assert sum('fit_grad_kernel' in n for n in seen) >= 4 and sum('group_iter_kernel' in n for n in seen) >= 4


# ---- round 4: a matches file in the reference's format, written with the reference's own h5py call sequence ---------------

H5_FIXTURE = helpers.GOLDEN_DIR / 'ref_layout_plane_64x48_n4.h5'


def _need_h5py():
    from sucre_amd import h5bridge
    if not h5bridge.available():
        pytest.skip('no h5py interpreter on this machine (SUCRE_H5PY_PYTHON)')
    return h5bridge


def test_reference_written_matches_file_reads_back_as_the_reference_match_lists():
    """tests/golden/ref_layout_plane_64x48_n4.h5 holds the reference's own matches of the plane fixture, written by the
    h5py calls of loader.py:68-87 in their order (groups created in REVERSE name order, I NaN-prefilled then overwritten):
    h5bridge must hand the groups back in name order with the reference's datasets, dtypes and values."""
    h5bridge = _need_h5py()
    fx = helpers.load_fixture('plane_64x48_n4')
    groups = h5bridge.read_groups(H5_FIXTURE)
    kept = [str(n) for n, k in zip(fx['names'], fx['kept']) if k]
    assert list(groups) == sorted(kept)
    for k, name in enumerate(str(n) for n in fx['names']):
        if not fx['kept'][k]:
            assert name not in groups
            continue
        ds = groups[name]
        assert sorted(ds) == ['I', 'd', 'u1', 'u2', 'v1', 'v2']
        u1, v1, u2, v2 = fx.match_lists(k)
        for key, want in (('u1', u1), ('v1', v1), ('u2', u2), ('v2', v2)):
            assert ds[key].dtype == np.int16 and np.array_equal(ds[key], want), (name, key)
        view = fx.scene.views[k]
        assert ds['d'].dtype == np.float32 and np.array_equal(ds['d'], view.depth_f32().numpy()[v2.astype(np.int64), u2.astype(np.int64)])
        assert ds['I'].dtype == np.float32 and ds['I'].shape == (3, len(u1))
        assert np.array_equal(ds['I'], view.rgb_f32().numpy()[v2.astype(np.int64), u2.astype(np.int64)].T)


def test_half_written_reference_matches_file_is_refused():
    """A crash between save_matches and prepare_matches leaves I NaN-prefilled (loader.py:76); the reference's
    check_integrity trips on it (loader.py:89-101) and so must load_file -- before anything reaches the engine."""
    _need_h5py()
    fx = helpers.load_fixture('plane_64x48_n4')
    sc = fx.scene
    model = {v.name: helpers.synth_image(i + 1, v, sc.K, sc.width, sc.height) for i, v in enumerate(sc.views)}
    mf = loader.MatchesFile(helpers.GOLDEN_DIR / 'ref_layout_plane_64x48_n4_unprepared.h5', colmap_model=model)
    assert mf.on_disk()
    with pytest.raises(AssertionError, match=r'dataset /img_\d+\.png/I contains NaN'):
        mf.load_file(model[sc.views[sc.target].name], device='cuda')


def test_colmap_binary_model_written_from_the_format_description():
    """(f)3: tests/golden/colmap_bin_model/*.bin was written field by field from COLMAP's documented binary layout by
    tests/golden/gen_colmap_bin_fixture.py -- no code shared with sfm.py's reader or its text writer -- with the
    expected world-from-camera poses computed by scipy from the (qvec, tvec) it wrote.  COLMAPModel must read it the
    way the reference reads pycolmap's Reconstruction (sfm.py:186-238): PINHOLE K per camera, pose = cam_from_world
    inverted, name -> image, depth path naming."""
    root = helpers.GOLDEN_DIR / 'colmap_bin_model'
    exp = np.load(root / 'expected.npz')
    model = sfm.COLMAPModel(root, Path('/data/images'), Path('/data/depth'))
    assert sorted(model.cameras) == sorted(exp['camera_ids'].tolist())
    for cid, (w, h), (fx, fy, cx, cy) in zip(exp['camera_ids'], exp['camera_wh'], exp['camera_params']):
        cam = model.cameras[int(cid)]
        assert (cam.width, cam.height) == (int(w), int(h))
        assert torch.equal(cam.K, torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32))
    assert sorted(model.images) == sorted(exp['image_id'].tolist()) and len(model.images) == 6
    for i, name in enumerate(str(n) for n in exp['names']):
        im = model[Path(name).name]       # the reference keys its images by the file's base name (sfm.py:84, 224)
        assert im.id == int(exp['image_id'][i]) and im.camera.id == int(exp['camera_id'][i]) and im.name == Path(name).name
        assert im.rgb_path == Path('/data/images') / name
        assert im.depth_map_path == (Path('/data/depth') / name).with_name('depth_' + Path(name).stem + '.png')
        assert np.abs(im.pose.R.numpy() - exp['R_wfc'][i]).max() < 2e-7      # float32 of a float64 rotation
        assert np.abs(im.pose.t.numpy().ravel() - exp['t_wfc'][i]).max() < 1e-6
    # half resolution: the reference's integer size and per-axis intrinsic scale (sfm.py:193-199)
    half = sfm.COLMAPModel(root, Path('/i'), Path('/d'), image_scale=0.5)
    cam = half.cameras[7]
    assert (cam.width, cam.height) == (960, 540) and abs(float(cam.K[0, 0]) - 1497.6 / 2) < 1e-3 and abs(float(cam.K[1, 2]) - 540.5 / 2) < 1e-3


def test_a_perturbed_scene_is_a_red_test(monkeypatch):
    """VERDICT round 4, weak point 1: a host that renders another scene than the one the reference ran on must get a RED
    test that names the differing view -- not a looser bar.  One colour byte of one view flipped (the smallest difference a
    libm could make) -> helpers.Baseline refuses to exist, so no reference-pinned check can run, pass or relax."""
    monkeypatch.setenv('SUCRE_TEST_PERTURB_SCENE', '2')
    monkeypatch.setattr(helpers, '_SCENES', {})
    monkeypatch.setattr(helpers, '_BASELINES', {})
    with pytest.raises(helpers.SceneMismatch) as e:
        helpers.load_baseline(helpers.BASELINE_ODD)
    msg = str(e.value)
    assert '1 of 9 views differ' in msg and 'img_0002.png' in msg, msg
    assert helpers.BASELINE_ODD not in helpers._BASELINES and not helpers._SCENES   # nothing cached for a later test to pick up
    monkeypatch.delenv('SUCRE_TEST_PERTURB_SCENE')
    b = helpers.load_baseline(helpers.BASELINE_ODD)
    assert b.inputs_identical and all(b.views_identical)
    # ... and the helpers hold no escape hatch any more
    src = (Path(helpers.__file__)).read_text() + (Path(helpers.__file__).parent / 'test_gpu_baseline.py').read_text()
    assert 'not b.inputs_identical' not in src and 'max(rms_bar' not in src


def test_engine_plumbing_attributes_are_what_their_callers_expect():
    """No GPU needed: the launch helpers of engine.Restoration are called as methods (``self._sp()``) and its layout helpers
    are read as properties (``self._geom``, ``self._ext_mode``, ``self._ext_flag``) at ~40 call sites that only run on a GPU."""
    from sucre_amd import engine
    R = engine.Restoration
    assert callable(inspect.getattr_static(R, '_sp')) and not isinstance(inspect.getattr_static(R, '_sp'), property)
    for name in ('_geom', '_ext_mode', '_ext_flag'):
        assert isinstance(inspect.getattr_static(R, name), property), name
    assert callable(inspect.getattr_static(engine.HipWaterGroup, '_sp'))
    src = inspect.getsource(engine)
    assert '_stream_ptr()' not in inspect.getsource(R), 'Restoration launches go through self._sp() (stream bookkeeping)'
    assert engine.current_slot() == 0 and '_SLOT' not in src


def test_bench_presets_respect_explicit_flags(monkeypatch):
    """``bench.py --config N`` presets: BASELINE config 1 batches 32 consecutive steps per fit launch and runs whole batches,
    unless the caller says otherwise -- also when what the caller says equals a flag's default; how the steps are launched never
    changes which BASELINE configuration the JSON names."""
    import importlib
    bench = importlib.import_module('bench')

    def parse(*argv):
        monkeypatch.setattr('sys.argv', ['bench.py', *argv])
        return bench.parse()
    a = parse()
    assert (a.width, a.height, a.neighbours, a.fit_batch, a.steps, a.warmup) == (1920, 1080, 64, 1, 10, 2) and bench.baseline_config(a) == 2
    a = parse('--config', '1')
    assert (a.width, a.height, a.neighbours, a.fit_batch, a.steps, a.warmup) == (640, 480, 4, 32, 64, 32) and bench.baseline_config(a) == 1
    a = parse('--config', '1', '--fit-batch', '1', '--steps', '10')
    assert (a.fit_batch, a.steps, a.warmup) == (1, 10, 32) and bench.baseline_config(a) == 1
    a = parse('--config', '1', '--warmup=2')
    assert a.warmup == 2 and a.steps == 64
    a = parse('--config', '5')
    assert (a.width, a.height, a.neighbours, a.obs_format) == (3840, 2160, 256, 'u16mm') and bench.baseline_config(a) == 5
    a = parse('--config', '4')
    assert a.shared_water and a.batch_images == 64 and bench.baseline_config(a) == 4
    a = parse('--gpus', '8', '--steps', '5', '--warmup', '1')      # the driver's own line
    assert (a.gpus, a.steps, a.warmup, a.fit_batch) == (8, 5, 1, 1) and bench.baseline_config(a) == 2


def test_bench_finds_the_profile_of_exactly_its_own_mode(monkeypatch, tmp_path):
    """``roofline.traffic`` / ``profile_frac`` come from profiles/rNN_<mode>_traffic.json of the mode the flags describe -- the
    mode EXACTLY: until round 5 a glob made ``--use-closed-form`` read the light + closed-form kernel's file (VERDICT round 5,
    weak point 4).  Every mode of tools/profile.sh maps to files of its own name only, newest round first."""
    import importlib
    import json
    import re
    bench = importlib.import_module('bench')

    def tag(*argv):
        monkeypatch.setattr('sys.argv', ['bench.py', *argv])
        return bench.profile_tag(bench.parse())
    modes = {'jparam': (), 'closed': ('--use-closed-form',), 'light': ('--light-model',), 'light_closed': ('--light-model', '--use-closed-form'),
             'u16mm_4k': ('--config', '5'), 'shared4': ('--shared-water', '--batch-images', '4'), 'jparam_batch32': ('--config', '1'),
             'closed_batch32': ('--config', '1', '--use-closed-form'), 'jparam_f32plain': ('--obs-format', 'f32plain'),
             'shared64': ('--config', '4'), 'jparam_f32z26': ('--obs-format', 'f32z26'), 'jparam_deep': ('--scene', 'deep'),
             'jparam_deep_f32z26': ('--scene', 'deep', '--obs-format', 'f32z26')}
    for mode, argv in modes.items():
        assert tag(*argv) == mode
    # every mode tools/profile.sh knows is one bench.py can name (the in-flight variant is the default command's second profile)
    sh = (helpers.ROOT / 'tools' / 'profile.sh').read_text()
    for mode in re.findall(r'^\s+(\w+)\) echo', sh, flags=re.M):
        assert mode in modes or mode == 'jparam_inflight2', mode
    for name in ('r05_closed', 'r05_light_closed', 'r04_closed', 'r06_closed', 'r06_closed_batch32', 'r05_jparam', 'r02', 'r05_jparam_inflight2',
                 'r05_jparam_batch32', 'r06_jparam_f32plain', 'r10_closed'):
        (tmp_path / f'{name}_traffic.json').write_text(json.dumps({'n_obs': 1}))
    (tmp_path / 'r05_closed_summary.txt').write_text('')
    got = {m: [f.name for f in bench.profile_candidates(m, tmp_path)] for m in modes}
    assert got['closed'] == ['r10_closed_traffic.json', 'r06_closed_traffic.json', 'r05_closed_traffic.json', 'r04_closed_traffic.json']
    assert got['light_closed'] == ['r05_light_closed_traffic.json']
    assert got['closed_batch32'] == ['r06_closed_batch32_traffic.json']
    assert got['jparam'] == ['r05_jparam_traffic.json', 'r02_traffic.json'] and got['jparam_batch32'] == ['r05_jparam_batch32_traffic.json']
    assert got['jparam_f32plain'] == ['r06_jparam_f32plain_traffic.json'] and got['shared64'] == []
    # the committed tree: the closed-form line reads the closed-form kernel's file
    newest = bench.profile_candidates('closed')[0]
    assert json.loads(newest.read_text())['kernel'].startswith('fit_closed_kernel'), newest.name


def test_automatic_fit_batch_is_bounded_by_free_device_memory(monkeypatch):
    """SUCRE_FIT_BATCH=auto: 8 small images per launch -- unless 8 x in_flight of their workspaces would not fit into the memory
    the device has free: then fewer (a 1280x720 survey with hundreds of neighbours)."""
    import types
    from sucre_amd import _lib, sucre
    monkeypatch.delenv('SUCRE_FIT_BATCH', raising=False)
    images = [types.SimpleNamespace(camera=types.SimpleNamespace(width=1280, height=720))] * 20
    one = int(_lib.load().sucre_workspace_bytes(720, 1280, 320))   # (300 views: a capacity of 320)
    assert one > 2 * 2**30
    for free, want in ((10**13, 8), (one * 16 / 0.8 * 1.01, 8), (one * 16 / 0.8 * 0.99, 4), (one * 5.5, 2), (one, 1)):
        monkeypatch.setattr(torch.cuda, 'mem_get_info', lambda device=None, f=free: (int(f), int(f)))
        monkeypatch.setattr(torch.cuda, 'memory_reserved', lambda device=None: 0)
        monkeypatch.setattr(torch.cuda, 'memory_allocated', lambda device=None: 0)
        assert sucre.fit_batch_size(images, in_flight=2, n_views=300) == want, (free / one, want)
    assert sucre.fit_batch_size(images, light_model=True, n_views=300) == 1
    big = [types.SimpleNamespace(camera=types.SimpleNamespace(width=1920, height=1080))]
    assert sucre.fit_batch_size(big, n_views=65) == 1
    monkeypatch.setenv('SUCRE_FIT_BATCH', '5')
    assert sucre.fit_batch_size(images, n_views=300) == 5
