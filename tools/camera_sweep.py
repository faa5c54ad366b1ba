#!/usr/bin/env python3
"""Randomised sweep of the MATCHING against the CPU oracle with everything the reference's Image.match_one_way accepts
(sfm.py:115-119 uses other.camera): neighbour views from other cameras (other sensor sizes and focal lengths), camera
matrices that are not of the pinhole form (skew, K[2][2] != 1: the general FMA chains of csrc/match.hip instead of the
short ones), cameras looking away or sitting behind the scene (no in-front-of-camera test in the reference: negative
quotients, points behind the camera), poses with large random twists.  Match maps and counts must be bit-identical.
    python3 tools/camera_sweep.py [n_cases] [seed0]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import helpers  # noqa: E402
from oracle import oracle  # noqa: E402
from sucre_amd import engine, sfm, synth  # noqa: E402


def random_rotation(rng, sigma):
    w = rng.normal(0, sigma, 3)
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    rng = np.random.default_rng(seed0)
    t0 = time.time()
    stats = {'views': 0, 'matches': 0, 'general_K': 0, 'behind': 0, 'other_size': 0}
    for case in range(n_cases):
        seed = seed0 + case
        W, H = int(rng.integers(17, 220)), int(rng.integers(17, 160))
        nn = int(rng.integers(1, 8))
        a = synth.make_scene(W, H, nn, seed=seed, relief=float(rng.choice([0.0, 0.15, 0.6])), rot_sigma=float(rng.choice([0.0, 0.05])),
                             far_views=int(rng.integers(0, 2)))
        tgt = a.views[a.target]
        others = [(v, a.K, H, W) for v in a.views]
        # the same survey through another camera: same poses (same seed), another sensor and focal length
        W2, H2 = int(rng.integers(17, 260)), int(rng.integers(17, 200))
        b = synth.make_scene(W2, H2, nn, seed=seed, relief=0.15)
        others += [(v, b.K, H2, W2) for v in b.views]
        stats['other_size'] += len(b.views)
        # cameras looking away / twisted hard / behind the seabed
        for v in list(a.views)[:3]:
            Rr = torch.from_numpy(random_rotation(rng, float(rng.choice([0.3, 1.5, 3.0]))).astype(np.float32))
            tt = v.t + torch.from_numpy(rng.normal(0, float(rng.choice([0.0, 0.5, 3.0])), v.t.shape).astype(np.float32))
            others.append((synth.SynthView(name=f'twist_{len(others)}.png', R=(v.R @ Rr).contiguous(), t=tt, depth_u16=v.depth_u16, rgb_u8=v.rgb_u8),
                           a.K, H, W))
            stats['behind'] += 1
        # a target / neighbours whose K is not of the pinhole form
        tK = a.K.clone()
        if rng.random() < 0.5:
            tK[0, 1] = float(rng.normal(0, 2.0))          # skew
            stats['general_K'] += 1
        gen = []
        for (v, K, h, w) in others:
            if rng.random() < 0.25:
                K = K.clone()
                K[0, 1] = float(rng.normal(0, 1.0))
                if rng.random() < 0.3:
                    K = K * float(rng.choice([0.5, 2.0]))  # K[2][2] != 1 (a scaled homogeneous matrix: same projection)
                stats['general_K'] += 1
            gen.append((v, K, h, w))
        others = gen

        def dv(v, K):
            return engine.DeviceView(depth=v.depth_f32().cuda().contiguous(), rgb=v.rgb_u8.cuda().contiguous(), K=K, R=v.R, t=v.t)
        views = [dv(v, K) for v, K, h, w in others]
        r = engine.Restoration(H, W, len(views))
        r.match(dv(tgt, tK), views)
        cam1 = oracle.make_cam(H, W, **helpers.cam_matrices(tK, tgt.R, tgt.t))
        counts = []
        for k, (v, K, h, w) in enumerate(others):
            cam2 = oracle.make_cam(h, w, **helpers.cam_matrices(K, v.R, v.t))
            m = oracle.match_view(tgt.depth_f32().numpy(), cam1, v.depth_f32().numpy(), cam2)
            ref = np.full((H, W), -1, np.int32)
            ref[m.v1.astype(np.int64), m.u1.astype(np.int64)] = m.v2.astype(np.int32) * w + m.u2.astype(np.int32)
            got = r.match_map(k).cpu().numpy()
            assert np.array_equal(got, ref), (case, k, 'match map', int((got != ref).sum()), (W, H), (w, h), K.tolist())
            counts.append(len(m))
        assert r.view_counts().cpu().numpy().tolist() == counts, (case, 'counts')
        # the host's overlap cull (sfm.Image.overlapping_views sizes the CLI's workspace by it): a view it drops has no match
        images = [sfm.Image(i + 1, Path(v.name), Path('depth_' + v.name), sfm.Pose(v.R, v.t), sfm.Camera(i + 1, w, h, K))
                  for i, (v, K, h, w) in enumerate(others)]
        timg = sfm.Image(10_000, Path(tgt.name), Path('depth_' + tgt.name), sfm.Pose(tgt.R, tgt.t), sfm.Camera(10_000, W, H, tK))
        d = tgt.depth_f32()
        if bool((d > 0).any()):
            timg.__dict__['_depth_range'] = (float(d[d > 0].min()), float(d[d > 0].max()))
            keep = set(timg.overlapping_views(images, 'cpu'))
            for k, n in enumerate(counts):
                if k not in keep:
                    assert n == 0, (case, k, 'the cull dropped a view with matches', n)
                    stats['culled'] = stats.get('culled', 0) + 1
                elif n == 0:
                    stats['kept_empty'] = stats.get('kept_empty', 0) + 1
        stats['views'] += len(others)
        stats['matches'] += int(sum(counts))
        if (case + 1) % 20 == 0:
            print(f'{case + 1} cases ok, {stats}, {time.time() - t0:.0f}s', flush=True)
    print('camera sweep ok', stats)


if __name__ == '__main__':
    main()
