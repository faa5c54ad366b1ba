"""BASELINE config 3 at its own size: 32 different target images of ONE synthetic survey, 1920x1080, 64 neighbours + self
each, restored through the code path ``bench.py --config 3`` runs (bench.survey_jobs; two images in flight on their own
HIP streams, two workspaces reused image after image, 200 Adam iterations each).

  * every image's per-view match counts (32 x 65 pairs) equal the oracle's ``match_view`` counts, and its observation
    count their sum;
  * every image's J and trace are, bit for bit, what restoring that image ALONE in a newly allocated workspace gives
    (workspace reuse and two slots in flight change nothing);
  * two of the images (the first and the last) against the oracle's own 5-iteration fit on the same inputs, at the bars
    of tests/test_gpu_fullsize.py.
"""
import copy
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(3000)
def test_config3_thirty_two_1080p_images_of_one_survey():
    sys.path.insert(0, str(ROOT))
    import bench
    from sucre_amd import engine, synth
    W, H, NEIGH, N_IMAGES, T, S = 1920, 1080, 64, 32, 200, 2
    survey, all_views, jobs, centre = bench.survey_jobs(synth, engine, W, H, NEIGH, N_IMAGES, 0, 'cuda')
    assert len(jobs) == 32 and all(len(v) == 65 for _, v in jobs) and len(set(centre)) == 32
    torch.cuda.synchronize()

    # -- the bench's own loop: image i on slot i % S (own stream + own workspace), nothing waits for anything -----------------
    restos = [engine.Restoration(H, W, NEIGH + 1) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    got = []
    for i, (tgt, views) in enumerate(jobs):
        with torch.cuda.stream(streams[i % S]):
            r = restos[i % S]
            r.match(tgt, views, min_cover=1e-6)
            r.fit_init(tgt)
            trace = r.fit(T, record_trace=True)
            got.append((r.J(), trace, r.view_counts().clone(), r.n_obs_device().clone()))
    torch.cuda.synchronize()
    del restos

    # -- every image alone, in a workspace allocated for it ---------------------------------------------------------------------
    for i, (tgt, views) in enumerate(jobs):
        r = engine.Restoration(H, W, NEIGH + 1)
        r.match(tgt, views, min_cover=1e-6)
        r.fit_init(tgt)
        trace = r.fit(T, record_trace=True)
        J = r.J()
        torch.cuda.synchronize()
        assert torch.equal(r.view_counts(), got[i][2]) and torch.equal(r.n_obs_device(), got[i][3]), (i, 'counts')
        assert torch.equal(trace, got[i][1]), (i, 'trace')
        assert torch.equal(torch.isnan(J), torch.isnan(got[i][0])) and torch.equal(torch.nan_to_num(J), torch.nan_to_num(got[i][0])), (i, 'J')
        del r
    counts = [g[2].cpu().numpy().tolist() for g in got]
    n_obs = [int(g[3].item()) for g in got]
    assert all(sum(c) == n for c, n in zip(counts, n_obs))   # min_cover 1e-6 drops nothing here
    traces = [g[1].cpu().numpy() for g in got]
    del got

    # -- the oracle: match counts of all 32 x 65 pairs -----------------------------------------------------------------------------
    host = {}

    def host_view(q):
        if q not in host:
            v = copy.copy(survey.views[q])
            v.depth_u16, v.rgb_u8 = v.depth_u16.cpu(), v.rgb_u8.cpu()
            host[q] = v
        return host[q]
    cams = {}

    def cam(q):
        if q not in cams:
            cams[q] = helpers.oracle_cam(survey, host_view(q))
        return cams[q]
    depth = {}

    def depth_f32(q):
        if q not in depth:
            depth[q] = host_view(q).depth_f32().numpy()
        return depth[q]
    for i, idx in enumerate(centre):
        sel = survey.neighbours(idx, NEIGH)
        want = [len(oracle.match_view(depth_f32(idx), cam(idx), depth_f32(q), cam(q))) for q in sel]
        assert counts[i] == want, (i, idx, [(a, b) for a, b in zip(counts[i], want) if a != b][:5])
    print(f'config 3: 32 images x 65 views, match counts of all {32 * 65} pairs equal the oracle; n_obs {min(n_obs)} .. {max(n_obs)}')

    # -- two of them against the oracle's fit ----------------------------------------------------------------------------------------
    for i in (0, N_IMAGES - 1):
        idx = centre[i]
        sel = survey.neighbours(idx, NEIGH)
        sc = synth.SynthScene(width=W, height=H, K=survey.K, views=[host_view(q) for q in sel], target=sel.index(idx), seed=0)
        per_view, samples = helpers.oracle_scene_samples(sc)
        tgt_h = sc.views[sc.target]
        Jo, po, to = oracle.fit(H, W, samples, oracle.init_J(tgt_h.rgb_u8.numpy(), tgt_h.depth_f32().numpy()), num_iter=5)
        tgt, views = jobs[i]
        r = engine.Restoration(H, W, NEIGH + 1)
        r.match(tgt, views)
        r.fit_init(tgt)
        t5 = r.fit(5).cpu().numpy()
        J = r.J().cpu().numpy()
        del r
        assert np.array_equal(t5, traces[i][:5])      # the 200-iteration run's first five rows are this run
        assert np.array_equal(np.isnan(J), np.isnan(Jo))
        rms = helpers.rms_per_channel(J, Jo)
        dpar = np.abs(t5[:, 1:] - to[:, 1:]).max()
        dcost = np.abs(t5[:, 0] / to[:, 0] - 1).max()
        print(f'config 3 image {i} (survey view {idx}): n_obs={n_obs[i]} rms(J)={rms} max|dparams|={dpar:.2e} max rel dcost={dcost:.2e}')
        assert rms.max() < 1e-5 and dpar < 1e-5 and dcost < 1e-5
