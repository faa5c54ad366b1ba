"""BASELINE config 4 at its per-rank size on the one GPU of the box: 64 images of ONE survey, 1920x1080 x 65 views each,
all sharing B, beta, gamma -- 64 workspaces (~125 GB of HBM), one launch per iteration over all of them
(engine.HipWaterGroup), the all-reduce of `dist.fit_shared_water` in place (a no-op at one rank; over RCCL at one rank in
tests/test_gpu_rccl.py, over 8 ranks on the driver's node).

  * `bench.py --config 4` itself (the command the driver would run on every rank), one step: its JSON line names the
    workload, the 64 images, the per-rank workspace bytes, and sustains a plausible rate;
  * a size-independent property of the shared-water objective at that size: a group made of 64 COPIES of one image has, for
    the water parameters, the one-image problem's gradient (every sum is 64 x the one-image sum, n_obs_total 64 x n_obs), and
    the 64 restored images must be bit-identical to each other.  In closed-form mode (J re-solved exactly, whatever the
    scale) the whole trajectory must therefore equal the one-image group's to rounding.  With J as a parameter only the
    first iteration does: the objective divides by the TOTAL observation count (sucre.py:145 with tied modules), so a
    pixel's J gradient is 64 x smaller -- down at Adam's eps, where the step is proportional to the gradient -- and J moves
    more slowly in a large scene; that is the composition of the reference's modules, not an artefact (the oracle does the same).
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(1500)
def test_bench_config4_preset_runs_its_per_rank_share():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--config', '4', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--num-iter', '20']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    cfg = rec['config']
    assert cfg['baseline_config'] == 4 and '64-image scene, 64 per rank' in cfg['workload'] and 'shared water parameters' in cfg['workload']
    assert cfg['workspace_bytes_per_rank'] > 64 * 1.5e9 and rec['n_gpus'] == 1
    assert rec['roofline']['kernel'] == 'group_iter_kernel' and rec['value'] > 0
    print(f"bench.py --config 4 (20 iterations): {rec['ms_per_step']:.0f} ms per 64-image step, group launch "
          f"{rec['roofline']['ms_per_launch'] * 1e3:.0f} us = {rec['roofline']['frac']:.3f} of 8 TB/s, "
          f"workspaces {cfg['workspace_bytes_per_rank'] / 2**30:.1f} GiB")


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('closed', [False, True], ids=['J-parameter', 'closed-form'])
def test_group_of_64_copies_is_the_one_image_problem(closed):
    from sucre_amd import dist as sdist
    from sucre_amd import engine, synth
    T, N = 6, 64
    scene = synth.make_scene(1920, 1080, 64, seed=0, device='cuda')
    views = engine.device_views_from_scene(scene, 'cuda')
    tgt = views[scene.target]

    def run(n_images):
        rs = []
        for _ in range(n_images):
            r = engine.Restoration(scene.height, scene.width, len(views))
            r.match(tgt, views)
            r.fit_init(tgt)
            rs.append(r)
        trace = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
        sdist.fit_shared_water(engine.HipWaterGroup(rs, use_closed_form=closed, trace=trace), T)
        torch.cuda.synchronize()
        Js = [r.J() for r in rs]
        n = rs[0].n_obs()
        return trace.cpu().numpy(), Js, n

    t1, J1, n1 = run(1)
    J1 = J1[0].cpu().numpy()
    tN, JN, nN = run(N)
    assert nN == n1 and n1 > 70_000_000
    for J in JN[1:]:   # same data, same parameters, same kernel: the same bits in every copy
        assert torch.equal(torch.nan_to_num(J), torch.nan_to_num(JN[0])) and torch.equal(torch.isnan(J), torch.isnan(JN[0]))
    J0 = JN[0].cpu().numpy()
    dcost = np.abs(tN[:, 0] / (N * t1[:, 0]) - 1).max()
    dpar = np.abs(tN[:, 1:] - t1[:, 1:]).max()
    rms = helpers.rms_per_channel(J0, J1)
    print(f'64 copies of one 1080p x 65-view image in one group ({N * n1} observations), closed={closed}: cost/64 rel {dcost:.2e}, '
          f'max|dparams| {dpar:.2e}, rms(J) {rms}')
    assert np.array_equal(np.isnan(J0), np.isnan(J1))
    if closed:
        assert dcost < 1e-5 and dpar < 1e-5 and rms.max() < 2e-5
    else:
        assert abs(tN[0, 0] / (N * t1[0, 0]) - 1) < 1e-5 and np.abs(tN[0, 1:] - t1[0, 1:]).max() < 1e-6   # iteration 1: same cost, same first step
        assert np.abs(tN[-1, 1:] - t1[-1, 1:]).max() > 1e-3                                                  # ... and then J lags (see above)
