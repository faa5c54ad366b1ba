"""EIGHT ranks, rehearsed: ``bench.py --gpus 8`` as the driver's scaling run starts it, at toy size.

On a node with eight GPUs this is RCCL over xGMI, one device per rank; on the one-GPU pool the eight ranks share cuda:0 and
talk over gloo (SUCRE_DIST_BACKEND) -- what is rehearsed is everything that does not depend on the transport: the rank ->
device map, eight watchdogs, the eight-way gathers of the JSON line, per-image sharding (no collective: every rank's J is the
bits of restoring that rank's image alone) and the shared-water extension at eight ranks (one all-reduce of ten float64 sums
per iteration: eight identical trajectories, equal to the one-process composition of all 32 images).
North star: "512-image scene sharded across 8 x MI355X with a single RCCL all-reduce for the shared global water parameters"."""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
pytestmark = [pytest.mark.gpu]
REHEARSAL = torch.cuda.device_count() < 8
BACKEND = 'gloo' if REHEARSAL else 'nccl'
W, H, NB, T = 320, 240, 8, 6
SMALL = ['--width', str(W), '--height', str(H), '--neighbours', str(NB), '--num-iter', str(T), '--no-cpu-baseline', '--solo-images', '1',
         '--steps', '2', '--warmup', '1', '--timeout-s', '400']


def _bench8(extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SUCRE_DIST_BACKEND')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if REHEARSAL:
        env['SUCRE_DIST_BACKEND'] = 'gloo'
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '8'] + SMALL + extra, env=env, capture_output=True, text=True,
                         timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _check_eight(rec):
    cfg = rec['config']
    assert rec['n_gpus'] == 8 and cfg['ranks_seen'] == 8 and cfg['dist_backend'] == BACKEND and rec['scaling'] == 'weak'
    assert len(cfg['devices']) == 8 and [d.split(':')[0] for d in cfg['devices']] == [f'rank {r}' for r in range(8)]
    if not REHEARSAL:
        assert all(f'(cuda:{r})' in cfg['devices'][r] for r in range(8)), cfg['devices']
    per_rank = cfg['ms_per_image_per_rank']
    assert len(per_rank['all']) == 8 and per_rank['min'] <= per_rank['max'] <= cfg['ms_per_image'] * 1.001 + 1e-6
    return cfg


@pytest.mark.timeout(900)
def test_per_image_mode_on_eight_ranks_equals_one_rank_bit_for_bit():
    cfg = _check_eight(_bench8(['--digest']))
    from sucre_amd import engine, synth
    assert len(cfg['J_sha256_per_rank']) == 8
    for rank, want in enumerate(cfg['J_sha256_per_rank']):   # rank r's scene is seed r
        scene = synth.make_scene(W, H, NB, seed=rank, device='cuda:0')
        views = engine.device_views_from_scene(scene, 'cuda:0')
        r = engine.Restoration(H, W, len(views), device='cuda:0')
        r.match(views[scene.target], views, min_cover=1e-6)
        r.fit_init(views[scene.target])
        r.fit(T)
        torch.cuda.synchronize()
        assert hashlib.sha256(r.J().cpu().numpy().tobytes()).hexdigest() == want, f'rank {rank}: J differs from the 1-rank run'


@pytest.mark.timeout(900)
def test_shared_water_on_eight_ranks_equals_the_one_process_composition():
    """BASELINE config 4 in small: 8 ranks x 4 images of a rank's own survey, all 32 sharing B, beta, gamma -- one group launch
    and one all-reduce per iteration.  Eight bitwise-equal trajectories; equal to ONE process fitting the 32 images in one
    group (sums of 32 images in one launch instead of 8 x 4 + an all-reduce: another association of the float64 adds) to 1e-6."""
    import importlib
    cfg = _check_eight(_bench8(['--shared-water', '--batch-images', '4', '--digest']))
    assert '32-image scene, 4 per rank' in cfg['workload'] and 'shared water' in cfg['workload']
    digests = cfg['shared_water_trace_sha256_per_rank']
    assert len(digests) == 8 and len(set(digests)) == 1, digests
    trace8 = np.asarray(cfg['shared_water_trace_rank0'], np.float64)
    assert trace8.shape == (T, 10) and hashlib.sha256(trace8.tobytes()).hexdigest() == digests[0]
    from sucre_amd import dist as sdist
    from sucre_amd import engine, synth
    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module('bench')
    rs, keep = [], []
    for rank in range(8):   # what every rank built: the same survey generator, seed = rank
        survey, all_views, jobs, centre = bench.survey_jobs(synth, engine, W, H, NB, 4, rank, torch.device('cuda:0'))
        keep.append(all_views)
        for tgt, views in jobs:
            x = engine.Restoration(H, W, len(views), device='cuda:0')
            x.match(tgt, views, min_cover=1e-6)
            x.fit_init(tgt)
            rs.append(x)
    assert len(rs) == 32
    trace = torch.zeros((T, 10), dtype=torch.float64, device='cuda:0')
    sdist.fit_shared_water(engine.HipWaterGroup(rs, trace=trace), T)
    torch.cuda.synchronize()
    one = trace.cpu().numpy()
    dpar, dcost = np.abs(trace8[:, 1:] - one[:, 1:]).max(), np.abs(trace8[:, 0] / one[:, 0] - 1).max()
    print(f'eight ranks over {BACKEND} vs one process, 32 images: max|dparams| = {dpar:.2e}, max rel dcost = {dcost:.2e}')
    assert dpar < 1e-6 and dcost < 1e-6


def test_a_512_image_scene_shards_into_64_contiguous_images_per_rank():
    from sucre_amd import dist as sdist
    ids = list(range(512))
    parts = [sdist.shard_images(ids, r, 8) for r in range(8)]
    assert all(p == list(range(64 * r, 64 * r + 64)) for r, p in enumerate(parts))
    assert sorted(sum(parts, [])) == ids
