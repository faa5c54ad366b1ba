"""Two ranks on TWO GPUs over RCCL (backend 'nccl') -- evidence the moment a node has two; on the one-GPU pool the same tests
run as a rehearsal over gloo (both ranks on cuda:0).

What the 8-GPU scaling run relies on, in small: bench.py's own rank launcher and the torchrun contract with one GPU per rank,
per-image mode (no collective: a rank's J is bit-identical to a 1-GPU restoration of the same image) and the shared-water
extension (one all-reduce of ten float64 sums per iteration over xGMI: both ranks' traces bitwise equal, and equal to the
one-process composition of the same two images)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import helpers

ROOT = Path(__file__).resolve().parent.parent
# SUCRE_TEST_NCCL2_ON_GLOO=1: the same tests on a ONE-GPU box with both ranks on cuda:0 over gloo -- a rehearsal of the test
# logic itself (launchers, digests, the one-process composition); the assertions about RCCL and two devices are then skipped.
# On a box with ONE GPU that rehearsal is what runs (so the file never skips): two GPUs -> RCCL, one GPU -> gloo.
REHEARSAL = os.environ.get('SUCRE_TEST_NCCL2_ON_GLOO') == '1' or torch.cuda.device_count() < 2
pytestmark = [pytest.mark.gpu]
BACKEND = 'gloo' if REHEARSAL else 'nccl'

SMALL = ['--width', '320', '--height', '240', '--neighbours', '8', '--num-iter', '6', '--no-cpu-baseline', '--solo-images', '1']


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SUCRE_DIST_BACKEND')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if REHEARSAL:
        env['SUCRE_DIST_BACKEND'] = 'gloo'
    return env


def _bench(extra, launcher=False):
    env = _clean_env()
    if launcher:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2']
    else:
        cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '2']
    out = subprocess.run(cmd + ['--steps', '2', '--warmup', '1'] + SMALL + extra, env=env, capture_output=True, text=True,
                         timeout=500, cwd=str(ROOT))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize('launcher', [False, True], ids=['own-launcher', 'torchrun'])
def test_per_image_mode_on_two_gpus_equals_one_gpu_bit_for_bit(launcher):
    rec = _bench(['--digest'], launcher)
    cfg = rec['config']
    assert rec['n_gpus'] == 2 and cfg['ranks_seen'] == 2 and cfg['dist_backend'] == BACKEND
    assert len(set(cfg['devices'])) == 2 and 'cuda:0' in cfg['devices'][0] and (REHEARSAL or 'cuda:1' in cfg['devices'][1])
    per_rank = cfg['ms_per_image_per_rank']
    assert len(per_rank['all']) == 2 and per_rank['min'] <= per_rank['max'] and per_rank['max'] <= cfg['ms_per_image'] * 1.001 + 1e-6
    # the same two images, one after the other, in THIS process on cuda:0 (rank r's scene is seed r, rendered on its device)
    import hashlib
    from sucre_amd import engine, synth
    for rank, want in enumerate(cfg['J_sha256_per_rank']):
        scene = synth.make_scene(320, 240, 8, seed=rank, device='cuda:0')
        views = engine.device_views_from_scene(scene, 'cuda:0')
        r = engine.Restoration(240, 320, len(views), device='cuda:0')
        r.match(views[scene.target], views, min_cover=1e-6)
        r.fit_init(views[scene.target])
        r.fit(6)
        torch.cuda.synchronize()
        assert hashlib.sha256(r.J().cpu().numpy().tobytes()).hexdigest() == want, f'rank {rank}: J differs from the 1-GPU run'


@pytest.mark.timeout(900)
@pytest.mark.parametrize('extra', [['--shared-water'], ['--shared-water', '--batch-images', '2', '--use-closed-form']],
                         ids=['shared-water', 'shared-water-group-closed'])
def test_bench_shared_water_on_two_gpus(extra):
    rec = _bench(extra)
    cfg = rec['config']
    assert cfg['ranks_seen'] == 2 and cfg['dist_backend'] == BACKEND and len(set(cfg['devices'])) == 2
    assert rec['value'] > 0 and rec['scaling'] == 'weak' and 'shared water' in cfg['workload']


@pytest.mark.timeout(900)
def test_shared_water_trace_over_rccl_equals_the_one_process_composition(golden, tmp_path):
    """tests/dist_worker.py, one rank per GPU, backend nccl: both ranks' traces bitwise equal; equal to the split-path composition
    of the same two images in one process (sums added by hand in rank order) to 1e-6 -- the float64 sum of two addends commutes,
    so they are expected to be the same bits -- and to the tied reference modules' golden trace."""
    from sucre_amd import engine
    T = int(golden['shared_trace'].shape[0])
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(_clean_env(), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(ROOT / 'tests' / 'dist_worker.py'), str(tmp_path), golden.name, str(T)], env=env))
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert str(r0['backend']) == BACKEND and int(r0['world']) == 2
    assert np.array_equal(r0['trace'], r1['trace']) and np.array_equal(r0['params'], r1['params'])
    # one process, two images, the sums added by hand
    sc = golden.scene
    views = engine.device_views_from_scene(sc, 'cuda:0')
    rs = []
    for t in (int(x) for x in golden['shared_targets']):
        x = engine.Restoration(sc.height, sc.width, len(views), device='cuda:0')
        x.match(views[t], views)
        x.fit_init(views[t])
        rs.append(x)
    trace = torch.zeros((T, 10), dtype=torch.float64, device='cuda:0')
    bes = [engine.HipWaterBackend(x, trace=trace if i == 0 else None) for i, x in enumerate(rs)]
    total = sum(be.n_obs() for be in bes)
    for be in bes:
        be.set_n_obs_total(total)
    for it in range(1, T + 1):
        sums = [be.grad(it).clone() for be in bes]
        tot = sums[0] + sums[1]
        for be in bes:
            be._sums.copy_(tot)
            be.step(it)
    torch.cuda.synchronize()
    one = trace.cpu().numpy()
    d = np.abs(r0['trace'] - one).max()
    print(f'two ranks over {BACKEND} vs one process: max |d trace| = {d:.2e} (bitwise: {np.array_equal(r0["trace"], one)})')
    assert d < 1e-6
    for r, x in ((r0, rs[0]), (r1, rs[1])):
        assert helpers.rms_per_channel(r['J'], x.J().cpu().numpy()).max() < 1e-6
    rt = golden['shared_trace']
    assert np.abs(r0['trace'][:, 1:] - rt[:, 1:]).max() < 1e-5 and np.abs(r0['trace'][:, 0] / rt[:, 0] - 1).max() < 1e-4
