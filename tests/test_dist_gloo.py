"""N>1 path on CPU: two -- and eight -- gloo ranks drive ``dist.fit_shared_water`` with an oracle-backed backend.

The data-path collective is the one all-reduce of ten float64 sums per iteration; per-image mode has none
(``shard_images`` only).  Checks: both ranks end with identical water parameters, and the 2-rank result equals
the single-process composition of the same two images (rank-count invariance)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from oracle import oracle
from sucre_amd import dist as sdist
from sucre_amd import synth

T = 12


class OracleBackend(sdist.WaterBackend):
    """WaterBackend over the CPU oracle (tests only)."""

    def __init__(self, images):
        self.images = images                      # list of oracle.SharedWaterImage handled by this rank
        self.pstate = np.zeros(27, np.float32); self.pstate[:9] = 0.1
        self.sums = torch.zeros(10, dtype=torch.float64)
        self.total = None

    def n_obs(self): return sum(im.n_obs for im in self.images)
    def set_n_obs_total(self, n): self.total = int(n)

    def grad(self, step):
        acc = np.zeros(10)
        for im in self.images:
            acc += im.grad(self.pstate[:9], step, self.total)
        self.sums.copy_(torch.from_numpy(acc))
        return self.sums

    def step(self, step):
        oracle.shared_step(self.pstate, self.sums.numpy(), step, self.total)


def make_image(seed):
    scene = synth.make_scene(48, 32, 3, seed=seed)
    _, samples = helpers.oracle_scene_samples(scene)
    tgt = scene.views[scene.target]
    return oracle.SharedWaterImage(32, 48, samples, oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy()))


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r, lr, w = sdist.init_process_group(backend='gloo')
    assert (r, w) == (rank, world)
    mine = sdist.shard_images([10, 11], rank, world)          # one image (seed) per rank
    be = OracleBackend([make_image(s) for s in mine])
    sdist.fit_shared_water(be, T)
    np.savez(Path(out_dir) / f'rank{rank}.npz', pstate=be.pstate, J=be.images[0].J, total=be.total)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_shared_water_two_ranks_equals_one_process(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert np.array_equal(r0['pstate'], r1['pstate'])        # identical step on every rank
    assert int(r0['total']) == int(r1['total'])
    # single process holding both images, no process group: same objective, same trajectory
    be = OracleBackend([make_image(10), make_image(11)])
    sdist.fit_shared_water(be, T)
    assert int(r0['total']) == be.total
    assert np.abs(be.pstate[:9] - r0['pstate'][:9]).max() < 1e-6
    assert helpers.rms_per_channel(be.images[0].J, r0['J']).max() < 1e-6
    assert helpers.rms_per_channel(be.images[1].J, r1['J']).max() < 1e-6
    assert np.abs(be.pstate[:9] - 0.1).max() > 1e-2          # the parameters actually moved


def _worker8(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    r, lr, w = sdist.init_process_group(backend='gloo')
    assert (r, w) == (rank, world)
    mine = sdist.shard_images(list(range(20, 20 + 2 * world)), rank, world)     # two images (seeds) per rank, contiguous
    assert mine == [20 + 2 * rank, 21 + 2 * rank]
    be = OracleBackend([make_image(s) for s in mine])
    sdist.fit_shared_water(be, 6)
    np.savez(Path(out_dir) / f'rank{rank}.npz', pstate=be.pstate, J0=be.images[0].J, J1=be.images[1].J, total=be.total)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_shared_water_eight_ranks_equals_one_process(tmp_path):
    """The north star's rank count on the CPU tier: eight gloo ranks, two images each, one all-reduce of ten float64 sums per
    iteration -- eight identical parameter sets, equal (1e-6) to ONE process holding all sixteen images, every rank's J equal to
    that process's J of the same image."""
    world = 8
    mp.spawn(_worker8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rs = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    for r in rs[1:]:
        assert np.array_equal(r['pstate'], rs[0]['pstate']) and int(r['total']) == int(rs[0]['total'])
    be = OracleBackend([make_image(s) for s in range(20, 20 + 2 * world)])
    sdist.fit_shared_water(be, 6)
    assert int(rs[0]['total']) == be.total
    assert np.abs(be.pstate[:9] - rs[0]['pstate'][:9]).max() < 1e-6
    for rank, r in enumerate(rs):
        for j, key in enumerate(('J0', 'J1')):
            assert helpers.rms_per_channel(be.images[2 * rank + j].J, r[key]).max() < 1e-6, (rank, key)


def test_shared_water_with_one_image_reduces_to_reference_fit():
    """world = 1, one image: the split iteration is the reference algorithm (golden-pinned oracle.fit)."""
    scene = synth.make_scene(48, 32, 3, seed=10)
    _, samples = helpers.oracle_scene_samples(scene)
    tgt = scene.views[scene.target]
    J0 = oracle.init_J(tgt.rgb_u8.numpy(), tgt.depth_f32().numpy())
    be = OracleBackend([oracle.SharedWaterImage(32, 48, samples, J0)])
    sdist.fit_shared_water(be, T)
    J, params, trace = oracle.fit(32, 48, samples, J0, num_iter=T)
    assert np.abs(be.pstate[:9] - params).max() < 1e-6
    assert helpers.rms_per_channel(be.images[0].J, J).max() < 1e-6


def test_bench_launcher_reports_a_failed_rank_and_stops_the_others(tmp_path):
    """`python bench.py --gpus 2` without a launcher starts its own ranks; a rank that fails (here: no GPU in this
    container, so every rank stops at its 'needs a GPU' assertion after the gloo rendezvous) must make the launcher exit
    non-zero promptly -- also when the failure is a signal (negative return code) -- instead of reporting the best rank's
    code or waiting in a collective."""
    import os
    import subprocess
    import sys
    import time
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip('needs a machine without a GPU (on a GPU box tests/test_gpu_dist.py runs the launcher for real)')
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.time()
    out = subprocess.run([sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--timeout-s', '120'],
                         env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 1 and time.time() - t0 < 100
    assert 'stopping the other ranks' in out.stderr and 'needs a GPU' in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith('{')]     # no JSON line from a failed job
    # per-rank progress lines on stderr name the stage a rank is in
    assert 'bench.py rank 0: rendezvous' in out.stderr and 'bench.py rank 1: rendezvous' in out.stderr


def test_bench_watchdog_names_the_stage_and_exits(tmp_path):
    """Every rank gives itself --timeout-s: a stuck rank says where it is and exits with code 3 (no re-exec)."""
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    code = ('import sys, time; sys.argv = ["bench.py"]; sys.path.insert(0, %r); import bench; '
            'bench.STAGE[0] = "barrier before the timed region"; bench.watchdog(5, 0.5); time.sleep(30)') % str(root)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3
    assert "rank 5: no result after 0 s, stuck in stage 'barrier before the timed region'" in out.stderr
