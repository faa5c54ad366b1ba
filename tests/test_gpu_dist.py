"""N > 1 on the GPU box: two real processes share the one GPU (gloo carries the all-reduce, the kernels are the HIP
ones), so the multi-process path -- HipWaterBackend under dist.fit_shared_water, and bench.py's own rank launcher --
is exercised before the driver runs it on eight GPUs over RCCL."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
def test_two_processes_shared_water_hip_backend(golden, tmp_path):
    """Both ranks' traces are bitwise equal (identical step everywhere) and match the golden produced by two
    reference SUCRe modules with tied B, beta, gamma (tests/golden/ref_harness.py::reference_shared_water)."""
    T = int(golden['shared_trace'].shape[0])
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2',
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SUCRE_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, str(ROOT / 'tests' / 'dist_worker.py'), str(tmp_path), golden.name, str(T)],
                                      env=env))
    assert [p.wait(timeout=500) for p in procs] == [0, 0]
    r0, r1 = np.load(tmp_path / 'rank0.npz'), np.load(tmp_path / 'rank1.npz')
    assert str(r0['backend']) == 'gloo' and int(r0['world']) == 2
    assert np.array_equal(r0['trace'], r1['trace'])
    assert np.array_equal(r0['params'], r1['params'])
    rt = golden['shared_trace']
    assert np.abs(r0['trace'][:, 1:] - rt[:, 1:]).max() < 1e-5
    assert np.abs(r0['trace'][:, 0] / rt[:, 0] - 1).max() < 1e-4
    for r, key in ((r0, 'shared_J0'), (r1, 'shared_J1')):
        assert np.array_equal(np.isnan(r['J']), np.isnan(golden[key]))
        assert helpers.rms_per_channel(r['J'], golden[key]).max() < 1e-4


@pytest.mark.timeout(600)
@pytest.mark.parametrize('extra', [['--digest'], ['--shared-water'], ['--shared-water', '--batch-images', '2', '--use-closed-form']],
                         ids=['per-image', 'shared-water', 'shared-water-group-closed'])
def test_bench_launches_its_own_ranks(extra):
    """`python bench.py --gpus 2` with no launcher environment starts two ranks itself and rank 0 prints one JSON
    line that saw both of them."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1', '--width', '320',
           '--height', '240', '--neighbours', '8', '--num-iter', '6', '--no-cpu-baseline', '--solo-images', '1'] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['ranks_seen'] == 2 and len(rec['config']['devices']) == 2
    assert rec['value'] > 0 and rec['scaling'] == 'weak'
    per_rank = rec['config']['ms_per_image_per_rank']
    assert len(per_rank['all']) == 2 and per_rank['min'] <= per_rank['max']   # a straggler would show here
    if '--digest' in extra:   # every rank restored its own image (seed = rank): two different J, each reported
        d = rec['config']['J_sha256_per_rank']
        assert len(d) == 2 and d[0] != d[1] and all(len(x) == 64 for x in d)


@pytest.mark.timeout(600)
def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- with two ranks on this box's one GPU (gloo instead of RCCL, which refuses
    two ranks on one device): rank 0 prints the one JSON line, and it saw both ranks."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['SUCRE_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--width', '320',
           '--height', '240', '--neighbours', '8', '--num-iter', '6', '--no-cpu-baseline', '--solo-images', '1']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500, cwd=str(ROOT))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 2 and rec['warmup'] == 1 and rec['config']['ranks_seen'] == 2
    assert rec['value'] > 0 and rec['scaling'] == 'weak' and rec['config']['dist_backend'] == 'gloo'


@pytest.mark.timeout(600)
def test_bench_json_line_keeps_the_contract():
    """`python bench.py` at N = 1: one JSON line with every key of the driver's contract, the roofline block and (on a
    tiny sample) the CPU baseline leg."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--steps', '2', '--warmup', '1', '--width', '320', '--height', '240',
           '--neighbours', '8', '--num-iter', '6', '--solo-images', '1', '--cpu-views', '3', '--cpu-iters', '2']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in rec, key
    assert rec['n_gpus'] == 1 and rec['steps'] == 2 and rec['warmup'] == 1 and rec['higher_is_better'] is True
    assert rec['unit'] == 'Mpix/s' and rec['dtype'] == 'f32' and rec['data'] == 'synthetic' and rec['vs_baseline'] is None
    assert 'workload' in rec['config'] and 'model' not in rec['config']
    roof = rec['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in roof, key
    assert roof['bound'] == 'hbm' and roof['peak'] == 8000.0 and abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-12
    cpu = rec['cpu_baseline']
    assert cpu['kind'] == 'port' and cpu['cores'] >= 1 and cpu['value'] > 0 and 'sample' in cpu
    assert abs(rec['value'] - 2 * 320 * 240 / 1e6 / (rec['ms_per_step'] * 2e-3)) < 1e-6 * rec['value']   # value = work / time


@pytest.mark.timeout(600)
def test_results_do_not_depend_on_who_else_uses_the_gpu(golden):
    """Two processes fitting at the same time (workgroups of a launch no longer start together, latencies stretch):
    every repetition in both processes, on the fused per-image path, on the single-launch group path and through the batch
    launches (round 5), must be the same bits as a run alone on the GPU.  (Caught in round 2: plan items loaded by a hand-issued s_load were copied
    before they had landed, and group sums could be read before all four waves' stores had drained.)"""
    T, reps = 40, 12
    cmd = [sys.executable, str(ROOT / 'tests' / 'concurrency_worker.py'), golden.name, str(reps), str(T)]

    def digests(out):
        line = [ln for ln in out.splitlines() if ln.startswith('DIGESTS ')]
        assert len(line) == 1, out[-2000:]
        return line[0].split()[1:]
    alone = subprocess.run(cmd[:3] + ['3', str(T)], capture_output=True, text=True, timeout=300)
    assert alone.returncode == 0, alone.stderr[-2000:]
    ref = digests(alone.stdout)
    assert ref[0] == ref[1] == ref[2]            # group of one == batch of one == fused fit, bit for bit
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=500) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        assert digests(so) == [ref[0]] * reps


@pytest.mark.timeout(900)
def test_images_in_flight_equal_one_by_one_under_load():
    """Different images of a survey restored on two and three HIP streams at once (engine.in_flight_slot's use), with a
    second thread keeping the GPU busy with unrelated work, against the same images restored one by one: same bits,
    every time.  Full size on purpose: the races this guards against (an LDS-DMA overwriting a ring slot whose reads
    had not returned; hand-offs read before they had drained) only showed under load."""
    import threading
    import torch
    from sucre_amd import engine, synth
    W, H = 1920, 1080
    survey = synth.make_survey(W, H, 6, 4, seed=3, device='cuda')
    views = engine.device_views_from_scene(survey, 'cuda')
    jobs = [(views[t], [views[q] for q in survey.neighbours(t, 12 + (t % 5))]) for t in (7, 8, 9, 10, 13, 14)]
    cap = max(len(v) for _, v in jobs)
    T = 120
    ref = []
    r = engine.Restoration(H, W, cap)
    for tgt, vs in jobs:
        r.match(tgt, vs); r.fit_init(tgt); t = r.fit(T); torch.cuda.synchronize()
        ref.append((r.J().clone(), t.clone()))
    del r
    stop = []

    def noise():
        g = torch.Generator(device='cuda'); g.manual_seed(1)
        while not stop:
            x = torch.rand((1_000_000, 3), device='cuda', generator=g)
            y = x[x[:, 0] > 0.5]
            float(torch.sort(y[:, 1]).values.exp().sum())
    th = threading.Thread(target=noise)
    th.start()
    try:
        for ns in (2, 3):
            restos = [engine.Restoration(H, W, cap) for _ in range(ns)]
            streams = [torch.cuda.Stream() for _ in range(ns)]
            outs = []
            for rep in range(2):
                for i, (tgt, vs) in enumerate(jobs):
                    with torch.cuda.stream(streams[i % ns]):
                        rr = restos[i % ns]
                        rr.match(tgt, vs); rr.fit_init(tgt)
                        outs.append((i, rr.fit(T), rr.J()))
            torch.cuda.synchronize()
            for i, t, J in outs:
                assert torch.equal(t, ref[i][1]), (ns, i)
                assert torch.equal(torch.nan_to_num(J), torch.nan_to_num(ref[i][0])), (ns, i)
            del restos
    finally:
        stop.append(1)
        th.join()


@pytest.mark.timeout(900)
def test_cli_sharded_over_two_processes_equals_one_process(tmp_path):
    """`python -m sucre_amd.sucre --image-ids ...` under WORLD_SIZE=2 (the torchrun contract; both ranks on this box's one
    GPU): every rank restores its shard of the images (dist.shard_images, no collective) and the union of the two
    ranks' output files equals the one-process run byte for byte (.pt tensors bit for bit, PNGs as files)."""
    import torch
    from sucre_amd import synth
    scene_dir = tmp_path / 'scene'
    survey = synth.make_survey(160, 120, 3, 2, seed=4)
    synth.write_to_disk(survey, scene_dir)
    base = [sys.executable, '-m', 'sucre_amd.sucre', '--image-dir', str(scene_dir / 'images'), '--depth-dir', str(scene_dir / 'depth'),
            '--model-dir', str(scene_dir / 'model'), '--image-ids', '1', '7', '--num-iter', '12']
    clean = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    clean['PYTHONPATH'] = str(ROOT) + os.pathsep + clean.get('PYTHONPATH', '')
    one = subprocess.run(base + ['--output-dir', str(tmp_path / 'one')], env=clean, capture_output=True, text=True, timeout=400, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    procs = []
    for rank in range(2):
        env = dict(clean, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2')
        procs.append(subprocess.Popen(base + ['--output-dir', str(tmp_path / 'two')], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=400) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    # every rank restored its own three images and nobody else's
    restored = [sorted(ln.split()[1].rstrip('.') for ln in so.splitlines() if ln.startswith('Restore ')) for so, _ in outs]
    names = sorted(v.name for v in survey.views)
    assert restored[0] == names[:3] and restored[1] == names[3:], restored
    files_one = sorted(f.name for f in (tmp_path / 'one').iterdir())
    files_two = sorted(f.name for f in (tmp_path / 'two').iterdir())
    assert files_one == files_two and len([f for f in files_one if f.endswith('.pt')]) == 6
    for name in files_one:
        a, b = tmp_path / 'one' / name, tmp_path / 'two' / name
        if name.endswith('.pt'):
            sa, sb = torch.load(a), torch.load(b)
            assert set(sa) == set(sb)
            for k in sa:
                assert torch.equal(torch.nan_to_num(sa[k]), torch.nan_to_num(sb[k])) and torch.equal(torch.isnan(sa[k]), torch.isnan(sb[k])), (name, k)
        else:
            assert a.read_bytes() == b.read_bytes(), name
