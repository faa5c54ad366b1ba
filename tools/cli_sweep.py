#!/usr/bin/env python3
"""Self-consistency sweep of the command line on the GPU box: random synthetic surveys on disk (sizes, grids), random
reference flags (--use-closed-form, --light-model, --image-scale, --num-iter, --min-cover, --save-interval,
--keep-matches), each restored twice -- once with the engine's default knobs and once with a random other setting of the
knobs that must NOT change a single output byte (images in flight, images per fit launch, packed views, overlap cull, decode / PNG threads and
processes, one rank vs two ranks).  Every file of the two output directories must be identical; with --keep-matches a third
run over the kept matches files (the import path instead of matching) must reproduce the outputs too.
    python3 tools/cli_sweep.py [n_cases] [seed0]"""
import filecmp
import os
import shutil
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def run(cmd, env):
    out = subprocess.run(cmd, env=env, capture_output=True, cwd=ROOT)
    assert out.returncode == 0, (cmd, out.stdout[-2000:].decode(errors='replace'), out.stderr[-3000:].decode(errors='replace'))
    return out


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.default_rng(seed0)
    t0 = time.time()
    base_env = {k: v for k, v in os.environ.items() if not k.startswith('SUCRE_')}
    base_env['PYTHONPATH'] = str(ROOT)
    for case in range(n_cases):
        tmp = Path(tempfile.mkdtemp(prefix='cli_sweep_'))
        try:
            W, H = int(rng.integers(48, 260)), int(rng.integers(40, 200))
            nx, ny = int(rng.integers(2, 5)), int(rng.integers(2, 4))
            run([sys.executable, '-m', 'sucre_amd.synth', '--out', str(tmp / 'scene'), '--width', str(W), '--height', str(H),
                 '--grid', str(nx), str(ny), '--seed', str(seed0 + case), '--device', 'cpu'], base_env)
            scene = tmp / 'scene'
            dirs = {p.name: p for p in scene.iterdir() if p.is_dir()}
            image_dir = next(p for n, p in dirs.items() if 'image' in n)
            depth_dir = next(p for n, p in dirs.items() if 'depth' in n)
            model_dir = next(p for n, p in dirs.items() if n not in (image_dir.name, depth_dir.name))
            flags = ['--num-iter', str(int(rng.choice([3, 15, 40])))]
            if rng.random() < 0.5:
                flags.append('--use-closed-form')
            if rng.random() < 0.3:
                flags.append('--light-model')
            if rng.random() < 0.3:
                flags += ['--image-scale', str(float(rng.choice([0.5, 0.75, 1.5])))]
            if rng.random() < 0.3:
                flags += ['--min-cover', str(float(rng.choice([0.0, 0.05, 0.3])))]
            if rng.random() < 0.3:
                flags += ['--save-interval', str(int(rng.choice([2, 7])))]
            if rng.random() < 0.3:
                flags.append('--keep-matches')
            common = ['--image-dir', str(image_dir), '--depth-dir', str(depth_dir), '--model-dir', str(model_dir),
                      '--image-ids', '1', str(nx * ny + 1), '--device', 'cuda'] + flags
            knobs = {}
            if rng.random() < 0.6:
                knobs['SUCRE_IMAGES_IN_FLIGHT'] = str(int(rng.choice([1, 3])))
            if rng.random() < 0.4:
                knobs['SUCRE_PACKED_VIEWS'] = '0'
            if rng.random() < 0.4:
                knobs['SUCRE_CULL_VIEWS'] = '0'
            if rng.random() < 0.4:
                knobs['SUCRE_IO_PROCESSES'] = str(int(rng.choice([0, 2])))
            if rng.random() < 0.3:
                knobs['SUCRE_DECODE_THREADS'] = '1'; knobs['SUCRE_WRITER_THREADS'] = '1'
            if rng.random() < 0.3:
                knobs['SUCRE_DECODE_IN_WORKERS'] = '0'
            two_ranks = rng.random() < 0.3
            if rng.random() < 0.3:
                knobs['SUCRE_HOST_TORCH_THREADS'] = '0'
            if rng.random() < 0.6:   # (round 5) the default is auto: images of these sizes fit 8 per launch
                knobs['SUCRE_FIT_BATCH'] = str(int(rng.choice([1, 2, 5])))
            run([sys.executable, '-m', 'sucre_amd.sucre', '--output-dir', str(tmp / 'a')] + common, base_env)
            env_b = dict(base_env, **knobs)
            if two_ranks:
                run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                     '--master-port', str(29600 + case % 200), '-m', 'sucre_amd.sucre', '--output-dir', str(tmp / 'b')] + common, env_b)
            else:
                run([sys.executable, '-m', 'sucre_amd.sucre', '--output-dir', str(tmp / 'b')] + common, env_b)
            fa = sorted(p.relative_to(tmp / 'a') for p in (tmp / 'a').rglob('*') if p.is_file())
            fb = sorted(p.relative_to(tmp / 'b') for p in (tmp / 'b').rglob('*') if p.is_file())
            assert fa == fb and fa, (case, 'file lists differ', fa[:5], fb[:5], flags, knobs, two_ranks)
            diff = [str(f) for f in fa if f.suffix not in ('.h5', '.npz') and not filecmp.cmp(tmp / 'a' / f, tmp / 'b' / f, shallow=False)]
            assert not diff, (case, 'files differ', diff[:6], flags, knobs, two_ranks, (W, H, nx, ny))
            reused = False
            if '--keep-matches' in flags:   # a second run over its own kept matches files (the import path) gives the same outputs
                # (to the tolerance of tests/test_gpu_api.py::test_cli_kept_matches_are_reused_in_every_mode: the second run
                # rebuilds the camera points on the host and may see the views in another order)
                import torch
                shutil.copytree(tmp / 'a', tmp / 'c')
                run([sys.executable, '-m', 'sucre_amd.sucre', '--output-dir', str(tmp / 'c')] + common, env_b)
                light = '--light-model' in flags
                for f in fa:
                    if f.suffix != '.pt':
                        continue
                    a, c = torch.load(tmp / 'a' / f), torch.load(tmp / 'c' / f)
                    assert set(a) == set(c), (case, f)
                    Ja, Jc = a['J'].numpy(), c['J'].numpy()
                    assert np.array_equal(np.isnan(Ja), np.isnan(Jc)), (case, f, 'NaN mask after reuse')
                    ok = np.isfinite(Ja).all(axis=-1) & np.isfinite(Jc).all(axis=-1)
                    if ok.any():
                        d = Ja[ok].astype(np.float64) - Jc[ok].astype(np.float64)
                        scale = max(1.0, float(np.abs(Ja[ok]).max()))
                        rms = float(np.sqrt((d * d).mean(axis=0)).max()) / scale
                        assert rms < (1e-4 if light else 1e-5), (case, f, 'J after reuse', rms, flags, knobs)
                reused = True
            print(f'case {case}: reused={reused} {W}x{H} grid {nx}x{ny} {" ".join(flags)} | {knobs} two_ranks={two_ranks} -> {len(fa)} files identical, '
                  f'{time.time() - t0:.0f}s', flush=True)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    print('cli sweep ok', n_cases, 'cases')


if __name__ == '__main__':
    main()
