#!/usr/bin/env python3
"""Wall-clock spans of the CLI's stages on every thread (monkeypatched timers, no profiler): who waits for whom in
tools/cli_survey_bench.py's survey?  usage (GPU box): python3 tools/cli_timeline.py [width height grid_x grid_y n]"""
import sys
import tempfile
import threading
import time
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from sucre_amd import engine, loader, sfm, sucre, synth  # noqa: E402

SPANS = []
T0 = [0.0]


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def timed(*a, **kw):
        t = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            SPANS.append((threading.current_thread().name, label, t - T0[0], time.perf_counter() - T0[0]))
    setattr(owner, name, timed)


def main():
    W, H, gx, gy, n = (int(a) for a in (sys.argv[1:6] + [1920, 1080, 8, 6, 48][len(sys.argv[1:6]):]))
    spacing = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1
    survey = synth.make_survey(W, H, gx, gy, seed=3, spacing=spacing, device='cuda')
    for owner, name in ((loader, '_imread_rgb_u8'), (loader, '_imread_depth_u16'), (sucre, '_restore_submit'),
                        (sucre, '_restore_enqueue_fit'), (sucre, '_restore_finish'), (sucre, '_write_outputs'),
                        (sucre.SUCRe, 'save_plots'), (sucre.SUCRe, '_plot_J_device'), (sucre.SUCRe, 'plot_J'),
                        (sucre.SUCRe, 'plot_reconstruction'), (torch, 'save'),
                        (sfm.Image, 'match_images'), (sfm.Image, 'overlapping_views'), (loader, 'prefetch_device_views'),
                        (loader.MatchesFile, 'check_integrity'), (loader.MatchesFile, 'prepare_matches'),
                        (loader.MatchesFile, 'load_matches'), (engine.Restoration, 'match'), (engine, 'acquire_restoration'),
                        (sucre, '_pull_results'), (sucre, '_adam_begin'), (sucre.SUCRe, '__init__'), (sfm.Image, 'device_view'),
                        (sfm.Image, 'depth_range'), (engine.Restoration, '__init__'), (engine.Restoration, 'fit_init'),
                        (engine.Restoration, 'fit'), (sfm, 'COLMAPModel'), (loader, 'prefetch_for_targets')):
        wrap(owner, name, f'{getattr(owner, "__name__", owner)}.{name}'.replace('sucre_amd.', ''))
    import os
    if os.environ.get('TIMELINE_NO_OUTPUTS'):   # experiment: what does the loop cost without the output stage?
        sucre._write_outputs = lambda job, keep, log=False: None
    if os.environ.get('TIMELINE_NO_PLOTS'):
        sucre.SUCRe.save_plots = lambda self, save_dir, iteration=None: None
    if os.environ.get('TIMELINE_NO_PNG'):
        sucre._save_png = lambda img, path: None
    if os.environ.get('TIMELINE_PNG_SLEEP'):
        sucre._save_png = lambda img, path: time.sleep(0.08)
    if os.environ.get('TIMELINE_PNG_BURN'):
        import zlib
        blob = os.urandom(1 << 20) * 6
        sucre._save_png = lambda img, path: zlib.compress(blob, 1)
    if os.environ.get('TIMELINE_PNG_NOWRITE'):
        import pathlib
        real = pathlib.Path.write_bytes
        pathlib.Path.write_bytes = lambda self, data: len(data) if str(self).endswith('.png') else real(self, data)
    if os.environ.get('TIMELINE_PNG_PROCS'):
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        import numpy as np
        from sucre_amd import _pixelio as _png
        pool = ProcessPoolExecutor(int(os.environ['TIMELINE_PNG_PROCS']), mp_context=mp.get_context('spawn'))
        list(pool.map(abs, range(64)))   # start the workers now
        sucre._save_png = lambda img, path: pool.submit(_png.write_rgb, str(path), np.asarray(img), 1).result()
    if os.environ.get('TIMELINE_NO_RECON'):
        from PIL import Image as _PI
        sucre.SUCRe.plot_reconstruction = lambda self: _PI.new('RGB', (8, 8))
    if os.environ.get('TIMELINE_NO_PLOTJ'):
        from PIL import Image as _PI
        sucre.SUCRe.plot_J = lambda self: _PI.new('RGB', (8, 8))
    if os.environ.get('TIMELINE_SWITCH'):
        sys.setswitchinterval(float(os.environ['TIMELINE_SWITCH']))
    from sucre_amd import _lib
    lib = _lib.load()
    for name in ('sucre_match_views', 'sucre_finalize_matches_fmt', 'sucre_check_store', 'sucre_fit_init', 'sucre_fit_run',
                 'sucre_export_J', 'sucre_select_ranks'):
        wrap(lib, name, 'lib.' + name)
    wrap(torch.Tensor, 'to', 'Tensor.to'); wrap(torch.Tensor, 'cpu', 'Tensor.cpu'); wrap(torch.Tensor, 'item', 'Tensor.item')
    wrap(torch.cuda.Stream, 'synchronize', 'Stream.synchronize'); wrap(torch.cuda.Stream, 'wait_stream', 'Stream.wait_stream')
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp)
        synth.write_to_disk(survey, root)
        first = 1 if n >= gx * gy else gx * (gy // 2) + 1
        argv = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(root / 'out'), '--image-ids', str(first), str(first + n)]
        def throttled():
            try:
                return dict(line.split() for line in open('/sys/fs/cgroup/cpu.stat').read().splitlines())
            except OSError:
                return {}
        ALLOCS = []
        real_finish = sucre._restore_finish

        def counting_finish(*a, **kw):
            st = torch.cuda.memory_stats()
            ALLOCS.append((st.get('num_device_alloc', 0), st.get('num_device_free', 0), st.get('reserved_bytes.all.current', 0) >> 20))
            return real_finish(*a, **kw)
        sucre._restore_finish = counting_finish
        before = throttled()
        T0[0] = time.perf_counter()
        sucre.main(argv)
        dt = time.perf_counter() - T0[0]
        after = throttled()
        if after:
            print('cgroup: cpu used %.2f s, throttled %d periods / %.3f s' % (
                (int(after['usage_usec']) - int(before['usage_usec'])) * 1e-6,
                int(after['nr_throttled']) - int(before['nr_throttled']),
                (int(after['throttled_usec']) - int(before['throttled_usec'])) * 1e-6))
    print('device allocations (hipMalloc, hipFree, reserved MiB) at finish 1, 10, 20, 30, last:',
          [ALLOCS[i] for i in (0, 9, 19, 29, len(ALLOCS) - 1) if i < len(ALLOCS)])
    print(f'CLI: {n} images in {dt:.2f}s = {dt / n * 1e3:.0f} ms/image')
    tot = defaultdict(lambda: [0, 0.0])
    for th, label, a, b in SPANS:
        kind = 'main' if th == 'MainThread' else th.rsplit('_', 1)[0].rsplit('-', 1)[0] if 'sucre' in th else th.split('_')[0]
        k = (kind, label)
        tot[k][0] += 1; tot[k][1] += b - a
    print('thread kind / span: calls, total s, ms per call')
    for (kind, label), (c, s) in sorted(tot.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
        print(f'  {kind:18s} {label:40s} {c:5d} {s:8.3f} {s / c * 1e3:8.1f}')
    print('main-thread timeline (s): submit[start-end] enqueue[end] finish[start-end]')
    rows = [(a, b, label) for th, label, a, b in SPANS if th == 'MainThread' and label in
            ('sucre._restore_submit', 'sucre._restore_enqueue_fit', 'sucre._restore_finish')]
    for a, b, label in sorted(rows)[:40]:
        print(f'  {a:7.3f} {b:7.3f} {(b - a) * 1e3:7.1f} ms  {label}')
    third = sorted((a, b) for th, label, a, b in SPANS if th == 'MainThread' and label == 'sucre._restore_submit')
    lo, hi = third[20][0], third[22][0]
    print('main thread, images 21-22:')
    for a, b, label in sorted((a, b, label) for th, label, a, b in SPANS if th == 'MainThread' and lo <= a < hi):
        if b - a > 0.0002:
            print(f'  {a:7.4f} {b:7.4f} {(b - a) * 1e3:7.2f} ms  {label}')
    print('everything the main thread did before the second image was submitted:')
    second = sorted(a for th, label, a, b in SPANS if th == 'MainThread' and label == 'sucre._restore_submit')[1]
    for a, b, label in sorted((a, b, label) for th, label, a, b in SPANS if th == 'MainThread' and a < second):
        print(f'  {a:7.3f} {b:7.3f} {(b - a) * 1e3:7.1f} ms  {label}')
    fin = sorted((a, b) for th, label, a, b in SPANS if th == 'MainThread' and label == 'sucre._restore_finish')
    sub = sorted((a, b) for th, label, a, b in SPANS if th == 'MainThread' and label == 'sucre._restore_submit')
    print(f'first submit {sub[0][0]:.3f}-{sub[0][1]:.3f}; finishes: first {fin[0][1]:.3f}, 10th {fin[9][1]:.3f}, last {fin[-1][1]:.3f} '
          f'-> {(fin[-1][1] - fin[9][1]) / (len(fin) - 10) * 1e3:.1f} ms per image in the loop; total {dt:.3f}')
    w = sorted((a, b) for th, label, a, b in SPANS if label == 'sucre._write_outputs')
    print('write_outputs: first start %.3f, last end %.3f, mean %.0f ms' % (w[0][0], max(b for _, b in w), sum(b - a for a, b in w) / len(w) * 1e3))
    d = sorted((a, b) for th, label, a, b in SPANS if label == 'loader._imread_rgb_u8')
    print('rgb decode: first start %.3f, last end %.3f, mean %.0f ms' % (d[0][0], max(b for _, b in d), sum(b - a for a, b in d) / len(d) * 1e3))


if __name__ == '__main__':
    main()
