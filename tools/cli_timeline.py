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
    W, H, gx, gy, n = (int(a) for a in (sys.argv[1:6] + [1920, 1080, 8, 6, 48][len(sys.argv) - 1:]))
    survey = synth.make_survey(W, H, gx, gy, seed=3, device='cuda')
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
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp)
        synth.write_to_disk(survey, root)
        first = 1 if n >= gx * gy else gx * (gy // 2) + 1
        argv = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(root / 'out'), '--image-ids', str(first), str(first + n)]
        T0[0] = time.perf_counter()
        sucre.main(argv)
        dt = time.perf_counter() - T0[0]
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
    print('everything the main thread did before the second image was submitted:')
    second = sorted(a for th, label, a, b in SPANS if th == 'MainThread' and label == 'sucre._restore_submit')[1]
    for a, b, label in sorted((a, b, label) for th, label, a, b in SPANS if th == 'MainThread' and a < second):
        print(f'  {a:7.3f} {b:7.3f} {(b - a) * 1e3:7.1f} ms  {label}')
    w = sorted((a, b) for th, label, a, b in SPANS if label == 'sucre._write_outputs')
    print('write_outputs: first start %.3f, last end %.3f, mean %.0f ms' % (w[0][0], max(b for _, b in w), sum(b - a for a, b in w) / len(w) * 1e3))
    d = sorted((a, b) for th, label, a, b in SPANS if label == 'loader._imread_rgb_u8')
    print('rgb decode: first start %.3f, last end %.3f, mean %.0f ms' % (d[0][0], max(b for _, b in d), sum(b - a for a, b in d) / len(d) * 1e3))


if __name__ == '__main__':
    main()
