#!/usr/bin/env python3
"""End-to-end timing of the reference-compatible CLI on a synthetic survey written to disk (PNG images, uint16
depth maps, COLMAP text model): where does a user's wall time go once the GPU part takes ~30 ms per image?
usage (GPU box): python3 tools/cli_survey_bench.py [width height grid_x grid_y n_restore [spacing]]"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from PIL import Image as PILImage  # noqa: E402

from sucre_amd import sfm, sucre, synth  # noqa: E402


def main():
    W, H, gx, gy, n = (int(a) for a in (sys.argv[1:6] + [1920, 1080, 6, 5, 8][len(sys.argv[1:6]):]))
    spacing = float(sys.argv[6]) if len(sys.argv) > 6 else 0.1   # camera spacing in footprints (0.1 = 90 % overlap)
    survey = synth.make_survey(W, H, gx, gy, seed=3, spacing=spacing, device='cuda')
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp)
        t0 = time.perf_counter()
        synth.write_to_disk(survey, root)
        print(f'wrote {len(survey.views)} views in {time.perf_counter() - t0:.1f}s', flush=True)
        first = 1 if n >= gx * gy else gx * (gy // 2) + 1
        argv = ['--image-dir', str(root / 'images'), '--depth-dir', str(root / 'depth'), '--model-dir', str(root / 'model'),
                '--output-dir', str(root / 'out'), '--image-ids', str(first), str(first + n)]
        profile = os.environ.get('SUCRE_CLI_PROFILE', '0') != '0'   # cProfile slows the main thread by ~25 %
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        if profile:
            pr.enable()
        sucre.main(argv)
        if profile:
            pr.disable()
        dt = time.perf_counter() - t0
        print(f'CLI: {n} images in {dt:.2f}s = {dt / n * 1e3:.0f} ms/image' + (' (under cProfile)' if profile else ''), flush=True)
        from sucre_amd import engine
        print(f'model of {len(survey.views)} images; workspaces held at the end (views of capacity, GB):',
              sorted((r.capacity, round(r.ws.numel() / 2 ** 30, 2)) for r in engine._POOL.values()), flush=True)
        if not profile:
            return
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35)
        print(s.getvalue()[:6000])


if __name__ == '__main__':
    main()
