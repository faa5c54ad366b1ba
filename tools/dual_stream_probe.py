#!/usr/bin/env python3
"""Experiment: do two images in flight on two HIP streams hide the per-launch ramp-up/tail of fit_grad_kernel?
Usage (GPU box): python3 tools/dual_stream_probe.py [n_images] [n_streams...]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from sucre_amd import engine, synth  # noqa: E402


def main():
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    stream_counts = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
    dev = torch.device('cuda', 0)
    scene = synth.make_scene(1920, 1080, 64, seed=0, device=dev)
    views = engine.device_views_from_scene(scene, dev)
    tgt = views[scene.target]
    for ns in stream_counts:
        restos = [engine.Restoration(1080, 1920, len(views), device=dev) for _ in range(ns)]
        streams = [torch.cuda.Stream(dev) for _ in range(ns)]

        outs = []

        def run(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % ns]):
                    r = restos[i % ns]
                    r.match(tgt, views, min_cover=1e-6)
                    r.fit_init(tgt)
                    tr = r.fit(200, record_trace=True)
                    outs.append((r.J(), tr))
                    if len(outs) > 6:       # soak check: every restoration of the same image gives the same bits
                        J, t = outs.pop(0)
                        assert torch.equal(torch.nan_to_num(J), torch.nan_to_num(outs[0][0])) and torch.equal(t, outs[0][1])
        run(ns)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run(n_images)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f'streams={ns}: {best / n_images * 1e3:.2f} ms/image  {n_images / best:.2f} images/s', flush=True)
        ref = restos[0].J().clone()
        for r in restos[1:]:
            assert torch.equal(torch.nan_to_num(r.J()), torch.nan_to_num(ref)), 'streams disagree'
        del restos
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
