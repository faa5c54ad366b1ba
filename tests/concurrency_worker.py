"""One of two processes that fit the same image on the one GPU at the same time (tests/test_gpu_dist.py): workgroups
of a launch then start at different times and memory latencies stretch, which is when hand-counted waits and
hand-offs show their races.  Prints one digest per repetition: fused per-image path, single-launch group path and the
batch launch (batch_iter_kernel + batch_tail_kernel) of the one image, in turn."""
import hashlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]


def main(fixture: str, reps: int, T: int) -> None:
    import helpers
    from sucre_amd import engine
    golden = helpers.load_fixture(fixture)
    sc = golden.scene
    views = engine.device_views_from_scene(sc, 'cuda')
    r = engine.Restoration(sc.height, sc.width, len(views))
    out = []
    for rep in range(reps):
        r.match(views[sc.target], views)     # matching, compaction and plans are redone every time: they are under test too
        r.fit_init(views[sc.target])
        if rep % 3 == 0:
            t = r.fit(T)
        elif rep % 3 == 2:
            t = engine.fit_batch([r], T)[0]
        else:
            t = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
            g = engine.HipWaterGroup([r], trace=t)
            g.set_n_obs_total(r.n_obs())
            for it in range(1, T + 1):
                g.grad(it)
                g.step(it)
            g.finish()
        torch.cuda.synchronize()
        out.append(hashlib.md5(t.cpu().numpy().tobytes() + r.J().cpu().numpy().tobytes()).hexdigest())
    print('DIGESTS ' + ' '.join(out), flush=True)


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
