"""One rank of the two-process shared-water GPU test (tests/test_gpu_dist.py): the real single-launch HipWaterGroup driven by
dist.fit_shared_water over a gloo process group, both ranks on the box's one GPU."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]


def main(out_dir: str, fixture: str, T: int) -> None:
    import helpers
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    rank, local_rank, world = sdist.init_process_group()
    golden = helpers.load_fixture(fixture)
    tgt = int(golden['shared_targets'][rank])
    dev = torch.device('cuda', local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    views = engine.device_views_from_scene(golden.scene, dev)
    r = engine.Restoration(golden.scene.height, golden.scene.width, len(views), device=dev)
    r.match(views[tgt], views)
    r.fit_init(views[tgt])
    trace = torch.zeros((T, 10), dtype=torch.float64, device=dev)
    be = engine.HipWaterGroup([r], trace=trace)
    sdist.fit_shared_water(be, T)
    torch.cuda.synchronize()
    import torch.distributed as dist
    np.savez(Path(out_dir) / f'rank{rank}.npz', trace=trace.cpu().numpy(), J=r.J().cpu().numpy(),
             params=r.params().cpu().numpy(), backend=dist.get_backend(), world=world)
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
