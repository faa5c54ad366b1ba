"""RCCL on the one-GPU box (tests/test_gpu_rccl.py): a process group of ONE rank over backend 'nccl' (= RCCL on ROCm), with
``dist.COLLECTIVE_AT_WORLD_1`` set so that every all-reduce of ``dist.fit_shared_water`` really goes through RCCL --
the int64 observation count, then per iteration the float64 view into the engine's own buffer (HipWaterBackend: the
workspace's sums; HipWaterGroup: the group buffer's), reduced on RCCL's stream between two launches on the launch stream.
A sum over one rank is the identity, so every trace, parameter and J must equal the run without a process group bit
for bit; anything else (a stale read, a reduce that lands after the next launch read the sums) shows as a difference."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]


def run_all(scene, views, T, use_stream):
    """The three shared-water drivers on the same image: split launches (HipWaterBackend), the single-launch group of one
    image and a group of two images; J-parameter and closed form.  Returns a list of (label, trace, params, [J...])."""
    from sucre_amd import dist as sdist
    from sucre_amd import engine
    out = []
    dev = torch.device('cuda', 0)
    stream = torch.cuda.Stream(dev) if use_stream else torch.cuda.current_stream(dev)
    with torch.cuda.stream(stream):
        for closed in (False, True):
            r = engine.Restoration(scene.height, scene.width, len(views), device=dev)
            r.match(views[scene.target], views)
            r.fit_init(views[scene.target])
            trace = torch.zeros((T, 10), dtype=torch.float64, device=dev)
            sdist.fit_shared_water(engine.HipWaterBackend(r, use_closed_form=closed, trace=trace), T)
            if closed:
                r.update_J()
            out.append((f'backend closed={closed}', trace.cpu().numpy(), r.params().cpu().numpy(), [r.J().cpu().numpy()]))
            r2 = engine.Restoration(scene.height, scene.width, len(views), device=dev)
            other = scene.target - 1
            r2.match(views[other], views)
            for rs, tgts in (([r], [scene.target]), ([r, r2], [scene.target, other])):
                for x, t in zip(rs, tgts):
                    x.fit_init(views[t])
                trace = torch.zeros((T, 10), dtype=torch.float64, device=dev)
                sdist.fit_shared_water(engine.HipWaterGroup(rs, use_closed_form=closed, trace=trace), T)
                out.append((f'group of {len(rs)} closed={closed}', trace.cpu().numpy(), rs[0].params().cpu().numpy(),
                            [x.J().cpu().numpy() for x in rs]))
    torch.cuda.synchronize()
    return out


def main(T: int) -> None:
    import torch.distributed as dist
    from sucre_amd import dist as sdist
    from sucre_amd import engine, synth
    scene = synth.make_scene(320, 240, 8, seed=21, device='cuda')
    views = engine.device_views_from_scene(scene, 'cuda')
    alone = run_all(scene, views, T, use_stream=False)
    assert not dist.is_initialized()
    os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    sdist.COLLECTIVE_AT_WORLD_1 = True
    rank, local_rank, world = sdist.init_process_group(backend='nccl')
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    calls = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append((t.dtype, t.is_cuda, int(t.numel())))
        return real(t, *a, **k)
    dist.all_reduce = counting
    try:
        for use_stream in (False, True):   # on the default stream and on a side stream (what bench.py's slots use)
            calls.clear()
            over_rccl = run_all(scene, views, T, use_stream)
            n_float = sum(1 for d, c, n in calls if d == torch.float64 and c)
            n_int = sum(1 for d, c, n in calls if d == torch.int64 and c)
            assert n_float == 6 * T and n_int == 6, (n_float, n_int, calls[:4])   # six fits: T sums each + their n_obs
            for (la, ta, pa, Ja), (lb, tb, pb, Jb) in zip(alone, over_rccl):
                assert la == lb
                assert np.array_equal(ta, tb), (la, 'trace', use_stream, np.abs(ta - tb).max())
                assert np.array_equal(pa, pb), (la, 'params', use_stream)
                for a, b in zip(Ja, Jb):
                    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), (la, 'J', use_stream)
                assert np.abs(ta[-1, 1:] - 0.1).max() > 1e-2   # the parameters moved
    finally:
        dist.all_reduce = real
    # a barrier and an object gather over RCCL, as bench.py's N > 1 path issues them
    dist.barrier(device_ids=[0])
    names = [None]
    dist.all_gather_object(names, torch.cuda.get_device_name(0))
    print(f'RCCL_OK backend={dist.get_backend()} world={dist.get_world_size()} device={names[0]} '
          f'all_reduce calls per run: {6 * T} float64 + 6 int64', flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
