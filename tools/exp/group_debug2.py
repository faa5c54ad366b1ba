import os, sys, subprocess, socket
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
if len(sys.argv) > 1:
    import numpy as np, torch
    import torch.distributed as dist
    import helpers
    from sucre_amd import engine, dist as sdist
    rank, lr, world = sdist.init_process_group()
    golden = helpers.load_fixture('plane_64x48_n4')
    views = engine.device_views_from_scene(golden.scene, 'cuda')
    tgt = int(golden['shared_targets'][rank])
    r = engine.Restoration(golden.scene.height, golden.scene.width, len(views))
    r.match(views[tgt], views); r.fit_init(views[tgt])
    T = 4
    tr = torch.zeros((T, 10), dtype=torch.float64, device='cuda')
    be = engine.HipWaterGroup([r], trace=tr)
    n = torch.tensor([be.n_obs()], dtype=torch.int64); sdist.all_reduce_sum(n); be.set_n_obs_total(int(n.item()))
    print(rank, 'total', int(n.item()), flush=True)
    for it in range(1, T + 1):
        s = be.grad(it)
        loc = s.cpu().numpy().copy()
        sdist.all_reduce_sum(s)
        print(rank, it, 'local', loc[:3], loc[9], 'reduced', s.cpu().numpy()[:3], s.cpu().numpy()[9], flush=True)
        be.step(it)
    be.finish()
    print(rank, 'trace', tr.cpu().numpy()[:, :4], flush=True)
    print(rank, 'golden', golden['shared_trace'][:T, :4], flush=True)
    dist.destroy_process_group()
else:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ps = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', LOCAL_WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SUCRE_DIST_BACKEND='gloo')
        ps.append(subprocess.Popen([sys.executable, __file__, 'worker'], env=env))
    print([p.wait() for p in ps])
