import os, sys, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
import ctypes as C
import numpy as np, torch
import helpers
from sucre_amd import engine, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else 'parent'
if mode == 'parent':
    ps = [subprocess.Popen([sys.executable, __file__, 'w', str(i)]) for i in range(2)]
    print([p.wait() for p in ps])
else:
    golden = helpers.load_fixture('plane_64x48_n4')
    views = engine.device_views_from_scene(golden.scene, 'cuda')
    tgt = golden.scene.target
    r = engine.Restoration(golden.scene.height, golden.scene.width, len(views))
    r.match(views[tgt], views)
    ws, H, W, n = r._geom
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    sums_t = r._region(_lib.WS_SUMS, torch.float64, 12)
    ref = None
    for rep in range(12):
        r.fit_init(views[tgt])
        allsums = []
        for it in range(1, 31):
            _lib.check(r.lib.sucre_fit_grad(ws, H, W, n, it, 0.05, 0.9, 0.999, 1e-8, 0, st))
            allsums.append(sums_t.clone())
            _lib.check(r.lib.sucre_fit_step(ws, H, W, n, it, 0.05, 0.9, 0.999, 1e-8, None, st))
        S = torch.stack(allsums).cpu().numpy()[:, :10]
        if ref is None:
            ref = S
        elif not np.array_equal(S, ref):
            bad = np.argwhere(S != ref); i0 = bad[:, 0].min()
            print(sys.argv[2], 'rep', rep, 'first differing iteration', i0 + 1, 'cols', sorted(set(bad[bad[:, 0] == i0][:, 1].tolist())),
                  'got', S[i0, 6:9], 'ref', ref[i0, 6:9], flush=True)
    print(sys.argv[2], 'done', flush=True)
