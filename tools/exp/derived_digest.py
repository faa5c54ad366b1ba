"""SHA-256 of the float32 matrices torch derives on THIS host (K.inverse(), R.T, -R.T @ t) for the config-2 scene."""
import hashlib, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from sucre_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W, H = (1920, 1080) if n == 64 else (640, 480)
# poses only: the renders are not needed
import sucre_amd.synth as S
orig = S.render_view
S.render_view = lambda *a, **k: (torch.zeros(1, 1, dtype=torch.int32), torch.zeros(1, 1, 3, dtype=torch.uint8))
scene = synth.make_scene(W, H, n, seed=0)
S.render_view = orig
print('threads', torch.get_num_threads(), torch.__config__.show().split('\n')[3:6])
hk = hashlib.sha256(scene.K.inverse().numpy().tobytes()).hexdigest()[:16]
print('Kinv', hk, scene.K.inverse().numpy().ravel().view(np.uint32))
ht = hashlib.sha256()
for i, v in enumerate(scene.views):
    tinv = -v.R.T @ v.t
    ht.update(tinv.numpy().tobytes())
    if i < 3:
        print(i, tinv.numpy().ravel().view(np.uint32))
print('tinv', ht.hexdigest()[:16])
