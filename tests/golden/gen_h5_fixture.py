"""A matches file in the reference's on-disk format, produced by the reference's own matching and written with the
reference's own sequence of h5py calls (task 7).

Build container only:  python tests/golden/gen_h5_fixture.py
  stage 1 (this interpreter: torch + the reference through ref_harness): the reference's match_two_way over the views of
          the committed fixture scene plane_64x48_n4, the min_cover rule (sfm.py:136), d = depth2[v2, u2] (sfm.py:137),
          the int16 casts of save_matches (loader.py:71-74) and I = rgb2[v2, u2].T (loader.py:87), views in image_list
          order -> a hand-over .npz;
  stage 2 (/opt/conda/bin/python3.9, the interpreter of this image that has h5py; the reference's loader itself cannot
          be imported there -- no torch): _h5_write_like_reference.py replays loader.py:68-87 call for call.
Writes tests/golden/ref_layout_plane_64x48_n4.h5 and ..._unprepared.h5 (I still NaN: what a crash between save_matches
and prepare_matches leaves, loader.py:76 -- check_integrity must refuse it).
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path[:0] = [str(HERE.parent.parent), str(HERE.parent), str(HERE)]

import helpers  # noqa: E402
import ref_harness as rh  # noqa: E402

H5PY_PYTHON = '/opt/conda/bin/python3.9'
NAME = 'plane_64x48_n4'


def main():
    fx = helpers.load_fixture(NAME)
    per_view, md, target = rh.reference_matches(fx.scene, min_cover=1e-6)
    sfm, _, _, _ = rh.import_reference()
    images = rh.build_reference_images(fx.scene, sfm)
    # image_list order is whatever the caller's list is (sucre.py:238-239: the COLMAP model's order): REVERSED here, so that
    # a reader that relied on the order of creation instead of h5py's iteration by name would be caught
    per_view, images = per_view[::-1], images[::-1]
    hand = {'order': np.array([r['name'] for r in per_view if r['kept']])}      # kept views only (sfm.py:136)
    for rec, other in zip(per_view, images):
        if not rec['kept']:
            continue
        n = rec['name']
        for k in ('u1', 'v1', 'u2', 'v2'):
            hand[f'{n}/{k}'] = rec[k].numpy()                                    # already .short() (loader.py:71-74)
            assert hand[f'{n}/{k}'].dtype == np.int16
        hand[f'{n}/d'] = rec['d'].numpy()
        hand[f'{n}/I'] = other.get_rgb()[rec['v2'].long(), rec['u2'].long()].T.numpy()   # loader.py:87
        assert hand[f'{n}/d'].dtype == np.float32 and hand[f'{n}/I'].dtype == np.float32
    with tempfile.TemporaryDirectory() as tmp:
        npz = Path(tmp) / 'hand.npz'
        np.savez(npz, **hand)
        for out, extra in ((HERE / f'ref_layout_{NAME}.h5', []), (HERE / f'ref_layout_{NAME}_unprepared.h5', ['--stop-before-prepare'])):
            out.unlink(missing_ok=True)
            subprocess.run([H5PY_PYTHON, str(HERE / '_h5_write_like_reference.py'), str(npz), str(out)] + extra, check=True)
            print(out.name, out.stat().st_size, 'bytes;', len(hand['order']), 'groups in write order', list(hand['order']))


if __name__ == '__main__':
    main()
