"""Stage 2 of gen_h5_fixture.py -- runs under an interpreter that has h5py (no torch there): writes a matches file
with exactly the h5py calls the reference's loader.MatchesFile makes, in its order:

  save_matches, loader.py:68-76, once per kept view IN image_list ORDER:
      h5py.File(path, 'a', libver='latest'); create_group(image2.name);
      create_dataset 'u1','v1','u2','v2' (int16), 'd' (float32), 'I' = full((3, n), nan, float32)
  prepare_matches, loader.py:78-87:
      h5py.File(path, 'r+', libver='latest'); for every group: group['I'][()] = rgb[v2, u2].T

usage: python3.9 _h5_write_like_reference.py <handover.npz> <out.h5> [--stop-before-prepare]
(the last form leaves I NaN-prefilled: the half-written spill check_integrity must refuse, loader.py:89-101)
"""
import sys

import h5py
import numpy as np


def main(argv):
    src, dst = argv[0], argv[1]
    stop = '--stop-before-prepare' in argv
    z = np.load(src)
    order = [str(n) for n in z['order']]
    for name in order:                                                    # sfm.py:130-138: one save_matches per kept view
        with h5py.File(dst, 'a', libver='latest') as f:
            group = f.create_group(name)
            group.create_dataset('u1', data=z[f'{name}/u1'])
            group.create_dataset('v1', data=z[f'{name}/v1'])
            group.create_dataset('u2', data=z[f'{name}/u2'])
            group.create_dataset('v2', data=z[f'{name}/v2'])
            group.create_dataset('d', data=z[f'{name}/d'])
            group.create_dataset('I', data=np.full((3, len(z[f'{name}/u1'])), np.nan, dtype=np.float32))
    if stop:
        return
    with h5py.File(dst, 'r+', libver='latest') as f:                      # loader.py:78-87
        for name in [g for g in f]:
            group = f[name]
            group['I'][()] = z[f'{name}/I']


if __name__ == '__main__':
    main(sys.argv[1:])
