"""Standalone npz <-> HDF5 converter for the reference's matches-file layout (loader.py:68-76).

Runs under ANY interpreter that has h5py + numpy (it must not import torch or this package):
    python _h5_helper.py write <in.npz> <out.h5>     keys "<group>/<dataset>" -> groups / datasets
    python _h5_helper.py read  <in.h5>  <out.npz>
"""
import sys

import h5py
import numpy as np


def main(argv):
    mode, src, dst = argv
    if mode == 'write':
        data = np.load(src)
        with h5py.File(dst, 'w', libver='latest') as f:
            for key in data.files:
                group, name = key.rsplit('/', 1)
                g = f.require_group(group)
                g.create_dataset(name, data=data[key])
    elif mode == 'read':
        out = {}
        with h5py.File(src, 'r', libver='latest') as f:
            for group_name, group in f.items():
                for name, ds in group.items():
                    out[f'{group_name}/{name}'] = ds[()]
        np.savez(dst, **out)
    else:
        raise SystemExit(f'unknown mode {mode}')


if __name__ == '__main__':
    main(sys.argv[1:])
