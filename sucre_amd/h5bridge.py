"""HDF5 access for the matches file in the reference's on-disk layout (loader.py:56-130):

    /<image name>/u1, v1, u2, v2   int16[n]        d  float32[n]        I  float32[3, n]

h5py is used in-process when importable; otherwise the small converter ``_h5_helper.py`` is run under an
interpreter that has it, named by the environment variable ``SUCRE_H5PY_PYTHON`` (no default: nothing outside the
environment is probed), with an ``.npz`` hand-over.  Without either, ``loader.MatchesFile`` keeps matches as ``.npz``.
"""
from __future__ import annotations

import os
import subprocess
import tempfile
from pathlib import Path

import numpy as np

_HELPER = Path(__file__).resolve().parent / '_h5_helper.py'
DATASETS = ('u1', 'v1', 'u2', 'v2', 'd', 'I')


def _external_python() -> str | None:
    cand = os.environ.get('SUCRE_H5PY_PYTHON')
    return cand if cand and Path(cand).exists() else None


def available() -> bool:
    try:
        import h5py  # noqa: F401
        return True
    except ImportError:
        return _external_python() is not None


def _flatten(groups: dict) -> dict:
    return {f'{g}/{k}': np.asarray(v) for g, ds in groups.items() for k, v in ds.items()}


def read_npz_groups(path: Path) -> dict:
    """The ``.npz`` fallback written by ``loader.MatchesFile.save`` when no h5py is around -> same dict as ``read_groups``."""
    with np.load(Path(path)) as data:
        return _unflatten({k: data[k] for k in data.files})


def _unflatten(flat) -> dict:
    groups: dict = {}
    for key in sorted(flat):
        g, k = key.rsplit('/', 1)
        groups.setdefault(g, {})[k] = np.asarray(flat[key])
    return dict(sorted(groups.items()))   # h5py iterates groups in name order; so do we


def write_groups(path: Path, groups: dict) -> Path:
    """groups = {image name: {dataset name: array}} -> HDF5 file at ``path``."""
    path = Path(path)
    try:
        import h5py
        with h5py.File(path, 'w', libver='latest') as f:
            for g, ds in groups.items():
                grp = f.create_group(g)
                for k, v in ds.items():
                    grp.create_dataset(k, data=np.asarray(v))
        return path
    except ImportError:
        py = _external_python()
        if py is None:
            raise RuntimeError('no h5py available (install it or point SUCRE_H5PY_PYTHON at an interpreter that has it)')
        with tempfile.TemporaryDirectory() as tmp:
            npz = Path(tmp) / 'groups.npz'
            np.savez(npz, **_flatten(groups))
            subprocess.run([py, str(_HELPER), 'write', str(npz), str(path)], check=True)
        return path


def read_groups(path: Path) -> dict:
    """HDF5 file -> {image name: {dataset name: array}}, groups in name order."""
    path = Path(path)
    try:
        import h5py
        out = {}
        with h5py.File(path, 'r', libver='latest') as f:
            for g, grp in f.items():
                out[g] = {k: ds[()] for k, ds in grp.items()}
        return out
    except ImportError:
        py = _external_python()
        if py is None:
            raise RuntimeError('no h5py available (install it or point SUCRE_H5PY_PYTHON at an interpreter that has it)')
        with tempfile.TemporaryDirectory() as tmp:
            npz = Path(tmp) / 'groups.npz'
            subprocess.run([py, str(_HELPER), 'read', str(path), str(npz)], check=True)
            with np.load(npz) as data:
                return _unflatten({k: data[k] for k in data.files})
