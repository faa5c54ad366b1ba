import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / 'tests'):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


# test infrastructure: this image keeps h5py in a second interpreter; the product only looks at the variable
import os  # noqa: E402
if 'SUCRE_H5PY_PYTHON' not in os.environ and Path('/opt/conda/bin/python3.9').exists():
    os.environ['SUCRE_H5PY_PYTHON'] = '/opt/conda/bin/python3.9'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # a fresh checkout has no built artefacts (they are git-ignored): build the HIP library and the CPU oracle once
    from sucre_amd import _lib
    if not _lib.LIB_PATH.exists() or not list((ROOT / 'oracle').glob('*.so')):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope='session', params=['plane_64x48_n4', 'relief_96x64_n6'])
def golden(request):
    import helpers
    return helpers.load_fixture(request.param)
