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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """What the reference-pinned checks MEASURED in this run (helpers.PARITY_ROWS), <= 25 lines, so that the tail of a
    ``pytest -q`` record carries numbers: per check the per-channel RMS(J), max |d B,beta,gamma|, max rel d cost vs the
    REFERENCE's stored outputs, whether the regenerated inputs were the reference's (they must be: a differing scene is a red
    test) and whether this host derives the reference host's float32 camera matrices ('own') or the stored ones were used."""
    import helpers
    lines = helpers.parity_summary_lines(25)
    if lines:
        terminalreporter.write_sep('=', 'parity vs the reference (measured in this run)')
        for ln in lines:
            terminalreporter.write_line(ln)
