"""First contact with RCCL before the driver's 8-rank run: a one-rank 'nccl' process group on the GPU box, every
all-reduce of the shared-water fit forced through it (tests/rccl_worker.py).  Runs in a child process: a process group
must not leak into the test session."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.timeout(600)
def test_shared_water_over_a_one_rank_rccl_group_equals_no_group_bit_for_bit():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR')}
    env.update(MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, str(ROOT / 'tests' / 'rccl_worker.py'), '10'], env=env, capture_output=True, text=True,
                         timeout=500)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert 'RCCL_OK backend=nccl world=1' in out.stdout, out.stdout[-1500:]
    print(out.stdout.strip().splitlines()[-1])
