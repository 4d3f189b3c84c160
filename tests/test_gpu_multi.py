"""More than one GPU: the partitioned solvers over real RCCL, one process per GPU, against the
single-GPU solve.  Skipped on a one-GPU box -- which is all the build box ever offered, so this test has NOT run on
hardware yet (docs/multi_gpu.md marks every N > 1 statement accordingly)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_partitioned_solvers_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box has %d" % (world, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tools", "check_dist.py")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "OK: %d ranks" % world in out.stdout
