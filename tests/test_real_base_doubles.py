"""The real-base branch of manifold_gp_amd/_compat.py (MRO mixin ahead of linear_operator.LinearOperator, constructor
forwarding, gpytorch.kernels.Kernel registration) executed against test doubles of the two packages
(tests/doubles/README.md: the documented call contract only; the image has neither package).  Each test runs
tests/doubles/run_real_base.py in a child process so that the doubles never enter this interpreter's module table."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(mode):
    env = dict(os.environ)
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "doubles", "run_real_base.py"), mode], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_real_base_branch_constructs_and_rebuilds_on_cpu():
    """graph_laplacian_operator.py:35-43: every tensor reaches LinearOperator.__init__, representation() is all tensors,
    representation_tree() rebuilds each of the five operators; _HipEntryPoints wins the MRO; settings fall through."""
    assert "REAL_BASE_CPU_OK" in _run("cpu")


@pytest.mark.gpu
def test_real_base_branch_computes_on_gpu():
    """The same branch with compute on cuda:0: library-side matmul / to_dense / diagonal dispatch into the HIP hooks after
    a rebuild from the representation, solve / inv_quad_logdet / _solve(num_tridiag) through _HipEntryPoints, and the
    kernel under gpytorch's parameter / constraint / prior registration returning linear_operator root operators."""
    assert "REAL_BASE_GPU_OK" in _run("gpu")
