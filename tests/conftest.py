import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no libmgp_hip.so (built artefacts are not tracked): build it once, exactly as
    __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU)."""
    so = os.path.join(ROOT, "manifold_gp_amd", "libmgp_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "manifold_gp_amd", "csrc", "build.sh")])


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return _load


def ref_round_equal(a, b, decimals=5, count=10):
    """The reference's own pass criterion (test/_test_functions.py:15-20): the first `count`
    entries agree after round(., 5)."""
    a = [round(float(v), decimals) for v in np.asarray(a).reshape(-1)[:count]]
    b = [round(float(v), decimals) for v in np.asarray(b).reshape(-1)[:count]]
    return a == b
