"""Prints the end-to-end posterior errors (HIP pipeline vs the reference pipeline's float64 goldens,
tests/golden/dumbbell_posterior.npz) for several eigensolver tolerances (run on the GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import manifold_gp_amd as mgp  # noqa: E402
from test_gpu_parity import _end_to_end_posterior_errors  # noqa: E402


def golden(name):
    return dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))


dev = torch.device("cuda:0")
for tol in (1e-5, 1e-6, 1e-7):
    for tag in ("k10", "k50"):
        for norm in ("symmetric", "randomwalk"):
            for nu in (1, 2):
                e = _end_to_end_posterior_errors(mgp, golden, dev, tag, norm, nu, eigen_tol=tol)
                print("tol %.0e %s %-10s nu=%d: %s" % (tol, tag, norm, nu, " ".join("%s %.2e" % kv for kv in e.items())), flush=True)
