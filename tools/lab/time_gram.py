"""Lab: mgp_gram_f64 (A^T A, fp64 accumulation) on the fp64 matrix cores against the vector-FMA kernel: us per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import gram_f64
for n, b in ((60000, 128), (60000, 64), (1000000, 64), (1000000, 128)):
    A = torch.randn(n, b, device="cuda:0")
    for on in (0, 1):
        _lib.lib().mgp_gram_set_mfma(on)
        for _ in range(3): gram_f64(A)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): gram_f64(A)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 * 1e6
        print("n %7d b %3d mfma %d: %.1f us  %.1f TFLOP/s" % (n, b, on, us, 2.0 * n * b * b / us / 1e6), flush=True)
_lib.lib().mgp_gram_set_mfma(1)
