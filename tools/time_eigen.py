#!/usr/bin/env python
"""Warm timing of the eigensolve (100 smallest eigenpairs of the C3 graph Laplacian)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd.solvers import lanczos_smallest
from manifold_gp_amd import _lib
if os.environ.get("MGP_GRAM") == "0":        # A/B: Gram blocks by fp64 vector FMAs instead of the fp64 matrix cores
    _lib.lib().mgp_gram_set_mfma(0)
if os.environ.get("MGP_MT") == "0":          # A/B: without the matrix-core tile SpMM
    _lib.lib().mgp_spmm_set_mt_mode(0)
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
data = wl["lap"].data
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tols = [float(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1e-5]
for tol in tols:
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, V, res = lanczos_smallest(data, m, tol=tol)
        torch.cuda.synchronize()
        print("tol %.0e eigensolve ms %.1f info %s max resid %.2e" % (tol, (time.perf_counter() - t0) * 1e3, lanczos_smallest.last_info, max(res)), flush=True)
