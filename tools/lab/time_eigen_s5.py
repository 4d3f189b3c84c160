"""Lab: eigensolve of the 1M-node swiss-roll graph (config C5, 50 pairs) at two tolerances."""
import os, sys, time, torch, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd.solvers import lanczos_smallest
from manifold_gp_amd import _lib
if os.environ.get("MGP_MT") == "0":          # A/B: without the matrix-core tile SpMM
    _lib.lib().mgp_spmm_set_mt_mode(0)
wl = bench.build_workload(argparse.Namespace(workload="s5", nodes=0, s5_order="random"), torch.device("cuda:0"), 0, 1)
data = wl["lap"].data
for tol in (1e-5,):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, V, res = lanczos_smallest(data, 50, tol=tol)
        torch.cuda.synchronize()
        print("tol %.0e eigensolve ms %.1f info %s max resid %.2e" % (tol, (time.perf_counter() - t0) * 1e3, lanczos_smallest.last_info, max(res)), flush=True)
