"""Lab: C3 CG solve wall time + SpMV back to back for a given build of the library (argv[1] = path of libmgp_hip.so, default the
tree's): run it once per variant in the SAME gpurun call to compare builds on the same box."""
import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from manifold_gp_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
y = wl["y"].view(-1, 1).contiguous()
import gc
t_b2b = bench.time_spmv_kernel(wl)
plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=0)
for _ in range(3000):
    plan.solve(y, copy=False)
gc.collect(); gc.disable()
res = []
for rep in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        plan.solve(y, copy=False)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / 200 * 1e6)
print("%s: SpMV back to back %.2f us; solve best %.2f median %.2f us (9 x 200), iters %d" % (_lib.LIB_PATH, t_b2b * 1e6, min(res), sorted(res)[4], plan.iters), flush=True)
