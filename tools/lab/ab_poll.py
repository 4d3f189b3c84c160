"""Lab: wall time per C3 CG solve against the completion poll's spin count (mgp_cg_set_poll_spin), same process."""
import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=0)
y = wl["y"].view(-1, 1).contiguous()
for _ in range(20):
    plan.solve(y, copy=False)
import gc; gc.collect(); gc.disable()
for spins in (0, 64, 0, 64, 16, 256):
    _lib.lib().mgp_cg_set_poll_spin(spins)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100):
            plan.solve(y, copy=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e6)
    print("spins %5d  %.2f us per solve (best of 5 x 100)" % (spins, best))
