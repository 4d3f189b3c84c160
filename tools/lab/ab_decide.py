"""Lab: wall time per C3 CG solve with the stopping decision inside the graph's last update launch
(mgp_cg_set_decide_in_update 1) against the separate decision + marker launches (0); same process, plans created per setting.
Also checks that both return the same solution, iteration count and residual."""
import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
y = wl["y"].view(-1, 1).contiguous()
import gc
sols = {}
y2 = y.clone()
for mode in (0, 1, 0, 1):
    _lib.lib().mgp_cg_set_decide_in_update(1 if mode else 0)
    plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=0)
    for _ in range(300):
        plan.solve(y, copy=False)
    gc.collect(); gc.disable()
    best = 1e9
    for rep in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            plan.solve(y, copy=False)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
    for _ in range(10):
        plan.solve(y2, copy=False); plan.solve(y, copy=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200):
        plan.solve(y2 if i & 1 else y, copy=False)
    torch.cuda.synchronize()
    alt = (time.perf_counter() - t0) / 200 * 1e6
    gc.enable()
    x = plan.solve(y).clone()
    sols[mode] = (x, plan.iters, plan.status, max(plan.resid), plan.applies)
    print("mode %d (0 separate decision launch, 1 decision in the last update)  %.2f us per solve (best of 7 x 200), "
          "%.2f alternating rhs;  iters %d status %d resid %.3e applies %d"
          % (mode, best, alt, plan.iters, plan.status, max(plan.resid), plan.applies), flush=True)
    plan.close()
for m in (1,):
    a, b = sols[0], sols[m]
    print("mode", m, "same solution:", torch.equal(a[0], b[0]), "same iters/status/resid/applies:", a[1:] == b[1:])
