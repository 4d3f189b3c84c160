"""Lab: eigensolve with the Gershgorin bound (mode 0) against the Krylov estimate of lambda_max (mode 1)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import lanczos_smallest
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 100
class A: workload, nodes, gpus, s5_order = name, 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
data = wl["lap"].data
out = {}
for mode in (0, 1, 0, 1):
    _lib.lib().mgp_lanczos_set_bound_mode(mode)
    best = 1e9
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, V, res = lanczos_smallest(data, m, tol=1e-5)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    out[mode] = (ev.clone(), V.clone())
    print("%s m %d bound mode %d: %.1f ms info %s max resid %.3e" % (name, m, mode, best, lanczos_smallest.last_info, float(max(res))), flush=True)
e0, e1 = out[0][0], out[1][0]
print("eigenvalue difference between the modes: max |d| %.3e (lambda_m %.4e)" % (float((e0 - e1).abs().max()), float(e0[-1])))
_lib.lib().mgp_lanczos_set_bound_mode(1)
