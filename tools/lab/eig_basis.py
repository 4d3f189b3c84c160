import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from manifold_gp_amd.solvers import lanczos_smallest
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
data = wl["lap"].data
for mb in (0, 104, 108, 112, 116, 120, 128, 144):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, V, res = lanczos_smallest(data, 100, tol=1e-5, max_basis=mb)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print("max_basis %3d: %.1f ms info %s max resid %.2e" % (mb, best, lanczos_smallest.last_info, max(res)), flush=True)
