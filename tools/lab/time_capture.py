import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
desc = wl["desc"].with_(scale=1.0, form=0, noise=0.0, nu=1)
torch.manual_seed(0)
for C in (1, 12):
    B = torch.randn(wl["graph"].n, C, device=dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan = CgPlan(desc, C, tol=2.5e-3, max_iter=1000, stop_mode=0)
        torch.cuda.synchronize(); tc = time.perf_counter() - t0
        ts = []
        for i in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); plan.solve(B); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("C %2d plan create %.2f ms; solves (ms): %s  iters %d" % (C, tc * 1e3, [round(t, 2) for t in ts], plan.iters))
        plan.close()
