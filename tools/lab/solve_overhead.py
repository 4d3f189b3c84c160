"""Lab: fixed cost of one multi-column plan.solve (graph launch, host wait, copies) against its iterations: the factor solves of
training run ~14 iterations of ~22 us each.  ms per solve at several tolerances -> slope (per iteration) and intercept."""
import argparse, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
desc = wl["desc"].with_(scale=1.0, form=0, noise=0.0)
dB = desc.with_(nu=1, kappa=desc.kappa / math.sqrt(2), scale=1.0, pre=None, post=None)
for C in (1, 12, 100):
    torch.manual_seed(0)
    B = torch.randn(wl["graph"].n, C, device=dev)
    pts = []
    for tol in (3e-1, 1e-1, 2.5e-3, 1e-4, 1e-6):
        plan = CgPlan(dB, C, tol=tol, max_iter=1000, stop_mode=0)
        for _ in range(5):
            plan.solve(B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            plan.solve(B)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 30 * 1e3
        pts.append((plan.iters, ms))
        plan.close()
    (i0, m0), (i1, m1) = pts[1], pts[-1]
    slope = (m1 - m0) / max(1, i1 - i0)
    print("C %3d: (iterations, ms per solve) %s -> %.1f us per iteration, intercept %.3f ms" % (C, [(i, round(m, 3)) for i, m in pts], slope * 1e3, m0 - slope * i0), flush=True)
