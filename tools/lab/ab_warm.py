"""Lab: does the timed region of bench.py (20 solves after 5 warm-up solves) see ramped GPU clocks?"""
import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=0)
y = wl["y"].view(-1, 1).contiguous()
for _ in range(20):
    plan.solve(y, copy=False)
import gc; gc.collect(); gc.disable()
def timed(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        plan.solve(y, copy=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
for idle_ms, pre in ((50, 0), (50, 0), (0, 0), (50, 300), (50, 1000), (5, 0), (50, 3000), (50, 0)):
    time.sleep(idle_ms / 1e3)
    for _ in range(pre):
        plan.solve(y, copy=False)
    for _ in range(5):
        plan.solve(y, copy=False)
    print("idle %3d ms, %4d extra solves, then 5 warm-up + 20 timed: %.2f us per solve   (next 20: %.2f, next 200: %.2f)"
          % (idle_ms, pre, timed(20), timed(20), timed(200)))
