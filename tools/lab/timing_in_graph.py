"""Lab: do the per-launch events of mgp_spmm_timing_* survive stream capture (hipExtLaunchKernelGGL inside a captured solve)?"""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
lib = _lib.lib()
y = wl["y"].view(-1, 1).contiguous()
plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8)
plan.solve(y, copy=False)                       # eager first solve
print("begin", lib.mgp_spmm_timing_begin(64), flush=True)
try:
    plan.solve(y, copy=False)                   # captured here
    print("captured solve ok, iters", plan.iters, flush=True)
    for _ in range(200):
        plan.solve(y, copy=False)               # replays
    torch.cuda.synchronize()
except Exception as e:
    print("capture with ext launches failed:", e, flush=True)
ms, cnt = ctypes.c_float(0), ctypes.c_int(0)
print("end", lib.mgp_spmm_timing_end(ctypes.byref(ms), ctypes.byref(cnt)), "launches", cnt.value, "avg us", ms.value / max(cnt.value, 1) * 1e3, flush=True)
