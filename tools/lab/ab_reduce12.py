import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
for mode in (1, 2, 1, 2):
    _lib.lib().mgp_cg_set_reduce_once(mode)
    print("reduce_once mode", mode, " 12 columns solve_ms", bench.multi_rhs_solve(wl, columns=12)["solve_ms"],
          " 16 columns", bench.multi_rhs_solve(wl, columns=16)["solve_ms"], " 4 columns", bench.multi_rhs_solve(wl, columns=4)["solve_ms"])
