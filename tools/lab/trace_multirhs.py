"""Lab: only the C = 100 one-hot solve of bench.py's cg_multi_rhs block (for rocprofv3 --kernel-trace --stats)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
out = bench.multi_rhs_solve(wl)
print({k: out[k] for k in ("iterations", "solve_ms", "spmm_launches", "spmm_gbs")}, out["whole_chain_cg"])
