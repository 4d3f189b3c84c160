"""Lab: one SpMM configuration on the C3 graph, 40 launches (for PMC passes): spmm_one.py <C> <v4 mode 0/2> [wide 0/2] [dict 0/2] [unused] [mt 0/1: the matrix-core tile kernel, on the relabelled / natural CSR with its image]"""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=int(os.environ.get("MGP_NODES", 0)), s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
mt = int(sys.argv[6]) if len(sys.argv) > 6 else 0
# (mt: what the wide products run on -- the chain-relabelled matrix where the graph has such an order; MGP_NO_CHAIN=1: the given order)
csr = ((lap.data.relabelled() if os.environ.get("MGP_NO_CHAIN") else lap.data.wide_relabelled()) or lap.data).csr(wide=True) if mt else lap.data.csr()
lib.mgp_spmm_set_mt_mode(mt)
C = int(sys.argv[1]); lib.mgp_spmm_set_v4_mode(int(sys.argv[2])); lib.mgp_spmm_set_tile_wide_mode(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
lib.mgp_spmm_set_dict_mode(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
ms = ctypes.c_float(0.0)
_lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 40, ctypes.byref(ms), _lib.stream()), "repeat")
print("C", C, "us per launch", ms.value / 40 * 1e3)
