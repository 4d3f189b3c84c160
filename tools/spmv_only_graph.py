#!/usr/bin/env python
"""C == 1 SpMV on the bench graph replayed from a hipGraph (mgp_spmm_repeat): for PMC passes."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd import _lib
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr(); v = torch.rand(g.n, 1, device="cuda:0"); out = torch.empty_like(v)
ms = ctypes.c_float(0)
lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 60, ctypes.byref(ms), _lib.stream())
print("us per launch", ms.value / 60 * 1e3)
torch.cuda.synchronize()
