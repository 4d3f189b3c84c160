#!/usr/bin/env python
"""Lab: C = 1 tile SpMV back to back (mgp_spmm_repeat, 3 x 200 launches) -- for A/B runs with MGP_LAB_LIB (e.g. the
-DMGP_C1_NOCONFLICT build: dictionary reads without bank conflicts, wrong results, timing only)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
import bench
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr(); v = torch.rand(g.n, 1, device="cuda:0"); out = torch.empty_like(v)
ms = ctypes.c_float(0)
best = 1e9
for _ in range(4):
    lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 200, ctypes.byref(ms), _lib.stream())
    best = min(best, ms.value / 200 * 1e3)
print("us per launch %.3f" % best)
