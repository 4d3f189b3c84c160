#!/usr/bin/env python
"""Time the fused SpMM (C > 1) on the bench graph for several column counts.  GPU box only."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd import _lib
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
a = A()
if len(sys.argv) > 1:
    a.workload, a.nodes = "s5", int(sys.argv[1])
wl = bench.build_workload(a, torch.device("cuda:0"), 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap._symmetric_twin().data.csr()
res = []
for C in (2, 4, 8, 11, 12, 16, 32, 64, 125, 128, 256):
    X = torch.rand(g.n, C, device="cuda:0"); Y = torch.empty_like(X)
    lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 5, None, _lib.stream())
    ms = ctypes.c_float(0)
    lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 20, ctypes.byref(ms), _lib.stream())
    us = ms.value / 20 * 1e3
    gather_gb = g.nnz * C * 4 / 1e9
    res.append(dict(C=C, us=round(us, 1), gather_TBps=round(gather_gb / (us * 1e-6) / 1e3, 2), gflops=round(2 * g.nnz * C / us / 1e3, 1)))
print(json.dumps(dict(workload=wl["name"], nnz=g.nnz, results=res), indent=1))
