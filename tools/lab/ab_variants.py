"""Lab: time the C in {4, 8, 12, 16} SpMM with another build of the library (A/B of kernel variants).
usage: ab_variants.py <path/to/libmgp_hip.so> [workload]"""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[2] if len(sys.argv) > 2 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr()
res = []
for C in (4, 8, 12, 16):
    X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
    ms = ctypes.c_float(0.0)
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 20, None, _lib.stream()), "repeat")
    best = 1e9
    for _ in range(3):
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 100, ctypes.byref(ms), _lib.stream()), "repeat")
        best = min(best, ms.value)
    res.append("C%d %.2f us" % (C, best * 10))
print(sys.argv[1].split("/")[-2], "  ".join(res))
