"""Lab: the 12-column SpMM of the C3 bench graph, 60 launches (for PMC passes)."""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
dev = torch.device("cuda:0")
C = int(sys.argv[1]) if len(sys.argv) > 1 else 12
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr()
X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
_lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 60, None, _lib.stream()), "repeat")
torch.cuda.synchronize()
ms = ctypes.c_float(0); best = 1e9
for _ in range(4):
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 100, ctypes.byref(ms), _lib.stream()), "repeat")
    best = min(best, ms.value / 100 * 1e3)
print("C %d: us per launch %.2f" % (C, best))
