import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd import _lib
class A: workload, nodes, gpus, s5_order = "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
sym = lap._symmetric_twin()
v = torch.rand(g.n, 1, device="cuda:0"); out = torch.empty_like(v)
for rnd in range(3):
    for mode in (1, 2, 3, 4):
        lib.mgp_spmm_set_tile_mode(mode)
        csr = sym.data.csr(); st = _lib.stream()
        lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 10, None, st)
        ms = ctypes.c_float(0.0)
        lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 200, ctypes.byref(ms), st)
        print("mode", mode, "us", round(ms.value / 200 * 1e3, 2), flush=True)
