"""Time the C in {4, 8, 12, 16} SpMM on the C3 bench graph: LDS-dictionary tile kernel vs the per-entry gather kernel."""
import ctypes, os, sys, time, argparse
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr()
for C in (4, 8, 12, 16):
    X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
    for mode in (1, 0):
        lib.mgp_spmm_set_tile_small_mode(mode)
        ms = ctypes.c_float(0.0)
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 20, None, _lib.stream()), "repeat")
        best = 1e9
        for _ in range(3):
            _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 100, ctypes.byref(ms), _lib.stream()), "repeat")
            best = min(best, ms.value)
        B = bench.spmm_bytes(g.n, g.M, C)
        print("C %2d  %s  %.2f us / launch  %.0f GB/s algorithmic" % (C, "tile-dictionary" if mode else "per-entry gather", best * 10, B / (best * 1e-5) / 1e9))
lib.mgp_spmm_set_tile_small_mode(1)
