"""Lab: multi-column SpMM (C > 16) on the bench graphs: wide tile kernel vs the per-entry gather kernel."""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
if len(sys.argv) > 2:
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr()
for C in (20, 32, 64, 84, 100, 128, 192, 256):
    X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
    out = []
    for mode in (1, 2, 0, 3):
        lib.mgp_spmm_set_dict_mode(2 if mode == 1 else 0)            # 2: the dictionary kernel wherever the shape allows
        lib.mgp_spmm_set_tile_wide_mode(2 if mode == 2 else 0)
        lib.mgp_spmm_set_v4_mode(0 if mode == 3 else 2)
        ms = ctypes.c_float(0.0)
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 5, None, _lib.stream()), "repeat")
        best = 1e9
        for _ in range(3):
            _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
            best = min(best, ms.value)
        out.append(best / 30 * 1e3)
    B = bench.spmm_bytes(g.n, g.M, C)
    print("C %3d  dict (lanes over columns) %.1f us (%.0f GB/s algorithmic)   wide tile %.1f us   float4 gather %.1f us   per-column gather %.1f us"
          % (C, out[0], B / out[0] / 1e3, out[1], out[2], out[3]), flush=True)
lib.mgp_spmm_set_tile_wide_mode(1); lib.mgp_spmm_set_v4_mode(1); lib.mgp_spmm_set_dict_mode(1)
