#!/usr/bin/env python
"""Time the C == 1 SpMV for several column-panel widths (0 = gather kernel) on the bench graph."""
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manifold_gp_amd import _lib  # noqa: E402
from manifold_gp_amd.graph import KnnGraph, LaplacianData  # noqa: E402


class A:
    workload, gpus = "c3", 1
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def main():
    dev = torch.device("cuda:0")
    wl = bench.build_workload(A(), dev, 0, 1)
    g = wl["graph"]
    B = bench.spmm_bytes(g.n, g.M)
    lib = _lib.lib()
    v = torch.rand(g.n, 1, device=dev)
    out = torch.empty_like(v)
    res = []
    datas = {}
    lib.mgp_spmm_set_panel_mode(1)
    for pw in (0, 32768):
        gg = KnnGraph.from_coo(g.edge_index, g.edge_value, g.n, panel_width=pw)
        datas[pw] = LaplacianData(gg, wl["eps"], True)
    for rnd in range(3):
      for nt in (0, 1):
        for pw, data in datas.items():
            lib.mgp_spmm_set_stream_nt(nt)
            pw = pw + nt
            lib.mgp_spmm_set_group_hint(8)
            lib.mgp_spmm_set_rows_in_flight(2)
            csr = data.csr()
            st = _lib.stream()
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 10, None, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 100, None, st)
            e1.record()
            torch.cuda.synchronize()
            res.append((pw, data.graph.panels, data.graph.nnz, round(e0.elapsed_time(e1) / 100 * 1e3, 2)))
    best = {}
    for pw, P, nnz, t in res:
        best[pw] = min(best.get(pw, (1e9,))[0], t), P, nnz
    for pw, (t, P, nnz) in best.items():
        print("panel_width %6d panels %3d nnz_padded %8d : %6.2f us  %7.1f GB/s" % (pw, P, nnz, t, B / t / 1e3))


if __name__ == "__main__":
    main()
