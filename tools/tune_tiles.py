#!/usr/bin/env python
"""Time the C == 1 SpMV on the bench graph: gather kernel vs the row-tile LDS-dictionary kernel for
tile_rows in {32, 64, 128}; prints the dictionary statistics (distinct columns per tile).  GPU box only."""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manifold_gp_amd import _lib  # noqa: E402
from manifold_gp_amd.graph import bfs_order, build_tiles  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--s5-order", default="morton")
    a = ap.parse_args()
    a.gpus = 1
    dev = torch.device("cuda:0")
    wl = bench.build_workload(a, dev, 0, 1)
    g, lap = wl["graph"], wl["lap"]
    sym = lap._symmetric_twin()
    v = torch.rand(g.n, 1, device=dev)
    B = bench.spmm_bytes(g.n, g.M)
    lib = _lib.lib()
    lib.mgp_spmm_set_group_hint(g.spmv_lanes)
    variants = {"gather": None}
    stats = {}
    for rows in (32, 64, 128):
        t = build_tiles(g.n, g.rowptr, g.col, g.nnz, tile_rows=rows)
        if t is None:
            continue
        variants["tile%d" % rows] = t
        d = (t["tile_ptr"][1:] - t["tile_ptr"][:-1]).float()
        stats["tile%d" % rows] = dict(total_cols=t["total_cols"], max_cols=t["max_cols"], max_entries=t["max_entries"],
                                      mean_cols=round(float(d.mean()), 1), reuse=round(g.nnz / max(t["total_cols"], 1), 2))
    # tiles over a breadth-first locality order (what KnnGraph picks for unordered inputs)
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    order = bfs_order(g.n, g.rowptr, g.col)
    torch.cuda.synchronize(); t_bfs = time.perf_counter() - t0
    tb = build_tiles(g.n, g.rowptr, g.col, g.nnz, tile_rows=64, order=order)
    if tb is not None:
        variants["tile64_bfs"] = tb
        d = (tb["tile_ptr"][1:] - tb["tile_ptr"][:-1]).float()
        stats["tile64_bfs"] = dict(total_cols=tb["total_cols"], max_cols=tb["max_cols"], mean_cols=round(float(d.mean()), 1),
                                   reuse=round(g.nnz / max(tb["total_cols"], 1), 2), bfs_order_ms=round(t_bfs * 1e3, 1))
    times = {k: [] for k in variants}
    ref = None
    for rnd in range(a.rounds):
        for name, t in variants.items():
            g.tiles = t
            lib.mgp_spmm_set_tile_mode(1 if t is not None else 0)
            sym.data.vals_t = g.tile_values(sym.data.vals)
            csr = sym.data.csr()
            out = torch.empty_like(v)
            st = _lib.stream()
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 10, None, st)
            ms = ctypes.c_float(0.0)
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), a.reps, ctypes.byref(ms), st)
            times[name].append(ms.value / a.reps * 1e3)
            if rnd == 0:
                if ref is None:
                    ref = out.clone()
                else:
                    stats.setdefault(name, {})["max_abs_diff_vs_gather"] = float((out - ref).abs().max())
    lib.mgp_spmm_set_tile_mode(1)
    res = []
    for name in variants:
        t = sorted(times[name])
        res.append(dict(kernel=name, us_median=round(t[len(t) // 2], 2), us_min=round(t[0], 2),
                        gbs=round(B / (t[len(t) // 2] * 1e-6) / 1e9, 1), **stats.get(name, {})))
    print(json.dumps(dict(workload=wl["name"], n=g.n, nnz=g.nnz, bytes=B, results=res), indent=1))


if __name__ == "__main__":
    main()
