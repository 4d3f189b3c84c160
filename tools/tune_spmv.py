#!/usr/bin/env python
"""Time the C == 1 gather (fallback) SpMV for every (lanes per row, rows in flight) shape on the bench graph
(interleaved rounds in one process, cdna_hip_programming.md rule 24).  GPU box only."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from manifold_gp_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--s5-order", default="morton")
    a = ap.parse_args()
    a.gpus = 1
    dev = torch.device("cuda:0")
    wl = bench.build_workload(a, dev, 0, 1)
    g, lap = wl["graph"], wl["lap"]
    sym = lap._symmetric_twin()
    v = torch.rand(g.n, 1, device=dev)
    B = bench.spmm_bytes(g.n, g.M)
    shapes = [(0, G, R) for G in (8, 16, 32, 64) for R in (1, 2, 4, 8) if R <= G]
    if a.quick:
        shapes = [(0, G, R) for (G, R) in ((8, 1), (8, 2), (8, 4), (16, 2), (16, 4))]
    lib_ = _lib.lib()
    lib_.mgp_spmm_set_tile_mode(0)          # this tool tunes the gather (fallback) kernel
    times = {s: [] for s in shapes}
    lib = _lib.lib()
    for rnd in range(a.rounds):
        for (L, G, R) in shapes:
            g.spmv_lanes = G
            lib.mgp_spmm_set_rows_in_flight(R)
            import ctypes
            lib.mgp_spmm_set_group_hint(G)
            csr = sym.data.csr()
            out = torch.empty_like(v)
            st = _lib.stream()
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 10, None, st)
            ms = ctypes.c_float(0.0)
            lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), a.reps, ctypes.byref(ms), st)
            times[(L, G, R)].append(ms.value / a.reps * 1e3)
    out = []
    for s in shapes:
        t = sorted(times[s])
        out.append(dict(layout=s[0], lanes=s[1], rows=s[2], us_median=round(t[len(t) // 2], 2), us_min=round(t[0], 2),
                        gbs=round(B / (t[len(t) // 2] * 1e-6) / 1e9, 1)))
    out.sort(key=lambda d: d["us_median"])
    print(json.dumps(dict(workload=wl["name"], bytes=B, results=out), indent=1))


if __name__ == "__main__":
    main()
