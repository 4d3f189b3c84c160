#!/usr/bin/env python
"""Lab: how many DISTINCT columns do R consecutive rows of the C3 graph touch, against their R x nnz entries?
(The ratio bounds what a row-group-merged SpMM -- one X-row load feeding R rows' accumulators -- can save.)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import bench
class A: workload, nodes, gpus, s5_order = (sys.argv[1] if len(sys.argv) > 1 else "c3"), 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g = wl["lap"].data.graph if hasattr(wl["lap"].data, "graph") else wl["lap"].data
rowptr = g.rowptr.cpu().numpy().astype(np.int64); col = g.col.cpu().numpy()
n = len(rowptr) - 1
deg = np.diff(rowptr)
print("n", n, "entries", int(rowptr[-1]), "mean row", deg.mean(), "max", deg.max(), flush=True)
order = None
t = getattr(g, "tiles", None)
if t is not None and "rowid" in t: order = t["rowid"].cpu().numpy(); print("tiles follow a row order")
rows = order if order is not None else np.arange(n)
rng = np.random.default_rng(0)
for R in (2, 4, 8, 16, 64):
    starts = rng.choice(n // R, size=min(3000, n // R), replace=False) * R
    ent = 0; uni = 0; hist = np.zeros(R + 1)
    for s in starts:
        cs = np.concatenate([col[rowptr[r]:rowptr[r + 1]] for r in rows[s:s + R]])
        cs = cs[cs >= 0]
        u, c = np.unique(cs, return_counts=True)
        ent += len(cs); uni += len(u); hist += np.bincount(c, minlength=R + 1)[:R + 1]
    print("R %2d: entries/group %.1f distinct %.1f ratio %.3f  multiplicity histogram %s" % (R, ent / len(starts), uni / len(starts), uni / ent, np.round(hist[1:] / hist.sum(), 3)[:8]), flush=True)
