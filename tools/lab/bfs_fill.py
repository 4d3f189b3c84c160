"""Lab: would a breadth-first relabelling of the C3 bench graph shrink the 16-row tiles' column lists (= the matrix-core SpMM's
steps and X gathers)?  Dictionary sizes of 16- / 64-row tiles in the given order and in BFS order."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from manifold_gp_amd.graph import build_tiles, bfs_order
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g = wl["graph"]
order = bfs_order(g.n, g.rowptr, g.col)
for rows in (16, 64):
    for name, o in (("given order", None), ("BFS order", order)):
        t = build_tiles(g.n, g.rowptr, g.col, g.nnz, tile_rows=rows, order=o)
        D = np.diff(t["tile_ptr"].cpu().numpy().astype(np.int64))
        steps = int((np.ceil(D / 16) * 4).sum()) if rows == 16 else 0
        print("rows %2d %-12s: dict mean %.0f p50 %d p90 %d max %d  reuse %.2f  steps(16-row) %d" % (rows, name, D.mean(), np.percentile(D, 50), np.percentile(D, 90), D.max(), t["reuse"], steps), flush=True)
