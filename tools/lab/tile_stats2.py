"""Lab: per-tile dictionary / entry / row-length distributions for the C3 bench graph and the C3-size manifold graph,
for 64-, 32- and 16-row tiles."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import manifold_gp_amd as mgp
from manifold_gp_amd.graph import build_tiles
from tools import synth
dev = torch.device("cuda:0")
def stats(name, g):
    rp = g.rowptr.cpu().numpy().astype(np.int64)
    rl = np.diff(rp)
    print(name, "n", g.n, "nnz", g.nnz, "row length mean %.1f p50 %d p90 %d p99 %d max %d" % ((rl.mean(),) + tuple(np.percentile(rl, [50, 90, 99, 100]))))
    for rows in (64, 32, 16):
        t = build_tiles(g.n, g.rowptr, g.col, g.nnz, tile_rows=rows)
        if t is None:
            print("  rows", rows, "no tiles"); continue
        D = np.diff(t["tile_ptr"].cpu().numpy().astype(np.int64))
        b = np.minimum(np.arange(0, g.n + rows, rows), g.n)
        ent = rp[b[1:]] - rp[b[:-1]]
        # max row length per tile vs mean (intra-tile imbalance)
        mx = np.array([rl[b[i]:b[i + 1]].max() for i in range(len(b) - 1)])
        mean_in = ent / np.maximum(b[1:] - b[:-1], 1)
        print("  rows %d: tiles %d; dict mean %.0f p50 %d p90 %d p99 %d max %d; entries mean %.0f p90 %d max %d; reuse %.2f; "
              "max row / mean row in tile: mean %.2f p90 %.2f" % ((rows, len(D), D.mean()) + tuple(np.percentile(D, [50, 90, 99, 100])) +
              (ent.mean(),) + tuple(np.percentile(ent, [90, 100])) + (t["reuse"], (mx / mean_in).mean(), np.percentile(mx / mean_in, 90))))
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
stats("C3 RMNIST-like", wl["graph"])
del wl
x_np, y_np, _ = synth.manifold_784(60000)
knn = mgp.utils.NearestNeighbors(torch.from_numpy(x_np).to(dev))
knn.graph(50)
stats("manifold_784 (as generated: random order)", knn.knn_graph)
print("  tiles chosen:", {k: v for k, v in knn.knn_graph.tiles.items() if k in ("rows", "max_cols", "max_entries", "reuse")}, "rowid" , knn.knn_graph.tiles.get("rowid") is not None)
rg = knn.knn_graph.relabelled() if knn.knn_graph.has_locality_order() else None
if rg is not None:
    class G: pass
    g2 = G(); g2.n, g2.nnz, g2.rowptr, g2.col = rg.n, rg.nnz, rg.rowptr, rg.col
    stats("manifold_784 relabelled (BFS order)", g2)
