#!/usr/bin/env python
"""Per-tile entry / dictionary size distribution of the bench graph's row tiles."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
class A: workload, nodes, gpus, s5_order = sys.argv[1] if len(sys.argv) > 1 else "c3", 0, 1, "morton"
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g = wl["graph"]
t = g.tiles
rows = t["rows"]
rp = (t.get("tile_rowptr") if t.get("tile_rowptr") is not None else g.rowptr).cpu().numpy().astype(np.int64)
n = g.n
b = np.minimum(np.arange(0, n + rows, rows), n)
ent = rp[b[1:]] - rp[b[:-1]]
ent = ent[ent >= 0]
D = np.diff(t["tile_ptr"].cpu().numpy().astype(np.int64))
for name, a in (("entries", ent), ("dict", D)):
    print(name, "tiles", len(a), "mean %.0f" % a.mean(), "p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(a, [50, 90, 99, 100])),
          "max/mean %.2f" % (a.max() / a.mean()))
print("rows", rows, "max_cols", t["max_cols"], "max_entries", t["max_entries"], "reuse %.1f" % t["reuse"])
