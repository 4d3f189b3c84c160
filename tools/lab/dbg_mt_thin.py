import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import manifold_gp_amd as mgp
from manifold_gp_amd.graph import LaplacianData, MtPlan, build_tiles
rng = np.random.default_rng(5)
t = np.sort(rng.random(8192))
x = np.stack([np.cos(6.28 * t) * (1 + t), np.sin(6.28 * t) * (1 + t), 0.3 * np.sin(40 * t)], 1).astype(np.float32)
nn = mgp.utils.NearestNeighbors(torch.from_numpy(x).cuda()); nn.graph(4)
g = nn.knn_graph
d = LaplacianData(g, 0.1, True)
print("order", g.has_locality_order(), "nnz", g.nnz, "nz", int((d.vals != 0).sum()))
tl = build_tiles(g.n, g.rowptr, g.col, g.nnz, tile_rows=16)
print("tiles16", None if tl is None else (tl["max_cols"], tl["total_cols"]))
if tl is not None:
    D = (tl["tile_ptr"][1:] - tl["tile_ptr"][:-1]).long()
    S = (D + 15) // 16 * 4
    print("D mean", float(D.float().mean()), "steps", int(S.sum()), "fill", int((d.vals != 0).sum()) / (64 * int(S.sum())))
