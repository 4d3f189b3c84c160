"""Lab: C = 1 tile SpMV with the tiles of each XCD slice dispatched heaviest first (tiles over a row order that
permutes whole 64-row tiles) against the natural tile order."""
import os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import manifold_gp_amd as mgp
from manifold_gp_amd import graph as G
dev = torch.device("cuda:0")
wname = sys.argv[1] if len(sys.argv) > 1 else "c3"
wl = bench.build_workload(argparse.Namespace(workload=wname, nodes=0, s5_order="morton"), dev, 0, 1)
g = wl["graph"]
print("natural order: %.2f us" % (bench.time_spmv_kernel(wl) * 1e6), {k: v for k, v in g.tiles.items() if not torch.is_tensor(v)})
n = g.n
rp = g.rowptr.cpu().numpy().astype(np.int64)
nt = -(-n // 64)
ent = np.array([rp[min(n, (t + 1) * 64)] - rp[t * 64] for t in range(nt)])
# grid = nt blocks (one tile per block up to 4096 blocks)
tpb = -(-nt // 4096)
nb = -(-nt // tpb)
per, rem = nb // 8, nb % 8
perm_blocks = []
for x in range(8):
    start = x * per + min(x, rem)
    cnt = per + (1 if x < rem else 0)
    blocks = list(range(start, start + cnt))
    # block b holds tiles [b*tpb, (b+1)*tpb): weight = entries of its tiles
    w = [ent[b * tpb:(b + 1) * tpb].sum() for b in blocks]
    blocks = [b for _, b in sorted(zip(w, blocks), key=lambda p: -p[0])]
    perm_blocks += blocks
tiles_perm = [t for b in perm_blocks for t in range(b * tpb, min(nt, (b + 1) * tpb))]
# the last (ragged) tile must stay last in the order: swap it to the end
last = nt - 1
tiles_perm.remove(last); tiles_perm.append(last)
order = np.concatenate([np.arange(t * 64, min(n, (t + 1) * 64)) for t in tiles_perm]).astype(np.int32)
assert len(order) == n and len(set(order.tolist())) == n
order_t = torch.from_numpy(order).to(dev)
t2 = G.build_tiles(n, g.rowptr, g.col, g.nnz, order=order_t)
print("sorted tiles:", {k: v for k, v in t2.items() if not torch.is_tensor(v)})
g.tiles = t2
lap2 = mgp.operators.GraphLaplacianOperator(g.edge_value, g.edge_index, n, torch.tensor([[wl["eps"]]], device=dev), wl["norm"], graph=g)
wl2 = dict(wl); wl2["lap"] = lap2
v = torch.rand(n, 1, device=dev)
ref = wl["lap"]._matmul(v) if hasattr(wl["lap"], "_matmul") else None
print("heaviest-first order: %.2f us" % (bench.time_spmv_kernel(wl2) * 1e6))
out2 = lap2._matmul(v)
print("max |difference| of L v between the two tilings: %.3g (|L v| max %.3g)" % (float((ref - out2).abs().max()), float(ref.abs().max())))
