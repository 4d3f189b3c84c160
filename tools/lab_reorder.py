#!/usr/bin/env python
"""Lab: does a bandwidth-reducing node ordering (RCM) speed up the C == 1 SpMV?  GPU box only."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch
from scipy.sparse.csgraph import reverse_cuthill_mckee

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manifold_gp_amd as mgp  # noqa: E402
from manifold_gp_amd import _lib  # noqa: E402
from manifold_gp_amd.graph import KnnGraph  # noqa: E402


def time_op(op, v, reps=100):
    for _ in range(10):
        op._matmul(v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        op._matmul(v)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


class A:
    workload, nodes, gpus = "c3", 0, 1


def main():
    dev = torch.device("cuda:0")
    wl = bench.build_workload(A(), dev, 0, 1)
    g = wl["graph"]
    n = g.n
    idx = g.edge_index.cpu().numpy()
    val = g.edge_value.cpu().numpy()
    adj = sp.coo_matrix((np.ones(idx.shape[1] * 2), (np.r_[idx[0], idx[1]], np.r_[idx[1], idx[0]])), shape=(n, n)).tocsr()
    t0 = time.time()
    perm = reverse_cuthill_mckee(adj, symmetric_mode=True)
    t_rcm = time.time() - t0
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    r, c = inv[idx[0]], inv[idx[1]]
    lo, hi = np.minimum(r, c), np.maximum(r, c)
    bw0 = np.abs(idx[0] - idx[1])
    bw1 = hi - lo
    print("rcm %.2fs; |i-j| median %d -> %d, p90 %d -> %d, max %d -> %d" % (
        t_rcm, np.median(bw0), np.median(bw1), np.percentile(bw0, 90), np.percentile(bw1, 90), bw0.max(), bw1.max()))
    g2 = KnnGraph.from_coo(torch.from_numpy(np.stack([lo, hi])).to(dev), torch.from_numpy(val).to(dev), n)
    eps = torch.tensor([[wl["eps"]]], device=dev)
    ops = {"orig": mgp.operators.GraphLaplacianOperator(g.edge_value, g.edge_index, n, eps, "symmetric", graph=g),
           "rcm": mgp.operators.GraphLaplacianOperator(g2.edge_value, g2.edge_index, n, eps, "symmetric", graph=g2)}
    v = torch.rand(n, 1, device=dev)
    lib = _lib.lib()
    res = []
    for name, op in ops.items():
        for L in (0, 1):
            for G in (8, 16, 32, 64):
                for R in (1, 2, 4):
                    op.graph.spmv_lanes = G
                    lib.mgp_spmm_set_rows_in_flight(R)
                    lib.mgp_spmm_set_entry_layout(L)
                    res.append((round(time_op(op, v), 2), name, L, G, R))
    res.sort()
    for r in res[:12]:
        print(r)
    print("best orig:", [r for r in res if r[1] == "orig"][0])


if __name__ == "__main__":
    main()
