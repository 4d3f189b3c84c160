#!/usr/bin/env python
"""Lab: random shapes through the k-NN candidate filter (mode 2) against the slab pipeline (mode 0), lists compared bit for bit.
Usage: knn_filter_fuzz.py [cases] [seed]"""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
L = _lib.lib()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for c in range(cases):
    N = int(rng.choice([1024, 1025, 1500, 3000, 4097, 9000, 20000, 33000]))
    d = int(rng.choice([32, 33, 48, 64, 100, 257]))
    k = int(rng.choice([1, 2, 7, 16, 50, 64, 100, 128, 200]))
    k = min(k, N)
    selfq = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        x = rng.normal(size=(N, d))
    elif kind == 1:
        cen = rng.normal(size=(30, d)) * 4
        x = cen[rng.integers(0, 30, N)] + 0.05 * rng.normal(size=(N, d))
    else:
        x = rng.normal(size=(N, d)); x[: N // 3] = x[0]          # a third of the points are one point
    x = torch.from_numpy(x.astype(np.float32)).to(dev)
    n = N if selfq else int(rng.choice([1024, 1300, 2500]))
    q = x if selfq else torch.from_numpy(rng.normal(size=(n, d)).astype(np.float32)).to(dev)
    nn = mgp.utils.NearestNeighbors(x)
    try:
        L.mgp_knn_set_filter(0); D0, I0 = nn.search(q, k); s0 = dict(nn.last_stats)
        L.mgp_knn_set_filter(2); D1, I1 = nn.search(q, k); s1 = dict(nn.last_stats)
    finally:
        L.mgp_knn_set_filter(1)
    ok = bool(torch.equal(I0, I1)) and bool(torch.equal(D0, D1))
    bad += 0 if ok else 1
    print("%3d N=%6d n=%6d d=%3d k=%3d kind=%d self=%d  %s  filter: failover %d wide %d cand %d" % (
        c, N, n, d, k, kind, selfq, "ok" if ok else "MISMATCH", s1["filter_failover_rows"], s1["rows_redone_wide"], s1["candidates"]), flush=True)
print("mismatches:", bad)
