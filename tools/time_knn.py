#!/usr/bin/env python
"""Warm timing of the k-NN search and the graph build at C3 size (60k x 784, k = 50)."""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from tools import synth
dev = torch.device("cuda:0")
bases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
x, y = synth.rmnist_like(bases, 100, seed=1337, device=dev)
x = x.contiguous()
def timed(fn, reps=3):
    out = None; ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return out, ts
knn = mgp.utils.NearestNeighbors(x)
(D, I), ts = timed(lambda: knn.search(x, 50))
print("search ms", [round(t, 1) for t in ts], "stats", getattr(knn, "last_stats", None))
_, ts = timed(lambda: knn.graph(50))
print("graph(50) (search + symmetrise + CSR + tiles) ms", [round(t, 1) for t in ts])
from manifold_gp_amd import _lib
_lib.lib().mgp_knn_set_symmetric(0)
(Ds, Is), ts = timed(lambda: knn.search(x, 50))
print("every tile (symmetric mode off): search ms", [round(t, 1) for t in ts], "stats", knn.last_stats)
print("identical to the upper-triangle search", bool(torch.equal(I, Is)), bool(torch.equal(D, Ds)))
_lib.lib().mgp_knn_set_symmetric(1)
_lib.lib().mgp_knn_set_mfma(0)
(D0, I0), ts = timed(lambda: knn.search(x, 50))
print("direct tiles: search ms", [round(t, 1) for t in ts], "stats", knn.last_stats)
print("identical", bool(torch.equal(I, I0)), bool(torch.equal(D, D0)))
_lib.lib().mgp_knn_set_mfma(1)
