#!/usr/bin/env python
"""Candidate filter of the matrix-core k-NN (mgp_knn_set_filter) against the key slab: identical lists on ragged shapes,
duplicates, clusters, out-of-sample chunks; timing at C3 size.  Usage: knn_filter_ab.py [bases]"""
import os, sys, time, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
from tools import synth
dev = torch.device("cuda:0")
L = _lib.lib()

def search(x, q, k, mode, index=True):
    L.mgp_knn_set_filter(mode)
    knn = mgp.utils.NearestNeighbors(x) if index else mgp.utils.NearestNeighbors(x)
    D, I = knn.search(q, k)
    torch.cuda.synchronize()
    st = dict(knn.last_stats)
    L.mgp_knn_set_filter(1)
    return D, I, st

def case(name, x, q, k):
    D0, I0, s0 = search(x, q, k, 0)
    D2, I2, s2 = search(x, q, k, 2)
    ok = bool(torch.equal(I0, I2)) and bool(torch.equal(D0, D2))
    print("%-44s N=%6d n=%6d d=%4d k=%3d identical=%s  slab %s | filter %s" % (name, x.shape[0], q.shape[0], x.shape[1], k, ok, s0, s2), flush=True)
    return ok

g = torch.Generator(device="cpu").manual_seed(7)
allok = True
x = torch.randn(3001, 64, generator=g).to(dev)
allok &= case("ragged gaussian self", x, x, 10)
x = torch.randn(8323, 96, generator=g).to(dev)
allok &= case("ragged gaussian self k=50", x, x, 50)
allok &= case("ragged gaussian self k=100 (stride 8)", x, x, 100)
q = torch.randn(1500, 96, generator=g).to(dev)
allok &= case("out-of-sample queries", x, q, 32)
xd = x.clone(); xd[1000:3000] = xd[0]            # 2001 duplicates of one point: lists overflow -> fail-over
allok &= case("2001 duplicates (overflow rows)", xd, xd, 20)
c = torch.randn(40, 128, generator=g) * 5
xc = (c[torch.randint(0, 40, (20000,), generator=g)] + 0.01 * torch.randn(20000, 128, generator=g)).to(dev)
allok &= case("tight clusters 20000 x 128", xc, xc, 50)
xf = (torch.randn(5000, 64, generator=g) * 1e-3 + 100.0).to(dev)
allok &= case("far from the origin (absolute bound fails)", xf, xf, 16)
print("ALL IDENTICAL" if allok else "MISMATCH", flush=True)

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
x, y = synth.rmnist_like(bases, 100, seed=1337, device=dev)
x = x.contiguous()
def timed(fn, reps=4):
    ts = []; out = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return out, ts
res = {}
for mode in (0, 1):
    L.mgp_knn_set_filter(mode)
    knn = mgp.utils.NearestNeighbors(x)
    (D, I), ts = timed(lambda: knn.search(x, 50))
    res[mode] = (D, I)
    print("mode %d: search ms %s stats %s" % (mode, [round(t, 2) for t in ts], knn.last_stats), flush=True)
    _lib.release_workspace("knn", dev)
print("60k identical:", bool(torch.equal(res[0][1], res[1][1])), bool(torch.equal(res[0][0], res[1][0])))
L.mgp_knn_set_filter(1)
