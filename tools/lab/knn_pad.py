#!/usr/bin/env python
"""Lab: first-attempt candidate count K' of the filtered search (MGP_KNN_CAND, lab build -DMGP_KNN_PAD_LAB): rows redone wide and
search time on three 60k x 784 data sets.  Usage: MGP_LAB_LIB=tools/lab/_kb_padlab/libmgp_hip.so python tools/lab/knn_pad.py 75 64 60"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
from tools import synth
dev = torch.device("cuda:0")
sets = {}
x, _ = synth.rmnist_like(600, 100, seed=1337, device=dev); sets["rmnist_like"] = x.contiguous()
xm, _, _ = synth.manifold_784(60000)
sets["manifold_784"] = torch.as_tensor(xm, dtype=torch.float32).to(dev).contiguous()
g = torch.Generator(device="cpu").manual_seed(3)
sets["gaussian_784"] = torch.randn(60000, 784, generator=g).to(dev)
sets["gaussian_64"] = torch.randn(60000, 64, generator=g).to(dev)
for name, x in sets.items():
    ref = None
    for cand in [int(a) for a in sys.argv[1:]]:
        os.environ["MGP_KNN_CAND"] = str(cand)
        knn = mgp.utils.NearestNeighbors(x)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); D, I = knn.search(x, 50); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        same = "" if ref is None else " identical=%s" % bool(torch.equal(ref, I))
        ref = I if ref is None else ref
        print("%-14s K'=%3d  ms %6.2f  wide %6d  failover %5d%s" % (name, cand, min(ts), knn.last_stats["rows_redone_wide"], knn.last_stats["filter_failover_rows"], same), flush=True)
