#!/usr/bin/env python
"""Lab: does storing the points in a PCA-Morton order speed up the C3 SpMV? (GPU box only)"""
import ctypes, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
from manifold_gp_amd.graph import LaplacianData
from tools import synth

def time_spmv(data, n):
    lib = _lib.lib(); dev = data.graph.device
    v = torch.rand(n, 1, device=dev); out = torch.empty_like(v)
    lib.mgp_spmm_set_group_hint(8); lib.mgp_spmm_set_rows_in_flight(2)
    csr = data.csr(); st = _lib.stream()
    lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 20, None, st)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 100, None, st)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    return best

dev = torch.device("cuda:0")
x_np, y_np = synth.rmnist_like(600, 100, seed=1337)
x = torch.from_numpy(x_np).to(dev)
def build(xx):
    knn = mgp.utils.NearestNeighbors(xx); knn.graph(50); g = knn.knn_graph
    idx = g.edge_index
    d = (idx[0] - idx[1]).abs().float()
    return g, LaplacianData(g, 0.255, True), [float((d <= w).float().mean()) for w in (64, 256, 1024, 4096)]
g0, d0, loc0 = build(x)
print("original order: spmv %.2f us, frac of edges within 64/256/1024/4096: %s" % (time_spmv(d0, g0.n), loc0))
# PCA (torch on device, lab only) -> Morton code of the top-3 scores
xc = x - x.mean(0, keepdim=True)
U, S, V = torch.pca_lowrank(xc, q=8, center=False)
for dims in (1, 2, 3):
    sc = (xc @ V[:, :max(dims, 1)]).cpu().numpy()
    if dims == 1:
        perm = np.argsort(sc[:, 0], kind="stable")
    else:
        pad = np.concatenate([sc, np.zeros((len(sc), 3 - sc.shape[1]))], 1) if sc.shape[1] < 3 else sc
        perm = synth.morton_order(pad.astype(np.float32), bits=12)
    xp = x[torch.from_numpy(perm).to(dev)].contiguous()
    g1, d1, loc1 = build(xp)
    print("PCA-%dD order: spmv %.2f us, within 64/256/1024/4096: %s" % (dims, time_spmv(d1, g1.n), loc1))
