#!/usr/bin/env python
"""Lab: a cold pipeline at C3 size, stage by stage: kernel constructor (k-NN + graph), first eval() (Laplacian, wide relabelling =
chain order + relabelled CSR + tile image, eigensolve), second eval()."""
import os, sys, time, warnings, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import graph as G
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev); x = x.contiguous()
def T(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t0) * 1e3
for rep in range(2):
    kern, t_ctor = T(lambda: mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev))
    kern.initialize(graphbandwidth=0.25, lengthscale=3.0)
    g = kern.knn.knn_graph
    _, t_chain = T(lambda: G.chain_order(g.n, g.rowptr, g.col, g.d2))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kern.warm_start = False
        _, t_e1 = T(lambda: kern.eval())
        _, t_e2 = T(lambda: kern.eval())
    print("rep %d: ctor (k-NN + graph) %.1f ms, chain order alone %.1f ms, first eval %.1f ms, second eval %.1f ms" % (rep, t_ctor, t_chain, t_e1, t_e2), flush=True)
