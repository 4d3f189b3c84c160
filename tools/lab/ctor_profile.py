#!/usr/bin/env python
"""Lab: where the kernel constructor's time goes at C3 size (cProfile of the second construction, CUDA_LAUNCH_BLOCKING-free: host view)."""
import os, sys, time, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev); x = x.contiguous()
def ctor():
    k = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
    torch.cuda.synchronize()
    return k
ctor()
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); ctor(); t1 = time.perf_counter(); pr.disable()
print("ctor ms %.1f" % ((t1 - t0) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
