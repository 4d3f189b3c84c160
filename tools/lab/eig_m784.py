#!/usr/bin/env python
"""Lab: eval() of the conditioned 60k workload (swiss roll in R^784) with the eigensolver's phase timing (MGP_EIG_TIMING=1)."""
import os, sys, time, warnings, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from tools import synth
dev = torch.device("cuda:0")
x_np, y_np, _ = synth.manifold_784(60600)
rng = np.random.default_rng(11); perm = rng.permutation(60600); tr = np.sort(perm[600:])
x, y = torch.from_numpy(x_np[tr]).to(dev), torch.from_numpy(y_np[tr]).to(dev)
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100, bump_scale=3.0, bump_decay=0.01).to(dev)
kern.initialize(graphbandwidth=0.3, lengthscale=3.0)
model = RiemannGP(x, y, GaussianLikelihood(1e-2).to(dev), ScaleKernel(kern, 1.0).to(dev)).to(dev)
kern.warm_start = False
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); model.eval(); torch.cuda.synchronize()
        print("eval ms %.2f info %s" % ((time.perf_counter() - t0) * 1e3, kern.eigen_info), flush=True)
