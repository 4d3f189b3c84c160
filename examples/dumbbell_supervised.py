#!/usr/bin/env python
"""The reference's 1-D dumbbell walk-through (examples/1D_supervised_learning.ipynb, config C1 of
BASELINE.json) on the MI355X path, written against the reference's own module names.

    python examples/dumbbell_supervised.py          (needs an MI355X; data = tests/golden/dumbbell_k10_loop.npz,
                                                     generated from the reference's dumbbell dataset)

`install_as_manifold_gp()` registers this package under the names `manifold_gp.kernels / operators /
utils / models`, so the model code below is the notebook's, import for import.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manifold_gp_amd  # noqa: E402

manifold_gp_amd.install_as_manifold_gp()
from manifold_gp.kernels import RiemannMaternKernel  # noqa: E402
from manifold_gp.models import GaussianLikelihood, RiemannGP, ScaleKernel  # noqa: E402


def main(quiet=False):
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
    x, y = torch.from_numpy(g["train_x"]).to(dev), torch.from_numpy(g["train_y"]).to(dev)
    xt, yt = torch.from_numpy(g["test_x"]).to(dev), torch.from_numpy(g["test_y"]).to(dev)

    kernel = RiemannMaternKernel(nu=2, x=x, nearest_neighbors=int(g["k"]), laplacian_normalization="randomwalk",
                                 num_modes=int(g["modes"]), bump_scale=float(g["bump"][0]),
                                 bump_decay=float(g["bump"][1])).to(dev)
    kernel.initialize(graphbandwidth=float(g["eps"]), lengthscale=float(g["kappa"]))
    likelihood = GaussianLikelihood(noise=1e-2).to(dev)
    model = RiemannGP(x, y, likelihood, ScaleKernel(kernel, outputscale=1.0).to(dev)).to(dev)

    # precision form (train_model.py:63-90 builds its loss from model.precision()): the data-fit term
    # y^T Q3 y with Q3 ~ (K + noise I)^-1, and the posterior mean at the graph nodes
    # K (K + noise I)^-1 y = (I + noise Q2)^-1 y as ONE sparse CG solve on the device
    from manifold_gp_amd.solvers import cg_solve
    with torch.no_grad():
        Q3 = model.precision()
        quad = float(torch.dot(y, Q3.matmul(y)))
        desc = model.precision(noise=False)._descriptor().with_(form=2, noise=float(likelihood.noise))
        node_mean, iters, resid = cg_solve(desc, y, tol=1e-6, stop_mode=1, max_iter=20000)
        solve_res = float((desc.apply(node_mean.view(-1, 1)).view(-1) - y).norm() / y.norm())

    model.eval()                                             # eigensolve of the graph Laplacian (HIP)
    model.posterior(xt, noisy_posterior=True)
    mean, std = model.posterior_mean, model.posterior_stddev
    rmse = float((mean - yt).square().mean().sqrt())
    out = dict(n_train=int(x.shape[0]), n_test=int(xt.shape[0]), modes=int(g["modes"]), quad_form=quad,
               precision_cg_iterations=int(iters), precision_solve_residual=solve_res,
               node_mean_rmse_vs_targets=float((node_mean.view(-1) - y).square().mean().sqrt()), test_rmse=rmse,
               mean_std=float(std.mean()), eigen_max_residual=float(max(kernel.eigen_residuals)))
    if not quiet:
        for k_, v_ in out.items():
            print("%-28s %s" % (k_, v_))
    return out


if __name__ == "__main__":
    main()
