#!/usr/bin/env python
"""Config C5 end to end on one GPU: swiss roll in R^3 (random input order), N points, k = 64: k-NN + graph
(+ locality order), Laplacian, eigensolve, features, out-of-sample features, posterior means (spectral /
covariance form and sparse precision form).  Prints one JSON object.  GPU box only."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd.solvers import cg_solve, kernel_block, lowrank_solve
from tools import synth


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    modes = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = torch.device("cuda:0")
    x_np, y_np = synth.swiss_roll(n + 1000, seed=11, order="random")
    x, y = torch.from_numpy(x_np[:n]).to(dev), torch.from_numpy(y_np[:n]).to(dev)
    xt, yt = torch.from_numpy(x_np[n:]).to(dev), torch.from_numpy(y_np[n:]).to(dev)
    res = dict(n_train=n, n_test=1000, modes=modes)
    kern, t = timed(lambda: mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=64, laplacian_normalization="symmetric",
                                                            num_modes=modes, bump_scale=3.0, bump_decay=0.01).to(dev))
    res["knn_graph_tiles_ms"] = round(t, 1)
    g = kern.knn.knn_graph
    res["tiles"] = dict(reuse=round(g.tiles["reuse"], 2), locality_order=g.tiles.get("rowid") is not None)
    D1, _ = kern.knn.search(x[:20000], 2)
    eps, eps_min = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), 0.0)
    eps = 3.0 * eps_min
    kern.initialize(graphbandwidth=eps, lengthscale=1.0)
    _, t = timed(lambda: kern.laplacian().data)
    res["laplacian_build_ms"] = round(t, 2)
    _, t = timed(lambda: kern.eval())
    res["eval_eigensolve_ms"] = round(t, 1)
    res["eigen_max_residual"] = float(max(kern.eigen_residuals))
    Z, t = timed(lambda: kern.features(x))
    res["features_insample_ms"] = round(t, 2)
    Zt, t = timed(lambda: kern.features(xt))
    res["features_oos_ms"] = round(t, 2)
    alpha, t = timed(lambda: lowrank_solve(Z, y, 1.0, 0.01))
    res["posterior_covariance_form_woodbury_ms"] = round(t, 1)
    K, t = timed(lambda: kernel_block(Zt, Z, 1.0))
    res["kernel_block_ms"] = round(t, 2)
    res["spectral_test_rmse"] = float((K @ alpha - yt).square().mean().sqrt())
    desc = kern.precision()._descriptor().with_(scale=1.0, form=2, noise=0.01)
    (sol, its, r), t = timed(lambda: cg_solve(desc, y, tol=1e-6, stop_mode=1, max_iter=5000))
    res["posterior_precision_form_cg_ms"] = round(t, 1)
    res["precision_cg_iters"] = its
    res["precision_true_residual"] = float((desc.apply(sol.view(-1, 1)).view(-1) - y).norm() / y.norm())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
