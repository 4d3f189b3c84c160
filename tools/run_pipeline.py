#!/usr/bin/env python
"""Stage-by-stage timing of the whole hot path at config C3/C4 size (GPU box only): k-NN, graph,
Laplacian build, eigensolve, features, out-of-sample features, kernel block, the two posterior-mean
solves (precision form with the sparse CG; covariance form with the low-rank CG) and a
semi-supervised Schur-complement matvec (10 % labelled).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp  # noqa: E402
from manifold_gp_amd.solvers import cg_solve, kernel_block, lowrank_cg, lowrank_solve  # noqa: E402
from tools import synth  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def main():
    bases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    dev = torch.device("cuda:0")
    res = {}
    x_np, y_np = synth.rmnist_like(bases, 101, seed=1337)        # 100 train + 1 test rotation per base
    n_all = x_np.shape[0]
    test_mask = np.zeros(n_all, bool)
    test_mask[100::101] = True
    x, y = torch.from_numpy(x_np[~test_mask]).to(dev), torch.from_numpy(y_np[~test_mask]).to(dev)
    xt, yt = torch.from_numpy(x_np[test_mask]).to(dev), torch.from_numpy(y_np[test_mask]).to(dev)
    hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
    res["n_train"], res["n_test"] = x.shape[0], xt.shape[0]

    kern, t = timed(lambda: mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50,
                                                            laplacian_normalization="randomwalk", num_modes=100,
                                                            bump_scale=3.0, bump_decay=0.01).to(dev))
    res["kernel_ctor_knn_graph_ms"] = round(t, 1)
    D1, _ = kern.knn.search(x[:20000], 2)
    eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
    kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
    _, t = timed(lambda: kern.laplacian().data)
    res["laplacian_build_ms"] = round(t, 3)
    _, t = timed(lambda: kern.eval())
    res["eval_eigensolve_ms"] = round(t, 1)
    res["eigen_max_residual"] = float(max(kern.eigen_residuals))
    Z, t = timed(lambda: kern.features(x))
    res["features_insample_ms"] = round(t, 3)
    Zt, t = timed(lambda: kern.features(xt))
    res["features_oos_ms(knn_search+fused_kernel)"] = round(t, 2)
    res["oos_in_support_frac"] = float((Zt.abs().sum(1) > 0).float().mean())
    K, t = timed(lambda: kernel_block(Zt, Z, hp["outputscale"]))
    res["kernel_block_mfma_ms"] = round(t, 3)
    res["kernel_block_tflops"] = round(2 * Zt.shape[0] * Z.shape[0] * Z.shape[1] / (t * 1e-3) / 1e12, 2)
    (alpha, its), t = timed(lambda: lowrank_cg(Z, y, hp["outputscale"], hp["noise"], tol=1e-6))
    res["posterior_covariance_form_lowrank_cg_ms"] = round(t, 2)
    res["lowrank_cg_iters"] = its
    alpha_w, t = timed(lambda: lowrank_solve(Z, y, hp["outputscale"], hp["noise"]))
    res["posterior_covariance_form_woodbury_ms"] = round(t, 2)
    res["woodbury_vs_cg_rel_diff"] = float((alpha_w - alpha).abs().max() / alpha.abs().max())
    mean_t = K @ alpha
    res["spectral_posterior_test_rmse"] = float((mean_t - yt).square().mean().sqrt())
    Q = kern.precision()
    desc = Q._descriptor().with_(scale=hp["outputscale"], form=2, noise=hp["noise"])
    (sol, its2, r2), t = timed(lambda: cg_solve(desc, y, tol=1e-6, stop_mode=1))
    res["posterior_precision_form_cg_ms(incl_plan)"] = round(t, 2)
    res["precision_cg_iters"] = its2
    # semi-supervised (C4): 10 % labelled, Schur complement matvec with a nested HIP CG
    torch.manual_seed(1337)
    mask = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
    mask[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
    S = mgp.operators.SchurComplementOperator(Q, mask)
    v = y[mask]
    with mgp.settings.cg_tolerance(1e-4), mgp.settings.cg_stop_mode(1), mgp.settings.max_cg_iterations(5000), \
            mgp.settings.cg_jacobi_preconditioner(True):
        (sv, t) = timed(lambda: S.matmul(v))
        w = torch.randn_like(v)
        sw = S.matmul(w)
    res["schur_matvec_ms(6k labelled, nested CG on 54k)"] = round(t, 2)
    res["schur_symmetry_rel"] = float((torch.dot(w, sv) - torch.dot(v, sw)).abs() / (sv.norm() * w.norm()))
    return res


if __name__ == "__main__":
    # twice in one process: the first pass pays code-object loading, workspace allocation and hipBLAS / rocPRIM start-up
    # in whatever stage touches them first; the second pass (fresh tensors, fresh kernel object) is what a training loop sees
    first = main()
    second = main()
    print(json.dumps({"first_pass": first, "second_pass": second}, indent=1))
