"""Prints the actual HIP-vs-golden errors behind the tolerances of tests/test_gpu_parity.py (run on the GPU box):
spectrum / feature Gram / out-of-sample Gram against the float64 goldens, posterior against the fp64 oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp  # noqa: E402

dev = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def main():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
    m = int(g["modes"])
    for tol in (1e-5, 1e-6, 1e-7):
        for norm in ("symmetric", "randomwalk"):
            p = norm + "_"
            kern = mgp.kernels.RiemannMaternKernel(nu=1, x=T(g["train_x"]), nearest_neighbors=int(g["k"]),
                                                   laplacian_normalization=norm, num_modes=m,
                                                   bump_scale=float(g["bump"][0]), bump_decay=float(g["bump"][1])).to(dev)
            kern.initialize(graphbandwidth=float(g["eps"]), lengthscale=float(g["kappa"]))
            kern.eigen_tol = tol
            kern.eval()
            ev = kern.eigval.cpu().numpy()
            ref = g[p + "evals_raw_f64"][:m]
            lam_max = float(np.abs(g[p + "diag"]).max()) * 2
            x = T(g["train_x"])
            Z = kern.features(x)
            gram = (Z[:64].double() @ Z[:64].double().t()).cpu().numpy()
            r64 = g[p + "features_gram_64_f64"]
            dg = kern(x, x, diag=True).cpu().numpy()
            Zt = kern.features(T(g["test_x"])).double().cpu().numpy()
            b = g[p + "oos_bump_f64"]
            sel = b > 0
            ext = (Zt / np.where(sel, b, 1)[:, None]) @ Z[:64].double().cpu().numpy().T
            ro = g[p + "oos_gram_f64"]
            print("tol %.0e %-10s evals abs err %.3e (/lam_max %.3e) resid max %.3e | gram %.3e diag rel %.3e | oos %.3e"
                  % (tol, norm, np.abs(ev[1:] - ref[1:]).max(), np.abs(ev[1:] - ref[1:]).max() / lam_max,
                     max(kern.eigen_residuals), np.abs(gram - r64).max() / np.abs(r64).max(),
                     np.abs(dg / g[p + "features_diag_f64"] - 1).max(), np.abs(ext[sel] - ro[sel]).max() / np.abs(ro).max()))
    # posterior
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from oracle.solvers import gp_posterior_lowrank
    for norm in ("symmetric", "randomwalk"):
        x, y = T(g["train_x"]), T(g["train_y"])
        s, noise = 0.7, 1e-2
        kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=int(g["k"]), laplacian_normalization=norm,
                                               num_modes=m, bump_scale=float(g["bump"][0]), bump_decay=float(g["bump"][1])).to(dev)
        kern.initialize(graphbandwidth=float(g["eps"]), lengthscale=float(g["kappa"]))
        model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
        model.eval()
        rng = np.random.default_rng(3)
        xt_np = g["train_x"][rng.choice(x.shape[0], 40, replace=False)] + rng.normal(scale=0.02, size=(40, x.shape[1])).astype(np.float32)
        xt = T(xt_np)
        model.posterior(xt)
        Z, Zs = kern.features(x).cpu().numpy(), kern.features(xt).cpu().numpy()
        mean_o, cov_o, _ = gp_posterior_lowrank(Z, g["train_y"], Zs, s, noise)
        print("posterior %-10s mean err/max %.3e  cov err/max %.3e" %
              (norm, np.abs(model.posterior_mean.cpu().numpy() - mean_o).max() / np.abs(mean_o).max(),
               np.abs(model.posterior_covar.cpu().numpy() - cov_o).max() / np.abs(cov_o).max()))


if __name__ == "__main__":
    main()
