"""Independent float64 look at the spectral stage at C3 size (run on the GPU box; the CPU part is the oracle's CSR):
fp64 Rayleigh-Ritz of the HIP solver's whole block, residuals, gap behind the kept modes, Davis-Kahan bound, and the
kernel entries / posterior of the HIP pipeline against the same quantities computed in float64 from the refined pairs.
Usage: probe_c3_spectral.py [tol ...]"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp  # noqa: E402
from tools import synth  # noqa: E402
from oracle.laplacian import LaplacianOracle  # noqa: E402
from oracle.sparse import laplacian_sym_csr  # noqa: E402
from oracle import spectral as osp  # noqa: E402
from oracle.solvers import gp_posterior_lowrank  # noqa: E402
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel  # noqa: E402
from manifold_gp_amd.solvers import kernel_block  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
x_all, y_all = synth.rmnist_like(606, 100, seed=1337, device=dev)
rng = np.random.default_rng(1337)
perm = rng.permutation(x_all.shape[0])
test_rows, train_rows = np.sort(perm[:600]), np.sort(perm[600:])
x, y = x_all[T(train_rows)].contiguous(), y_all[T(train_rows)].contiguous()
xt = x_all[T(test_rows)].contiguous()
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100,
                                       bump_scale=3.0, bump_decay=0.01).to(dev)
D, I = kern.knn.search(x, 50)
eps, eps_min = synth.bandwidth_rule(D[:, 1].cpu().numpy(), hp["graphbandwidth"])
kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
n, m = x.shape[0], 100
graph = kern.knn.knn_graph
t0 = time.time()
lo = LaplacianOracle(graph.edge_value.cpu().numpy(), graph.edge_index.cpu().numpy(), n, eps, "randomwalk", True, dtype=np.float64)
L = laplacian_sym_csr(lo)
print("oracle CSR %.1fs, eps %.4f" % (time.time() - t0, eps), flush=True)
s, noise = hp["outputscale"], hp["noise"]
Dt, It = kern.knn.search(xt, 50)
Dt_np, It_np = Dt.cpu().numpy().astype(np.float64), It.cpu().numpy()
rows = np.random.default_rng(5).choice(n, 256, replace=False)
for tol in [float(a) for a in sys.argv[1:]] or [1e-5]:
    kern.eigen_tol, kern.keep_eigen_block = tol, True
    torch.cuda.synchronize(); t0 = time.time()
    kern.eval()
    torch.cuda.synchronize(); t_eig = time.time() - t0
    blk = kern.eigen_block
    V = blk["evecs"].double().cpu().numpy()
    b = V.shape[1]
    t0 = time.time()
    Q, _ = np.linalg.qr(V)
    LQ = L @ Q
    H = Q.T @ LQ
    th, S = np.linalg.eigh(0.5 * (H + H.T))
    X = Q @ S
    R = LQ @ S - X * th[None, :]
    rn = np.linalg.norm(R, axis=0)
    gap = th[m] - th[m - 1]
    print("tol %.0e: eigensolve %.0f ms, info %s, HIP resid max %.2e | fp64 RR %.1fs: theta[98:103] %s gap %.3e, fp64 resid max(first m) %.2e "
          "||R_m||_F %.2e -> Davis-Kahan sin(theta) <= %.2e ; HIP evals err vs fp64 RR %.2e"
          % (tol, t_eig * 1e3, mgp.solvers.lanczos_smallest.last_info, max(kern.eigen_residuals), time.time() - t0,
             np.array2string(th[m - 2:m + 3], precision=6), gap, rn[:m].max(), np.linalg.norm(R[:, :m]), np.linalg.norm(R[:, :m]) / gap,
             np.abs(blk["evals"].cpu().numpy()[1:m] - th[1:m]).max()), flush=True)
    # float64 pipeline from the refined pairs (riemann_kernel.py:126-149)
    lam = th[:m].copy(); lam[0] = 0.0
    Phi = X[:, :m] * (lo.degree ** -0.5)[:, None]
    Phi /= np.linalg.norm(Phi, axis=0, keepdims=True)
    Z64 = osp.features_insample(lam, Phi, 2, hp["lengthscale"])
    Zt64 = osp.features_oos(lo, lam, Phi, 2, hp["lengthscale"], Dt_np, It_np, 3.0, 0.01)
    Z, Zt = kern.features(x), kern.features(xt)
    e_lam = np.abs(kern.eigval.cpu().numpy()[1:] - lam[1:]).max()
    # kernel entries: 256 x 60000 in-sample block, 600 x 60000 cross block
    K1 = kernel_block(Z[T(rows)].contiguous(), Z, s).double().cpu().numpy()
    K1r = s * (Z64[rows] @ Z64.T)
    K2 = kernel_block(Zt, Z, s).double().cpu().numpy()
    K2r = s * (Zt64 @ Z64.T)
    e_k1, e_k2 = np.abs(K1 - K1r).max() / np.abs(K1r).max(), np.abs(K2 - K2r).max() / np.abs(K2r).max()
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    model.eval = lambda: None   # (kern already evaluated with this tolerance)
    model._cache = None
    model.posterior(xt)
    mean_o, cov_o, alpha_o = gp_posterior_lowrank(Z64, y.cpu().numpy(), Zt64, s, noise)
    mean, cov = model.posterior_mean.double().cpu().numpy(), model.posterior_covar.double().cpu().numpy()
    print("   eigval err %.2e | kernel entries 256x60000 %.2e, 600x60000 %.2e | posterior mean %.2e var %.2e cov %.2e (rel. to max)"
          % (e_lam, e_k1, e_k2, np.abs(mean - mean_o).max() / np.abs(mean_o).max(),
             np.abs(np.diag(cov) - np.diag(cov_o)).max() / np.abs(np.diag(cov_o)).max(), np.abs(cov - cov_o).max() / np.abs(cov_o).max()), flush=True)
