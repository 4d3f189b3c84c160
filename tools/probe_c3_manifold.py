"""C3-size workload with a conditioned spectrum (tools/synth.py::manifold_784): the HIP pipeline end to end against an
INDEPENDENT float64 eigendecomposition (scipy shift-invert eigsh = ARPACK + SuperLU on the oracle's CSR) followed by the
oracle's float64 features / posterior.  Prints the errors for each eigensolver tolerance given on the command line.
Run on the GPU box.  Usage: probe_c3_manifold.py [m] [tol ...]"""
import os, sys, time
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
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402

dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tols = [float(a) for a in sys.argv[2:]] or [1e-6]
n_all, k, nu = 60600, 50, 2
eps, kappa, s, noise, bump = 0.3, 3.0, 1.0, 1e-2, (3.0, 0.01)
t0 = time.time()
x_np, y_np, _ = synth.manifold_784(n_all)
print("data %.1fs" % (time.time() - t0), flush=True)
rng = np.random.default_rng(11)
perm = rng.permutation(n_all)
te, tr = np.sort(perm[:600]), np.sort(perm[600:])
x, y, xt = T(x_np[tr]), T(y_np[tr]), T(x_np[te])
n = x.shape[0]
kern = mgp.kernels.RiemannMaternKernel(nu=nu, x=x, nearest_neighbors=k, laplacian_normalization="randomwalk", num_modes=m,
                                       bump_scale=bump[0], bump_decay=bump[1]).to(dev)
kern.initialize(graphbandwidth=eps, lengthscale=kappa)
D, I = kern.knn.search(x, k)
print("eps_min rule %.4f; M %d" % (synth.bandwidth_rule(D[:, 1].cpu().numpy(), 0.0)[1], kern.knn.knn_graph.M), flush=True)
Dt, It = kern.knn.search(xt, k)
g = kern.knn.knn_graph
t0 = time.time()
lo = LaplacianOracle(g.edge_value.cpu().numpy(), g.edge_index.cpu().numpy(), n, eps, "randomwalk", True, dtype=np.float64)
L = laplacian_sym_csr(lo)
lmax = 2.0 * float(np.abs(lo.diag).max())
shift = 1e-3 * lmax
lu = spla.splu((L + shift * sp.identity(n)).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
               options=dict(SymmetricMode=True))
t1 = time.time()
w, U = spla.eigsh(L, k=m + 4, sigma=-shift, which="LM", OPinv=spla.LinearOperator((n, n), matvec=lu.solve, dtype=np.float64),
                  tol=1e-12)
o = np.argsort(w)
w, U = w[o], U[:, o]
R = L @ U - U * w[None, :]
print("checker: CSR+LU %.1fs eigsh %.1fs; residual max %.1e; lam[0,1,m-1,m] %s; lmax bound %.3f; gap %.3e = %.2e lmax"
      % (t1 - t0, time.time() - t1, np.linalg.norm(R, axis=0).max(), w[[0, 1, m - 1, m]], lmax, w[m] - w[m - 1],
         (w[m] - w[m - 1]) / lmax), flush=True)
gap = float(w[m] - w[m - 1])
lam = w[:m].copy()
lam[0] = 0.0
Phi = U[:, :m] * (lo.degree ** -0.5)[:, None]
Phi /= np.linalg.norm(Phi, axis=0, keepdims=True)
Z64 = osp.features_insample(lam, Phi, nu, kappa)
Zt64 = osp.features_oos(lo, lam, Phi, nu, kappa, Dt.double().cpu().numpy(), It.cpu().numpy(), bump[0], bump[1])
mean_o, cov_o, alpha_o = gp_posterior_lowrank(Z64, y_np[tr], Zt64, s, noise)
print("oracle posterior: mean rms %.3f, test rmse vs y %.3f, support %.2f" % (np.sqrt((mean_o ** 2).mean()),
      np.sqrt(((mean_o - y_np[te]) ** 2).mean()), (np.abs(Zt64).sum(1) > 0).mean()), flush=True)
rows = rng.choice(n, 256, replace=False)
for tol in tols:
    kern.eigen_tol = tol
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    torch.cuda.synchronize(); t0 = time.time()
    model.eval()
    torch.cuda.synchronize(); t_eval = time.time() - t0
    model.posterior(xt)
    from manifold_gp_amd.solvers import lanczos_smallest as _ls
    res = max(kern.eigen_residuals)
    mean, cov = model.posterior_mean.double().cpu().numpy(), model.posterior_covar.double().cpu().numpy()
    Z = kern.features(x).double().cpu().numpy()
    Zt = kern.features(xt).double().cpu().numpy()
    e = dict(evals=float(np.abs(kern.eigval.cpu().numpy()[1:] - lam[1:]).max() / lmax),
             kernel=float(np.abs(Z[rows] @ Z.T - Z64[rows] @ Z64.T).max() / np.abs(Z64[rows] @ Z64.T).max()),
             cross=float(np.abs(Zt @ Z.T - Zt64 @ Z64.T).max() / np.abs(Zt64 @ Z64.T).max()),
             mean=float(np.abs(mean - mean_o).max() / np.abs(mean_o).max()),
             var=float(np.abs(np.diag(cov) - np.diag(cov_o)).max() / np.abs(np.diag(cov_o)).max()),
             cov=float(np.abs(cov - cov_o).max() / np.abs(cov_o).max()))
    print("tol %.0e: eval %.0f ms, info %s, resid %.2e (%.1e lmax), |R|sqrt(m)/gap %.2e; %s"
          % (tol, 1e3 * t_eval, _ls.last_info, res, res / lmax, res * np.sqrt(m) / gap,
             " ".join("%s %.2e" % kv for kv in e.items())), flush=True)
