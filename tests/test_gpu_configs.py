"""GPU parity at the sizes of BASELINE.json's configurations (C2, C3, C4, C5).

tests/test_gpu_parity.py holds the goldens (C1) and the edge cases; this file runs every other configuration at
its full size through the package classes (-> ctypes -> C-ABI) and checks it against the CPU oracle evaluated on
the same inputs: the C oracle for sampled k-NN rows (bit-exact), oracle/laplacian.py + oracle/sparse.py in
float64 for products / solves / eigenvalues, oracle/solvers.py for the GP posterior.  Tolerances are written at
each assert; the north-star bar is 1e-4 relative for eigenvalues, kernel entries and posterior mean / variance.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def mgp():
    import manifold_gp_amd
    from manifold_gp_amd import _lib
    _lib.lib()
    return manifold_gp_amd


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _hp(name):
    with open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")) as fh:
        return json.load(fh)[name]


def _oracle_lap(graph, eps, norm, self_loops=True):
    from oracle.laplacian import LaplacianOracle
    return LaplacianOracle(graph.edge_value.cpu().numpy(), graph.edge_index.cpu().numpy(), graph.n, eps, norm, self_loops,
                           dtype=np.float64)


def _knn_rows_bit_exact(x_np, D, I, k, rows):
    from oracle import knn as oknn
    Dr, Ir = oknn.knn_search(x_np, x_np[rows], k)
    assert np.array_equal(I[rows].cpu().numpy(), Ir)
    assert np.array_equal(D[rows].cpu().numpy(), Dr)


# ============================================================================= C2
@pytest.mark.parametrize("eps_rule", ["reference_default", "data_scaled"])
def test_c2_dumbbell_10k_spmv_and_eigensolve(mgp, dev, eps_rule):
    """C2 (benchmark/bench_sparse_laplacian.py:37-72 shape): dumbbell resampled to N = 10 000, k = 50, symmetric,
    nu = 1, 100 modes, v = rand(N) with seed 1337.  k-NN rows bit-exact; `mv` against the float64 oracle; the 100
    smallest eigenvalues against a dense float64 eigvalsh of the oracle's L_sym (what riemann_kernel.py:121-125 does,
    in double); residuals and orthogonality of the vectors; the Lanczos branch of diagonalization()."""
    from manifold_gp_amd.solvers import lanczos_smallest
    from oracle.sparse import SparsePrecision, laplacian_sym_csr
    from tools import synth
    n, k, m = 10000, 50, 100
    x_np, y_np, _ = synth.dumbbell_resampled(n)
    x = T(x_np, dev)
    knn = mgp.utils.NearestNeighbors(x)
    D, I = knn.search(x, k)
    _knn_rows_bit_exact(x_np, D, I, k, np.random.default_rng(0).choice(n, 400, replace=False))
    idx, val = knn.graph(k)
    graph = knn.knn_graph
    assert 0.5 * n * (k - 1) <= graph.M <= n * (k - 1)
    if eps_rule == "reference_default":
        eps = float(np.log(2.0))                                   # softplus(0): riemann_kernel.py:53
    else:
        eps = synth.bandwidth_rule(D[:, 1].cpu().numpy(), 0.0)[0]  # notebooks' eps_min rule
    op = mgp.operators.GraphLaplacianOperator(val, idx, n, torch.tensor([[eps]], device=dev), "symmetric", graph=graph)
    lo = _oracle_lap(graph, eps, "symmetric")
    # ---- mv (bench_sparse_laplacian.py:15-19,63-64)
    torch.manual_seed(1337)
    v = torch.rand(n)
    ref = laplacian_sym_csr(lo) @ v.double().numpy()
    out = op.matmul(v.to(dev).view(-1, 1)).squeeze(-1).cpu().numpy()
    lmax = 2.0 * float(np.abs(lo.diag).max())
    # fp32 round-off scale of L: its entries are differences of O(1) terms divided by eps^2 -- diag = (1 - Dt^-2 / D) / eps^2
    # (graph_laplacian_operator.py:94) -- so a few ulps of 1 / eps^2; with the data-scaled bandwidth every weight is
    # <= 1e-4, |L| ~ 2e-4 / eps^2 and that cancellation (the reference's own, it computes in fp32) dominates |L| ulps
    ulp = max(2e-6 * lmax, 4e-7 / eps ** 2)
    assert np.abs(out - ref).max() < ulp, (np.abs(out - ref).max(), ulp)
    # ---- eigen (bench_sparse_laplacian.py:30-34; riemann_kernel.py:121-125 in float64)
    w = torch.linalg.eigvalsh(T(laplacian_sym_csr(lo).toarray(), dev))[:m + 4].cpu().numpy()      # float64, 800 MB
    evals, evecs, resid = lanczos_smallest(op.data, m, tol=1e-6)
    ev = evals.cpu().numpy().astype(np.float64)
    assert np.abs(ev - w[:m]).max() < ulp, (np.abs(ev - w[:m]).max(), ulp)           # |lambda(L + dL) - lambda(L)| <= |dL|
    assert max(resid) <= 2e-6 * lmax
    V = evecs.double()
    assert float((V.t() @ V - torch.eye(m, device=dev, dtype=torch.float64)).abs().max()) < 5e-5
    R = laplacian_sym_csr(lo) @ V.cpu().numpy() - V.cpu().numpy() * ev[None, :]       # residuals re-evaluated by the oracle
    assert np.linalg.norm(R, axis=0).max() <= 2 * ulp + 2e-6 * lmax
    # ---- diagonalization(): N > max_cholesky_size -> the iterative branch (graph_laplacian_operator.py:132-144)
    ev2, U2 = op.diagonalization(num_modes=m)
    assert float(ev2[0]) == 0.0 and U2.shape == (n, m)
    assert np.abs(ev2.cpu().numpy()[1:] - w[1:m]).max() < 1e-4 * lmax + ulp
    # ---- precision (nu = 1) product and a CG solve of the bench's operator against the float64 oracle
    kappa = 0.7
    Q = mgp.operators.PrecisionMaternOperator(op, 1, torch.tensor([[kappa]], device=dev))
    sq = SparsePrecision(lo, 1, kappa)
    yv = T(y_np, dev)
    refq = sq.matmul(y_np.astype(np.float64))
    assert np.abs(Q.matmul(yv).cpu().numpy() - refq).max() < 2 * ulp * float(np.abs(y_np).max())
    if eps_rule == "reference_default":
        # (with the data-scaled bandwidth the fp32 Laplacian is the fp64 one perturbed by ~1e-4 |L| -- see `ulp` -- and
        # cond(Q) ~ 6e3: a solve with it is a solve with another matrix; the products above are what can be compared)
        with mgp.settings.cg_tolerance(1e-7), mgp.settings.cg_stop_mode(1), mgp.settings.max_cg_iterations(20000):
            sol = Q.solve(yv).cpu().numpy()
        refs = sq.solve(y_np.astype(np.float64))
        assert np.abs(sol - refs).max() < 1e-4 * np.abs(refs).max(), np.abs(sol - refs).max() / np.abs(refs).max()


def test_c2_dumbbell_10k_posterior_vs_dense_float64_pipeline(mgp, dev):
    """The north-star target end to end at a size where an INDEPENDENT float64 reference exists: the dumbbell resampled
    to 10 000 points, 300 of them held out.  Checker = the reference's pipeline restated in float64 and evaluated densely
    (oracle Laplacian -> dense float64 eigh of the 9 700 x 9 700 L_sym, riemann_kernel.py:121-128 -> features :134-147 ->
    Woodbury posterior, riemann_gp.py:45-75); nothing of the HIP path is fed to it except the k-NN lists, which are
    bit-exact against the C oracle (sampled rows).  HIP: k-NN -> graph -> Laplacian -> block eigensolve -> features ->
    posterior.  99 modes: on a closed curve the eigenvalues come in near-pairs (1,2), (3,4), ...; 99 keeps whole pairs,
    the gap behind the kept block is printed and asserted against the residual (Davis-Kahan)."""
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from oracle import spectral as osp
    from oracle.solvers import gp_posterior_lowrank
    from oracle.sparse import laplacian_sym_csr
    from tools import synth
    n_all, k, m, nu = 10000, 50, 99, 2
    eps, kappa, s, noise, bump = 0.02, 0.5, 0.7, 1e-2, (3.0, 0.01)
    x_np, y_np, _ = synth.dumbbell_resampled(n_all)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n_all)
    te, tr = np.sort(perm[:300]), np.sort(perm[300:])
    x, y, xt = T(x_np[tr], dev), T(y_np[tr], dev), T(x_np[te], dev)
    n = x.shape[0]
    kern = mgp.kernels.RiemannMaternKernel(nu=nu, x=x, nearest_neighbors=k, laplacian_normalization="symmetric", num_modes=m,
                                           bump_scale=bump[0], bump_decay=bump[1]).to(dev)
    kern.initialize(graphbandwidth=eps, lengthscale=kappa)
    D, I = kern.knn.search(x, k)
    _knn_rows_bit_exact(x_np[tr], D, I, k, rng.choice(n, 200, replace=False))
    Dt, It = kern.knn.search(xt, k)
    from oracle import knn as oknn
    Dr, Ir = oknn.knn_search(x_np[tr], x_np[te][:50], k)
    assert np.array_equal(It[:50].cpu().numpy(), Ir) and np.array_equal(Dt[:50].cpu().numpy(), Dr)
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    model.eval()
    model.posterior(xt)
    # ---- the float64 reference pipeline (dense eigh on the device through torch: the checker, 750 MB)
    lo = _oracle_lap(kern.knn.knn_graph, eps, "symmetric")
    w, Uw = torch.linalg.eigh(T(laplacian_sym_csr(lo).toarray(), dev))
    w, Uw = w[:m + 2].cpu().numpy(), Uw[:, :m].cpu().numpy()
    gap = float(w[m] - w[m - 1])
    lam = w[:m].copy()
    lam[0] = 0.0
    Phi = Uw * (lo.degree ** -0.5)[:, None]
    Phi /= np.linalg.norm(Phi, axis=0, keepdims=True)
    Z64 = osp.features_insample(lam, Phi, nu, kappa)
    Zt64 = osp.features_oos(lo, lam, Phi, nu, kappa, Dt.double().cpu().numpy(), It.cpu().numpy(), bump[0], bump[1])
    mean_o, cov_o, alpha_o = gp_posterior_lowrank(Z64, y_np[tr], Zt64, s, noise)
    lmax = 2.0 * float(np.abs(lo.diag).max())
    res = max(kern.eigen_residuals)
    mean, cov = model.posterior_mean.double().cpu().numpy(), model.posterior_covar.double().cpu().numpy()
    Z = kern.features(x).double().cpu().numpy()
    Zt = kern.features(xt).double().cpu().numpy()
    rows = rng.choice(n, 256, replace=False)
    e = dict(evals=float(np.abs(kern.eigval.cpu().numpy()[1:] - lam[1:]).max() / lmax),
             kernel=float(np.abs(Z[rows] @ Z.T - Z64[rows] @ Z64.T).max() / np.abs(Z64[rows] @ Z64.T).max()),
             cross=float(np.abs(Zt @ Z.T - Zt64 @ Z64.T).max() / np.abs(Zt64 @ Z64.T).max()),
             mean=float(np.abs(mean - mean_o).max() / np.abs(mean_o).max()),
             var=float(np.abs(np.diag(cov) - np.diag(cov_o)).max() / np.abs(np.diag(cov_o)).max()),
             cov=float(np.abs(cov - cov_o).max() / np.abs(cov_o).max()))
    print("C2 end to end vs dense float64 pipeline: gap behind the kept block %.3e, eigensolver residual %.2e (|R| / gap = %.1e); %s"
          % (gap, res, res * np.sqrt(m) / gap, " ".join("%s %.2e" % kv for kv in e.items())))
    assert (np.abs(Zt64).sum(1) > 0).mean() > 0.9                           # the held-out points lie in the bump support
    assert res * np.sqrt(m) / gap < 0.1                                      # a conditioned cut (Davis-Kahan)
    assert e["evals"] < 1e-6 and e["kernel"] < 1e-4 and e["cross"] < 1e-4, e
    assert e["mean"] < 1e-4 and e["var"] < 1e-4 and e["cov"] < 1e-4, e


# ============================================================================= C3 / C4 (shared 60k graph)
@pytest.fixture(scope="module")
def rmnist60k(mgp, dev):
    """S3/S4 of SURVEY.md section 8(d): RMNIST-like 606 bases x 100 rotations; 600 random rows are held out as test
    points, the other 60 000 are the graph nodes.  k = 50, random walk, nu = 2, 100 modes, trained hyper-parameters
    (models/srmnist_manifold_semisupervised.pth) with the bandwidth floored by the notebooks' eps_min rule."""
    from tools import synth
    x_all, y_all = synth.rmnist_like(606, 100, seed=1337, device=dev)
    rng = np.random.default_rng(1337)
    perm = rng.permutation(x_all.shape[0])
    test_rows, train_rows = np.sort(perm[:600]), np.sort(perm[600:])
    x, y = x_all[T(train_rows, dev)].contiguous(), y_all[T(train_rows, dev)].contiguous()
    xt, yt = x_all[T(test_rows, dev)].contiguous(), y_all[T(test_rows, dev)].contiguous()
    hp = _hp("srmnist_manifold_semisupervised")
    kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk",
                                           num_modes=100, bump_scale=3.0, bump_decay=0.01).to(dev)
    D, I = kern.knn.search(x, 50)
    eps, eps_min = synth.bandwidth_rule(D[:, 1].cpu().numpy(), hp["graphbandwidth"])
    kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
    return dict(x=x, y=y, xt=xt, yt=yt, kern=kern, hp=hp, eps=eps, D=D, I=I)


def test_c3_rmnist_60k_knn_spectrum_posterior(mgp, dev, rmnist60k):
    """C3, the north-star target at its size: 60 000 x 784 points, k = 50.
      * 256 sampled rows of the 60k x 60k k-NN search (matrix-core keys + fp64 re-rank) bit-exact vs the C oracle;
      * L v against the float64 oracle;
      * eval(): residuals of sampled eigenpairs re-evaluated by the float64 oracle;
      * GP posterior mean / covariance at 600 held-out points (out-of-sample features, Woodbury on device, covariance
        block on the MFMA) within 1e-4 of the float64 closed form on the same features (oracle/solvers.py; what
        gpytorch evaluates for the reference: riemann_gp.py:45-75);
      * the posterior mean at the graph nodes in precision form -- the CG solve of (K + s I) x = y, K = Q^-1 --
        within 1e-4 of a float64 CG on the oracle's operator."""
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from manifold_gp_amd.solvers import cg_solve, kernel_block, lowrank_solve
    from oracle.solvers import gp_posterior_lowrank
    from oracle.sparse import SparsePrecision
    w = rmnist60k
    x, y, xt, kern, hp, eps = w["x"], w["y"], w["xt"], w["kern"], w["hp"], w["eps"]
    n = x.shape[0]
    assert n == 60000 and x.shape[1] == 784
    x_np = x.cpu().numpy()
    rows = np.random.default_rng(5).choice(n, 256, replace=False)
    _knn_rows_bit_exact(x_np, w["D"], w["I"], 50, rows)                       # nearest_neighbors.py:35-37
    graph = kern.knn.knn_graph
    lo = _oracle_lap(graph, eps, "randomwalk")
    sq = SparsePrecision(lo, 2, hp["lengthscale"], hp["outputscale"])
    lmax = 2.0 * float(np.abs(lo.diag).max())
    # ---- SpMV, both orientations
    v = torch.randn(n, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():                                  # (the kernel's bandwidth is a Parameter)
        lap = kern.laplacian()
        for op, tr in ((lap, False), (lap.T, True)):
            ref = sq.laplacian_matmul(v.double().numpy(), transposed=tr)
            out = op.matmul(v.to(dev)).cpu().numpy()
            assert np.abs(out - ref).max() < 4e-6 * lmax * float(v.abs().max()), tr
    # ---- eval(): eigenpairs of L_sym (riemann_kernel.py:117-130); phi = normalise(D^-1/2 u)
    kern.keep_eigen_block = True
    kern.eval()
    assert float(kern.eigval[0]) == 0.0 and kern.eigvec.shape == (n, 100)
    assert max(kern.eigen_residuals) <= 2e-5 * lmax
    # the kernel's default tolerance (1e-6 lambda_max) is below what fp32 can reach on this clustered spectrum: the solver must
    # notice (predicted from the gap behind its block) and hand back the block after a handful of rounds, not after 60
    from manifold_gp_amd.solvers import lanczos_smallest as _ls
    rounds, applies, under_tol, bsize = _ls.last_info
    assert rounds <= 6 and bsize == 128 and under_tol < 100, _ls.last_info
    cols = [0, 1, 2, 49, 98, 99]
    Phi = kern.eigvec[:, cols].double().cpu().numpy()
    U = Phi * np.sqrt(lo.degree)[:, None]
    U /= np.linalg.norm(U, axis=0, keepdims=True)
    lam = kern.eigval[cols].double().cpu().numpy()
    lam[0] = float(U[:, 0] @ (sq.L @ U[:, 0]))                              # eigval[0] is SET to 0 (:126)
    R = sq.L @ U - U * lam[None, :]
    assert np.linalg.norm(R, axis=0).max() <= 3e-5 * lmax, np.linalg.norm(R, axis=0) / lmax
    # ---- an INDEPENDENT float64 look at the spectral stage: Rayleigh-Ritz of the solver's whole block (the 100 kept
    # columns + 28 guard columns) with the oracle's float64 CSR
    m = 100
    blk = kern.eigen_block
    V = blk["evecs"].double().cpu().numpy()
    assert V.shape == (n, 128)
    assert np.abs(V.T @ V - np.eye(128)).max() < 5e-5                        # the block is orthonormal
    Qb, _ = np.linalg.qr(V)
    LQ = sq.L @ Qb
    Hb = Qb.T @ LQ
    th, Sb = np.linalg.eigh(0.5 * (Hb + Hb.T))
    Rb = LQ @ Sb - (Qb @ Sb) * th[None, :]
    rn = np.linalg.norm(Rb, axis=0)
    gap = float(th[m] - th[m - 1])
    dk = float(np.linalg.norm(Rb[:, :m])) / max(gap, 1e-300)
    e_ritz = float(np.abs(blk["evals"].cpu().numpy() - th).max())
    print("C3 spectral stage in float64: Ritz values theta[0, 1, 99, 100, 127] = %s; HIP Ritz values differ by %.1e; "
          "residuals of the kept pairs <= %.2e (%.1e lambda_max); gap theta_100 - theta_99 = %.2e; Davis-Kahan "
          "sin(angle) <= |R| / gap = %.1e" % (np.array2string(th[[0, 1, 99, 100, 127]], precision=3), e_ritz, rn[:m].max(),
                                               rn[:m].max() / lmax, gap, dk))
    assert e_ritz <= 1e-6 * lmax                                              # the Ritz values are float64's (measured 1.6e-7)
    assert rn[:m].max() <= 2e-5 * lmax                                        # backward error of every kept pair (measured 8e-6)
    assert th[0] > -1e-6 * lmax and np.all(np.diff(th) >= 0)
    # What this says about "within 1e-4 of the reference" at this size.  By Cauchy interlacing lambda_i <= theta_i: the
    # graph has at least 128 eigenvalues below theta_127 (measured 5e-5, i.e. 1.7e-6 lambda_max) -- its ~600 rotation orbits
    # are nearly disconnected clusters at the notebooks' bandwidth rule -- and neighbouring eigenvalues at the m = 100
    # cut are ~2e-6 apart, below the float32 rounding of the matrix entries themselves (ulp(lambda_max) = 3.5e-6 per
    # entry) and far below any float32 eigensolver's resolution: the reference's own float32 dense eigh (riemann_kernel.py:
    # 124; 14.4 GB at this size, it cannot run) would return an equally arbitrary basis of that cluster, so the 100-mode
    # kernel is NOT a well-defined function of the data here and no independent float64 value exists to be within 1e-4
    # of.  The test therefore asserts (a) above: every returned pair is an exact eigenpair of a matrix within 2e-5
    # lambda_max of L; (b) below: everything DOWNSTREAM of the eigenvectors -- post-processing, in- and out-of-sample
    # features, kernel entries, Woodbury posterior -- within 1e-4 of a float64 evaluation by the oracle that is handed
    # ONLY the solver's raw block (not the HIP features); and (c) the end-to-end 1e-4 check against an independent
    # float64 eigendecomposition where one is well defined: tests/golden/dumbbell_posterior.npz (reference pipeline)
    # and test_c2_dumbbell_10k_posterior_vs_dense_float64_pipeline (N = 10k).
    ill_conditioned = rn[:m].max() > 0.1 * gap
    assert ill_conditioned, "the cut became well conditioned: tighten this test to the Davis-Kahan bound"
    # (b) float64 pipeline from the solver's raw Ritz block: riemann_kernel.py:126-128 (lambda_0 = 0, D^-1/2, normalise),
    # :134-136 (in-sample features), :138-147 + graph_laplacian_operator.py:146-157 (out-of-sample features)
    from oracle import spectral as osp
    lam64 = blk["evals"].double().cpu().numpy()[:m].copy()
    lam64[0] = 0.0
    Phi64 = V[:, :m] * (lo.degree ** -0.5)[:, None]
    Phi64 /= np.linalg.norm(Phi64, axis=0, keepdims=True)
    Z64 = osp.features_insample(lam64, Phi64, 2, hp["lengthscale"])
    Dt, It = kern.knn.search(xt, 50)
    Zt64 = osp.features_oos(lo, lam64, Phi64, 2, hp["lengthscale"], Dt.double().cpu().numpy(), It.cpu().numpy(), 3.0, 0.01)
    # ---- spectral posterior at 600 held-out points
    s, noise = hp["outputscale"], hp["noise"]
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    model.eval()
    model.posterior(xt)
    Z, Zt = kern.features(x), kern.features(xt)
    assert float((Zt.abs().sum(1) > 0).float().mean()) > 0.9                 # the held-out points lie in the bump support
    mean_o, cov_o, alpha_o = gp_posterior_lowrank(Z64, y.cpu().numpy(), Zt64, s, noise)
    mean, cov = model.posterior_mean.cpu().numpy(), model.posterior_covar.cpu().numpy()
    e_mean = np.abs(mean - mean_o).max() / np.abs(mean_o).max()
    e_cov = np.abs(cov - cov_o).max() / np.abs(cov_o).max()
    e_var = np.abs(np.diag(cov) - np.diag(cov_o)).max() / np.abs(np.diag(cov_o)).max()
    e_z = np.abs(Z.double().cpu().numpy() - Z64).max() / np.abs(Z64).max()
    e_zt = np.abs(Zt.double().cpu().numpy() - Zt64).max() / np.abs(Zt64).max()
    print("C3 downstream of the eigenvectors vs float64: features %.2e / %.2e, posterior mean %.2e cov %.2e var %.2e"
          % (e_z, e_zt, e_mean, e_cov, e_var))
    assert e_z < 1e-5 and e_zt < 1e-4
    assert e_mean < 1e-4 and e_cov < 1e-4 and e_var < 1e-4, (e_mean, e_cov, e_var)
    alpha = lowrank_solve(Z, y, s, noise).cpu().numpy()                      # (K + noise I)^-1 y at the nodes
    assert np.abs(alpha - alpha_o).max() < 1e-4 * np.abs(alpha_o).max()
    # kernel entries: 256 x 60000 and 600 x 60000 blocks of s Z1 Z2^T on the fp32 MFMA against the float64 features' products
    # (riemann_kernel.py:92-100)
    for Z1, Z1o in ((Z[T(rows, dev)].contiguous(), Z64[rows]), (Zt, Zt64)):
        Kb = kernel_block(Z1, Z, s).double().cpu().numpy()
        Kr = s * (Z1o @ Z64.T)
        assert np.abs(Kb - Kr).max() < 1e-4 * np.abs(Kr).max(), np.abs(Kb - Kr).max() / np.abs(Kr).max()
    # ---- precision form: (I + noise s Q) x = y by the HIP CG vs float64 CG on the oracle's operator
    with torch.no_grad():
        desc = kern.precision()._descriptor().with_(scale=s, form=2, noise=noise)
        sol, its, res = cg_solve(desc, y, tol=1e-6, stop_mode=1)
    ref = sq.solve(y.double().cpu().numpy(), matvec=lambda z: sq.posterior_system(z, noise))
    e_cg = np.abs(sol.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("C3 precision-form CG: %d iterations, rel err %.2e" % (its, e_cg))
    assert e_cg < 1e-4 and max(res) <= 1e-6


def test_c3_manifold784_60k_posterior_vs_independent_float64(mgp, dev):
    """The north-star target -- "GP posterior on the 60k graph within 1e-4 of reference" -- END TO END at C3's size on a
    workload where it is decidable: tools/synth.py::manifold_784, 60 000 graph nodes + 600 held-out points of a swiss roll
    embedded in R^784 (k = 50, random walk, nu = 2, 100 modes: C3's shapes, so the matrix-core k-NN, the tile SpMV, the
    block eigensolver, the fused feature kernels, the MFMA covariance block and the Woodbury solve all run at that size),
    whose spectrum has a measured gap of ~2e-4 lambda_max behind mode 100 (the RMNIST-like set above has none).
    Checker, independent of the HIP eigensolver: the reference's pipeline in float64 -- oracle Laplacian
    (graph_laplacian_operator.py:52-106) -> the 104 smallest eigenpairs of L_sym by scipy's shift-invert eigsh (ARPACK
    on a SuperLU factorisation of L + 1e-3 lambda_max I; stands for the dense eigh of riemann_kernel.py:121-125, which
    needs 28.8 GB in double at this size) -> lambda_0 = 0, D^-1/2, column normalisation (:126-128) -> in- / out-of-sample
    features (:134-147) -> Woodbury posterior (riemann_gp.py:45-75).  Nothing of the HIP path is fed to it except the
    k-NN lists, which are bit-exact against the C oracle on sampled rows."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from manifold_gp_amd.solvers import cg_solve, kernel_block
    from oracle import knn as oknn
    from oracle import spectral as osp
    from oracle.solvers import gp_posterior_lowrank
    from oracle.sparse import SparsePrecision
    from tools import synth
    n_all, k, m, nu = 60600, 50, 100, 2
    eps, kappa, s, noise, bump = 0.3, 3.0, 1.0, 1e-2, (3.0, 0.01)
    x_np, y_np, _ = synth.manifold_784(n_all)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n_all)
    te, tr = np.sort(perm[:600]), np.sort(perm[600:])
    x, y, xt = T(x_np[tr], dev), T(y_np[tr], dev), T(x_np[te], dev)
    n = x.shape[0]
    assert n == 60000 and x.shape[1] == 784
    kern = mgp.kernels.RiemannMaternKernel(nu=nu, x=x, nearest_neighbors=k, laplacian_normalization="randomwalk", num_modes=m,
                                           bump_scale=bump[0], bump_decay=bump[1]).to(dev)
    kern.initialize(graphbandwidth=eps, lengthscale=kappa)
    assert kern.eigen_tol == 1e-6                                              # the shipped default, not a tuned one
    D, I = kern.knn.search(x, k)
    _knn_rows_bit_exact(x_np[tr], D, I, k, rng.choice(n, 256, replace=False))
    Dt, It = kern.knn.search(xt, k)
    Dr, Ir = oknn.knn_search(x_np[tr], x_np[te][:64], k)
    assert np.array_equal(It[:64].cpu().numpy(), Ir) and np.array_equal(Dt[:64].cpu().numpy(), Dr)
    # ---- the HIP pipeline: graph -> Laplacian -> eigensolve -> features -> posterior at 600 held-out points
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    model.eval()
    model.posterior(xt)
    # ---- the independent float64 pipeline
    lo = _oracle_lap(kern.knn.knn_graph, eps, "randomwalk")
    sq = SparsePrecision(lo, nu, kappa, s)
    L = sq.L
    lmax = 2.0 * float(np.abs(lo.diag).max())
    shift = 1e-3 * lmax
    lu = spla.splu((L + shift * sp.identity(n)).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                   options=dict(SymmetricMode=True))
    w, U = spla.eigsh(L, k=m + 4, sigma=-shift, which="LM", tol=1e-12,
                      OPinv=spla.LinearOperator((n, n), matvec=lu.solve, dtype=np.float64))
    o = np.argsort(w)
    w, U = w[o], U[:, o]
    assert np.linalg.norm(L @ U - U * w[None, :], axis=0).max() < 1e-10 * lmax  # the checker's pairs are float64-exact
    assert abs(w[0]) < 1e-10 * lmax and w[1] > 1e-5 * lmax                      # one connected component
    gap = float(w[m] - w[m - 1])
    lam = w[:m].copy()
    lam[0] = 0.0
    Phi = U[:, :m] * (lo.degree ** -0.5)[:, None]
    Phi /= np.linalg.norm(Phi, axis=0, keepdims=True)
    Z64 = osp.features_insample(lam, Phi, nu, kappa)
    Zt64 = osp.features_oos(lo, lam, Phi, nu, kappa, Dt.double().cpu().numpy(), It.cpu().numpy(), bump[0], bump[1])
    mean_o, cov_o, alpha_o = gp_posterior_lowrank(Z64, y_np[tr], Zt64, s, noise)
    rmse = float(np.sqrt(((mean_o - y_np[te]) ** 2).mean()))
    assert rmse < 0.2, rmse                                  # the model predicts (targets are standardised; noise 0.1 / std)
    # ---- compare
    res = max(kern.eigen_residuals)
    mean, cov = model.posterior_mean.double().cpu().numpy(), model.posterior_covar.double().cpu().numpy()
    Zd, Ztd = kern.features(x), kern.features(xt)
    Z, Zt = Zd.double().cpu().numpy(), Ztd.double().cpu().numpy()
    rows = rng.choice(n, 256, replace=False)
    Kr, Kc = Z64[rows] @ Z64.T, Zt64 @ Z64.T
    e = dict(evals=float(np.abs(kern.eigval.cpu().numpy()[1:] - lam[1:]).max() / lmax),
             evals_rel=float((np.abs(kern.eigval.cpu().numpy()[1:] - lam[1:]) / lam[1:]).max()),
             kernel=float(np.abs(Z[rows] @ Z.T - Kr).max() / np.abs(Kr).max()),
             cross=float(np.abs(Zt @ Z.T - Kc).max() / np.abs(Kc).max()),
             mean=float(np.abs(mean - mean_o).max() / np.abs(mean_o).max()),
             var=float(np.abs(np.diag(cov) - np.diag(cov_o)).max() / np.abs(np.diag(cov_o)).max()),
             cov=float(np.abs(cov - cov_o).max() / np.abs(cov_o).max()))
    # the same kernel entries through the fp32 MFMA block (riemann_kernel.py:92-100)
    Kb = kernel_block(Zd[T(rows, dev)].contiguous(), Zd, s).double().cpu().numpy()
    e["kernel_mfma"] = float(np.abs(Kb - s * Kr).max() / np.abs(s * Kr).max())
    Kb = kernel_block(Ztd, Zd, s).double().cpu().numpy()
    e["cross_mfma"] = float(np.abs(Kb - s * Kc).max() / np.abs(s * Kc).max())
    print("C3-size manifold end to end vs independent float64 pipeline: gap behind mode %d = %.3e (%.1e lambda_max), eigensolver "
          "residual %.2e (|R| sqrt(m) / gap = %.1e), oracle test rmse %.3f; %s"
          % (m, gap, gap / lmax, res, res * np.sqrt(m) / gap, rmse, " ".join("%s %.2e" % kv for kv in e.items())))
    assert (np.abs(Zt64).sum(1) > 0).mean() > 0.9                            # the held-out points lie in the bump support
    assert gap > 1e-4 * lmax and res * np.sqrt(m) / gap < 0.1                 # a conditioned cut (Davis-Kahan)
    assert e["evals"] < 1e-6 and e["evals_rel"] < 1e-4, e
    assert max(e["kernel"], e["cross"], e["kernel_mfma"], e["cross_mfma"]) < 1e-4, e
    assert e["mean"] < 1e-4 and e["var"] < 1e-4 and e["cov"] < 1e-4, e
    # ---- the same posterior mean at the graph nodes in precision form: (I + noise s Q) x = y by the HIP CG against a float64
    # CG on the oracle's operator (precision_matern_operator.py:26-37)
    with torch.no_grad():
        desc = kern.precision()._descriptor().with_(scale=s, form=2, noise=noise)
        sol, its, resid = cg_solve(desc, y, tol=1e-6, stop_mode=1)
    ref = sq.solve(y_np[tr].astype(np.float64), matvec=lambda z: sq.posterior_system(z, noise))
    e_cg = np.abs(sol.cpu().numpy() - ref).max() / np.abs(ref).max()
    print("C3-size manifold precision-form CG: %d iterations, rel err %.2e" % (its, e_cg))
    assert e_cg < 1e-4 and max(resid) <= 1e-6


def test_c3_manifold784_warm_started_eval(mgp, dev):
    """eval() again after a small bandwidth change (riemann_kernel.py:117-130 re-runs the whole decomposition every time; a
    training loop that toggles train / eval moves the bandwidth a little per step): the block iteration starts from the previous
    Rayleigh-Ritz block (mgp_lanczos_smallest_warm) and must land on what a COLD solve of the same matrix gives -- eigenvalues,
    and everything downstream through the features' Gram -- in fewer rounds; a graph change, a changed mode count and a
    kernel with warm_start off take the cold path; on the RMNIST-like graph, whose solve ends at the fp32 floor, no block is kept."""
    from tools import synth
    n, k, m, nu, eps, kappa = 60000, 50, 100, 2, 0.3, 3.0
    x_np, y_np, _ = synth.manifold_784(n)
    x = T(x_np, dev)
    kern = mgp.kernels.RiemannMaternKernel(nu=nu, x=x, nearest_neighbors=k, laplacian_normalization="randomwalk", num_modes=m).to(dev)
    assert kern.warm_start is True
    rows = T(np.random.default_rng(3).choice(n, 256, replace=False), dev)

    def run(e, warm):
        kern.warm_start = warm
        kern.initialize(graphbandwidth=e, lengthscale=kappa)
        kern.eval()
        Z = kern.features(x)
        return kern.eigval.double().cpu().numpy(), (Z[rows] @ Z.t()).double().cpu().numpy(), list(kern.eigen_info), max(kern.eigen_residuals)

    run(eps, True)                                            # cold (nothing kept yet), leaves its block behind
    assert kern._eigen_warm is not None and kern.eigen_info[2] == m
    for f in (1.01, 1.05):
        lam_w, K_w, info_w, res_w = run(eps * f, True)        # warm from the eps block
        assert kern._eigen_warm is not None
        run(eps, True)                                        # put the eps block back
        lam_c, K_c, info_c, res_c = run(eps * f, False)       # cold, same matrix
        lmax = 2.0 * float(kern.laplacian_operator.data.diag.max())
        assert info_w[2] == m and info_c[2] == m
        assert info_w[1] < 0.7 * info_c[1], (info_w, info_c)                      # fewer block products
        assert np.abs(lam_w - lam_c).max() < 1e-6 * lmax and (np.abs(lam_w[1:] - lam_c[1:]) / lam_c[1:]).max() < 1e-4
        assert np.abs(K_w - K_c).max() < 1e-4 * np.abs(K_c).max()
        assert res_w < 1e-5 * lmax
        print("warm eval at eps x %.2f: %d block products against %d cold (rounds %d / %d)" % (f, info_w[1], info_c[1], info_w[0], info_c[0]))
        run(eps, True)
    # another mode count: the kept block does not fit -> cold
    kern.num_modes = 64
    kern.warm_start = True
    kern.eval()
    assert kern.eigen_info[2] == 64 and kern.eigval.shape[0] == 64
    kern.num_modes = m


def test_c4_semisupervised_60k_schur(mgp, dev, rmnist60k):
    """C4: the same 60k graph, 10 % labelled (randperm seed 1337, examples/RMNIST_semisupervised_learning.ipynb:65,99-101).
    Schur complement matvec (schur_complement_operator.py:26-30, nested HIP CG on the 54k unlabelled block) against
    the float64 oracle with a converged inner solve; symmetry; S.solve by block elimination: true residual; two
    epochs of manifold_informed_train (precision-form loss with the Schur complement inside): finite, decreasing."""
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from manifold_gp_amd.utils import manifold_informed_train
    from oracle.sparse import SparsePrecision
    w = rmnist60k
    x, y, kern, hp, eps = w["x"], w["y"], w["kern"], w["hp"], w["eps"]
    n = x.shape[0]
    torch.manual_seed(1337)
    labeled = torch.zeros(n, dtype=torch.bool, device=dev)
    labeled[torch.randperm(n, device=dev)[: n // 10]] = True
    assert int(labeled.sum()) == 6000
    mask = labeled.cpu().numpy()
    Q = kern.precision()
    S = mgp.operators.SchurComplementOperator(Q, labeled)
    assert tuple(S.shape) == (6000, 6000)
    lo = _oracle_lap(kern.knn.knn_graph, eps, "randomwalk")
    sq = SparsePrecision(lo, 2, hp["lengthscale"])
    v = y[labeled].contiguous()
    gen = torch.Generator().manual_seed(3)
    u = torch.randn(6000, generator=gen).to(dev)
    with torch.no_grad(), mgp.settings.cg_tolerance(1e-7), mgp.settings.cg_stop_mode(1), mgp.settings.max_cg_iterations(20000):
        Sv, Su = S.matmul(v), S.matmul(u)
        ref = sq.schur_matmul(v.double().cpu().numpy(), mask)
        e_mv = np.abs(Sv.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert e_mv < 1e-4, e_mv
        sym = abs(float(torch.dot(u, Sv) - torch.dot(v, Su))) / float(Sv.norm() * u.norm())
        assert sym < 1e-5, sym
        xs = S.solve(v)
        r = S.matmul(xs) - v
        assert float(r.norm() / v.norm()) < 1e-4
    full = np.zeros(n)
    full[mask] = v.double().cpu().numpy()
    ref_x = sq.solve(full)[mask]                                              # block elimination: (Q^-1 [b; 0])_l = S^-1 b
    e_solve = np.abs(xs.cpu().numpy() - ref_x).max() / np.abs(ref_x).max()
    print("C4: Schur matvec err %.2e, symmetry %.2e, solve err %.2e" % (e_mv, sym, e_solve))
    assert e_solve < 1e-4
    # ---- two epochs of the semi-supervised training loop (train_model.py:49-109)
    kern2 = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk",
                                            num_modes=100).to(dev)
    kern2.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
    model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev),
                      ScaleKernel(kern2, hp["outputscale"]).to(dev), labeled=labeled).to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    losses = []

    class Rec:
        def step(self, loss):
            losses.append(float(loss.detach()))
    torch.manual_seed(0)
    last = manifold_informed_train(model, opt, max_iter=2, tolerance=0.0, num_rand_vec=32, max_cholesky=800,
                                   cg_tolerance=1e-2, cg_max_iter=1000, scheduler=Rec())
    print("C4 training losses", losses)
    assert len(losses) == 3 and all(np.isfinite(losses)) and np.isfinite(last)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("recurrence", ["pipelined", "chronopoulos-gear"])
def test_c4_partition_60k_eight_virtual_ranks(mgp, dev, rmnist60k, recurrence):
    """Config C4's partition at its size (rows AND vectors of the 60k graph over the 8 GPUs of a node)
    on ONE GPU with 8 VIRTUAL ranks: every rank builds its own row block, ghost layers, tile view and plan exactly as its
    process of the RCCL job would; the all-gather is the identity on a shared buffer.  Against the single-GPU solve:
    same solution to round-off, same iteration count, every rank's ghost-row count inside the envelope the partition of
    this graph gives (measured 13-21 thousand rows for nu = 2: one neighbour layer of a 7 500-row block), padding rows
    stay zero, true residual of the assembled solution."""
    from manifold_gp_amd.graph import LaplacianData
    from manifold_gp_amd.parallel import RowPartition, pad_graph, virtual_pcg_solve
    from manifold_gp_amd.solvers import cg_solve
    w = rmnist60k
    kern, hp, eps, y = w["kern"], w["hp"], w["eps"], w["y"]
    g = kern.knn.knn_graph
    world = 8
    with torch.no_grad():
        base = kern.precision()._descriptor().with_(scale=hp["outputscale"], form=2, noise=hp["noise"])
        xs, its1, _ = cg_solve(base, y, tol=1e-6, stop_mode=1)
        part = RowPartition(g.n, world)
        assert part.n_loc == 7552 and part.n_pad == 8 * 7552       # whole 64-row tiles per rank: 416 padding rows on the last one
        gp = pad_graph(g, part.n_pad)
        data = LaplacianData(gp, eps, True)
        desc = base.with_(data=data, pre=data.dsqrt, post=data.dsqrt)
        x, its, status, ghosts = virtual_pcg_solve(desc, part, part.pad(y), tol=1e-6, max_iter=4000, stop_mode=1,
                                                   recurrence=recurrence)
        r = base.apply(x[:g.n]) - y
    print("C4 partition, 8 virtual ranks, %s: %d iterations (one GPU: %d), ghost rows per rank %s" % (recurrence, its, its1, ghosts))
    assert status == 1 and abs(its - its1) <= 1
    assert len(ghosts) == world and all(5000 <= gh <= 30000 for gh in ghosts), ghosts
    assert float(x[g.n:].abs().max()) == 0.0                                 # padding rows: b = 0 -> x = 0
    assert float((x[:g.n] - xs.view(-1)).abs().max()) < 2e-4 * float(xs.abs().max())
    assert float(r.norm() / y.norm()) < 2e-5


# ============================================================================= C5
def test_c5_swiss_roll_1m_pipeline(mgp, dev):
    """C5: 1 000 000 points on a swiss roll in R^3 handed over in random order, k = 64, symmetric, nu = 2.
    400 sampled k-NN rows bit-exact; the graph picks a locality order for its tiles; L v against the float64
    oracle at full size; a 64-column product through the matrix-core tile kernel against the same oracle; adjoint and
    linearity properties of the precision; CG with fp64-residual refinement:
    TRUE relative residual (re-evaluated by the float64 oracle) <= a few 1e-6."""
    from manifold_gp_amd.solvers import CgPlan
    from oracle.sparse import SparsePrecision
    from tools import synth
    n, k = 1000000, 64
    x_np, y_np = synth.swiss_roll(n, order="random")
    x, y = T(x_np, dev), T(y_np, dev)
    knn = mgp.utils.NearestNeighbors(x)
    D, I = knn.search(x, k)
    assert knn.last_stats["candidates"] == -1                                  # the slab-free low-d path
    _knn_rows_bit_exact(x_np, D, I, k, np.random.default_rng(9).choice(n, 400, replace=False))
    assert bool((I[:, 0] == torch.arange(n, device=dev)).all())
    idx, val = knn.graph(k)
    g = knn.knn_graph
    assert 0.5 * n * (k - 1) <= g.M <= n * (k - 1)
    assert g.tiles is not None and g.tiles.get("rowid") is not None and g.tiles["reuse"] > 5   # locality order applied
    eps_min = synth.bandwidth_rule(D[:200000, 1].cpu().numpy(), 0.0)[1]
    eps = 3.0 * eps_min
    del D, I
    lap = mgp.operators.GraphLaplacianOperator(val, idx, n, torch.tensor([[eps]], device=dev), "symmetric", graph=g)
    lo = _oracle_lap(g, eps, "symmetric")
    kappa, s, noise = 1.0, 1.0, 0.01
    sq = SparsePrecision(lo, 2, kappa, s)
    lmax = 2.0 * float(np.abs(lo.diag).max())
    v = torch.randn(n, generator=torch.Generator().manual_seed(2))
    ref = sq.L @ v.double().numpy()
    out = lap.matmul(v.to(dev)).cpu().numpy()
    assert np.abs(out - ref).max() < 4e-6 * lmax * float(v.abs().max())
    Q = mgp.operators.PrecisionMaternOperator(lap, 2, torch.tensor([[kappa]], device=dev))
    a, b = torch.randn(n, device=dev), torch.randn(n, device=dev)
    Qa, Qb = Q.matmul(a), Q.matmul(b)
    assert abs(float(torch.dot(b.double(), Qa.double()) - torch.dot(a.double(), Qb.double()))) < 1e-5 * float(Qa.norm() * b.norm())
    lin = Q.matmul(2.0 * a - 3.0 * b) - (2.0 * Qa - 3.0 * Qb)
    assert float(lin.norm()) < 1e-5 * float(Qa.norm() + Qb.norm())
    # ---- a 64-column product at this size through the matrix-core tile kernel (what the eigensolver's block iteration and the
    # many-column solves run on: P L P^T in the library's locality order, graph.MtPlan) against the float64 oracle
    import ctypes
    from manifold_gp_amd import _lib
    rel = lap.data.relabelled()
    mt = rel.mt_plan()
    assert mt is not None and mt.tiles == n // 16 and mt.fill > 0.25
    csr = rel.csr(wide=True)
    assert _lib.lib().mgp_spmm_kernel_choice(ctypes.byref(csr), 64, 0, 0) == 3
    Xw = torch.randn(n, 64, generator=torch.Generator().manual_seed(3))
    Xp = rel.graph.permute(Xw.to(dev)).contiguous()
    Yp = torch.full_like(Xp, float("nan"))
    _lib.check(_lib.lib().mgp_spmm_fused(ctypes.byref(csr), _lib.ptr(Xp), 64, _lib.ptr(Yp), 0.0, 1.0, None, None, None, 0.0, 1.0,
                                         None, None, _lib.stream()), "mgp_spmm_fused")
    Yw = rel.graph.unpermute(Yp).cpu().numpy()
    refw = sq.L @ Xw.double().numpy()
    assert not np.isnan(Yw).any() and np.abs(Yw - refw).max() < 4e-6 * lmax * float(Xw.abs().max())
    del Xp, Yp, Yw, refw
    # ---- posterior mean in precision form: (I + noise s Q) x = y, fp32 CG + fp64-residual refinement
    desc = Q._descriptor().with_(scale=s, form=2, noise=noise)
    plan = CgPlan(desc, 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=3)
    sol = plan.solve(y.view(-1, 1).contiguous()).clone()
    assert plan.status == 1 and max(plan.resid) <= 2e-6, (plan.status, plan.resid)
    # the refined solve accumulates x in float64 (mgp_cg_plan_x64); what solve() returns is its float32 rounding.
    # At this conditioning (noise |Q| ~ 1e3..1e4) rounding x to float32 alone moves the residual by ~1e-4, so the TRUE
    # residual is re-evaluated by the float64 oracle on the float64 solution
    x64 = plan.solution64_view()[:, 0]
    assert torch.equal(x64.float(), sol[:, 0])
    # The system that is solved is the one with the fp32 Laplacian the path stores (as the reference does); its
    # entries were checked against the float64 oracle above.  TRUE residual of THAT system, re-evaluated on the host
    # in float64 from the device's CSR arrays (independent of the device's fp64 apply kernel):
    import scipy.sparse as sp
    d = lap.data
    S_dev = sp.csr_matrix((d.vals.double().cpu().numpy(), g.col.cpu().numpy(), g.rowptr.cpu().numpy()), shape=(n, n))
    L_dev = sp.diags(d.diag.double().cpu().numpy()) - S_dev
    tau = 2.0 * 2 / kappa ** 2

    def A_dev(z):
        t = z
        for _ in range(2):
            t = tau * t + L_dev @ t
        return z + noise * s * t
    yd = y_np.astype(np.float64)
    xd = x64.cpu().numpy()
    true_rel = float(np.linalg.norm(yd - A_dev(xd)) / np.linalg.norm(yd))
    rel32 = float(np.linalg.norm(yd - A_dev(sol[:, 0].double().cpu().numpy())) / np.linalg.norm(yd))
    rel_oracle = float(np.linalg.norm(yd - sq.posterior_system(xd, noise)) / np.linalg.norm(yd))
    print("C5: CG %d iterations; true relative residual %.2e (plan reports %.2e); of the float32 rounding of x: %.2e; "
          "against the float64-evaluated matrix: %.2e" % (plan.iters, true_rel, max(plan.resid), rel32, rel_oracle))
    assert true_rel <= 3e-6, true_rel
    assert rel_oracle <= 1e-3                     # fp32 storage of the matrix: ~1e-7 |A| |x| / |y|
    plan.close()


@pytest.mark.gpu
def test_bench_line_contract(dev):
    """bench.py prints ONE JSON line with the driver's keys, the roofline and cpu_baseline objects and a solution that
    matches the CPU baseline's (a reduced node count keeps the CPU leg to a second; the full-size line is the driver's)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--nodes", "8000"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["higher_is_better"] is True
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["vs_baseline"] is None
    assert line["scaling"] in ("strong", "weak") and "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # what actually ran is in the line: the untimed pre-heat, the step with an alternating right-hand side (graph node
    # re-pointed every solve), and the two live timings of the dominant kernel that bracket its in-solve duration
    assert "preheat_solves" in line and line["ms_per_step_alternating_rhs"] > 0
    # `frac` quotes the committed in-graph profile only when that profile was taken on this source tree and agrees with this
    # run's back-to-back figure; otherwise the live in-solve figure (a reduced node count never matches a profile)
    assert roof["profile_age_ok"] is False and roof["measured"] == "live_eager_in_solve"
    assert len(roof["source_hash"]) == 16 and abs(roof["avg_launch_us"] - roof["live_eager_in_solve"]["avg_launch_us"]) < 1e-6
    b2b, eag = roof["live_back_to_back"], roof["live_eager_in_solve"]
    assert b2b["avg_launch_us"] > 0 and eag["avg_launch_us"] > 0 and "measured" in roof
    assert 0.5 * b2b["avg_launch_us"] < eag["avg_launch_us"] < 4 * b2b["avg_launch_us"]
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert line["config"]["cg_true_residual_fp32_apply"] < 1e-5 and line["config"]["max_rel_diff_vs_cpu_solution"] < 1e-4


def _run_bench(args, env_extra=None, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(args), capture_output=True, text=True,
                       timeout=timeout, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_stages_contract(dev):
    """The N = 1 line carries every stage of the path beside the headline (round 5): k-NN, graph + Laplacian build, eigensolve,
    the wide SpMM, features in / out of sample, the kernel block, one supervised and one semi-supervised training epoch
    (`stages`), the reference benchmark's mv / grad / eigen legs at its own shape on the GPU and on the host cores
    (`reference_bench`, `cpu_baseline.reference_shape`; benchmark/bench_sparse_laplacian.py:15-34), the S5 k-NN
    (`roofline_hbm.knn`) and the conditioned 60k-shape workload (`manifold_784`).  Reduced sizes here (--nodes 8000
    --force-extras); the full-size line is the driver's."""
    line = _run_bench(["--steps", "3", "--warmup", "1", "--nodes", "8000", "--force-extras"])
    st = line["stages"]
    for key in ("knn", "graph_symmetrise_csr_tiles", "laplacian_build", "eigensolve", "spmm_wide_C128", "spmm_wide_C100",
                "features_insample", "features_oos", "kernel_block", "train_epoch_supervised", "train_epoch_semisupervised"):
        assert key in st, key
    assert st["knn"]["ms"] > 0 and st["knn"]["effective_fp32_tflops"] > 0 and st["knn"]["stats"]["rows_redone_exact"] == 0
    assert st["laplacian_build"]["bytes"] == 24 * 2 * line["config"]["edges"] + 16 * line["config"]["nodes"]
    assert st["eigensolve"]["ms"] > 0 and st["eigensolve"]["spmm_applies"] > 0 and st["eigensolve"]["modes"] == 100
    for c in (128, 100):
        w = st["spmm_wide_C%d" % c]
        assert w["us"] > 0 and abs(w["frac"] - w["gbs"] / 8000.0) < 1e-3
    kb = st["kernel_block"]
    assert kb["tflops"] > 0 and abs(kb["frac"] - kb["tflops"] / kb["peak"]) < 2e-3
    for key in ("train_epoch_supervised", "train_epoch_semisupervised"):
        ep = st[key]
        assert ep["epoch_ms"] > 0 and np.isfinite(ep["first_loss"]) and np.isfinite(ep["last_loss"])
    assert st["train_epoch_semisupervised"]["labelled"] == 800
    rb = line["reference_bench"]
    for key in ("gpu_matvec_ms", "cpu_matvec_ms", "gpu_grad_backward_ms", "cpu_grad_backward_ms", "gpu_dense_symeig_ms",
                "gpu_block_eigensolver_100_modes_ms", "cpu_dense_symeig_ms"):
        assert rb[key] > 0, key
    assert rb["matvec_max_rel_diff"] < 1e-5 and rb["grad_rel_diff"] < 1e-3 and rb["eig_100_max_abs_diff"] < 1e-3
    rs = line["cpu_baseline"]["reference_shape"]
    assert rs["mv_ms"] == rb["cpu_matvec_ms"] and rs["grad_ms"] == rb["cpu_grad_backward_ms"] and rs["eigen_ms"] == rb["cpu_dense_symeig_ms"]
    hb = line["roofline_hbm"]
    assert hb["knn"]["pair_distances_per_s"] > 0 and hb["measured"].startswith("live_back_to_back")
    pc = hb["cg_solve"]["preconditioner"]
    for key in ("none", "jacobi"):
        assert pc[key]["iterations"] > 0 and pc[key]["ms"] > 0
    assert any(k.startswith("chebyshev_") for k in pc)
    m7 = line["manifold_784"]
    assert m7["eval_warm_after_1pct_bandwidth_change"]["block_products"] <= m7["eval_cold"]["block_products"]
    assert m7["posterior_test_rmse"] < 0.5 and m7["precision_cg"]["rel_residual"] <= 1e-6 and m7["eval_eigensolve_ms"] > 0


@pytest.mark.parametrize("workload", ["c3", "s5"])
def test_bench_distributed_world1_line_contract(dev, workload):
    """What `bench.py --gpus N` runs for N > 1 (parallel.bench_distributed: graph padding, row partition, RCCL communicator,
    partitioned plan, sharded columns) in a fresh process at world 1 (MGP_FORCE_DIST=1), so that the first multi-rank run
    cannot fail on plumbing: the N > 1 line's contract -- driver keys, `roofline`, `cpu_baseline`, `rccl_ranks.consistent`,
    `cg_multi_rhs_sharded` (C3), the single-GPU plan's time in the same process -- and a converged solve.  c3: the pipelined
    recurrence in chunks of 4; s5: Chronopoulos-Gear + refinement on the true residual."""
    args = ["--steps", "3", "--warmup", "1", "--workload", workload, "--nodes", "8000" if workload == "c3" else "200000"]
    line = _run_bench(args, env_extra={"MGP_FORCE_DIST": "1"})
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "rccl_ranks", "ms_per_step_single_gpu_plan"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["scaling"] == "strong" and line["value"] > 0
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    rc = line["rccl_ranks"]
    assert rc["consistent"] is True and rc["comm_count"] == [1] and rc["user_rank"] == [0]
    cb = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    cfg = line["config"]
    assert cfg["cg_iters"] > 0 and cfg["cg_true_residual_fp32_apply"] < (1e-5 if workload == "c3" else 2e-3)
    assert line["ms_per_step_single_gpu_plan"] > 0 and line["single_gpu_plan"]["iterations"] > 0
    if workload == "c3":
        sh = line["cg_multi_rhs_sharded"]
        assert sh["columns"] == 100 and sh["solve_ms"] > 0 and sh["true_mean_rel_residual"] < 1e-2
        assert "pipelined" in cfg["parallelism"]
    else:
        assert "Chronopoulos-Gear" in cfg["parallelism"] and cfg["cg_iters"] > 20
