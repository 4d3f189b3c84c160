"""CPU ORACLE (test infrastructure) -- iterative solvers and GP posterior formulas.

linear_operator / gpytorch are absent (un-pinned third-party wheels; parity unpinned), so
`linear_cg` restates the published algorithm as summarised in SURVEY.md Appendix B, anchored on
the reference call sites precision_matern_operator.py:53, schur_complement_operator.py:28,
train_model.py:68, and the posterior is the dense fp64 textbook form of what
manifold_gp/models/riemann_gp.py:27-75 + gpytorch's ExactGP evaluate.
"""
import numpy as np


def linear_cg(matmul, rhs, tolerance=1e-2, max_iter=1000, min_iter=10, precond=None,
              stop_updating_after=1e-10, eps=1e-10):
    """linear_operator.utils.linear_cg restated: rhs columns are normalised by their 2-norm,
    iterate at least `min_iter`, stop when the MEAN over columns of the residual 2-norm
    < tolerance (relative because of the normalisation) or at max_iter; columns whose
    residual < stop_updating_after freeze.  Returns (solution, iterations, residual_norms)."""
    rhs = np.asarray(rhs)
    squeeze = rhs.ndim == 1
    if squeeze:
        rhs = rhs[:, None]
    dt = rhs.dtype
    rhs_norm = np.linalg.norm(rhs, axis=0, keepdims=True)
    rhs_norm = np.where(rhs_norm < eps, 1.0, rhs_norm).astype(dt)
    b = rhs / rhs_norm
    x = np.zeros_like(b)
    r = b - matmul(x)
    z = precond(r) if precond is not None else r.copy()
    p = z.copy()
    rz = (r * z).sum(axis=0, keepdims=True)
    it = 0
    rn = np.linalg.norm(r, axis=0)
    for it in range(1, max_iter + 1):
        q = matmul(p)
        pq = (p * q).sum(axis=0, keepdims=True)
        alpha = rz / np.where(np.abs(pq) < eps, eps, pq)
        active = (rn >= stop_updating_after)[None, :]
        alpha = np.where(active, alpha, 0).astype(dt)
        x = x + alpha * p
        r = r - alpha * q
        z = precond(r) if precond is not None else r
        rz_new = (r * z).sum(axis=0, keepdims=True)
        beta = rz_new / np.where(np.abs(rz) < eps, eps, rz)
        p = z + beta.astype(dt) * p
        rz = rz_new
        rn = np.linalg.norm(r, axis=0)
        if it >= min_iter and rn.mean() < tolerance:
            break
    x = x * rhs_norm
    return (x[:, 0] if squeeze else x), it, rn


def cg_tight(matmul, rhs, tol=1e-10, max_iter=10000, precond=None):
    """Plain fp64 (P)CG to a tight relative residual: converged answer for parity."""
    return linear_cg(matmul, np.asarray(rhs, np.float64), tolerance=tol, max_iter=max_iter,
                     min_iter=1, precond=precond)


def gp_posterior_lowrank(Z_train, y, Z_test, outputscale, noise, mean_const=0.0):
    """Dense fp64 GP posterior with K = outputscale * Z Z^T (riemann_kernel.py:92-100 wrapped
    by ScaleKernel) and Gaussian noise: mean, covariance at the test features.
    Solved by Woodbury with an m x m Cholesky -- what gpytorch does for
    LowRankRootAddedDiagLinearOperator."""
    Zt = np.asarray(Z_train, np.float64)
    Zs = np.asarray(Z_test, np.float64)
    y = np.asarray(y, np.float64) - mean_const
    m = Zt.shape[1]
    # (s Z Z^T + n I)^-1 = 1/n [I - Z (n/s I + Z^T Z)^-1 Z^T]
    C = np.eye(m) * (noise / outputscale) + Zt.T @ Zt
    Lc = np.linalg.cholesky(C)

    def solve(B):
        t = np.linalg.solve(Lc.T, np.linalg.solve(Lc, Zt.T @ B))
        return (B - Zt @ t) / noise

    alpha = solve(y)
    Kst = outputscale * (Zs @ Zt.T)
    mean = mean_const + Kst @ alpha
    cov = outputscale * (Zs @ Zs.T) - Kst @ solve(Kst.T)
    return mean, cov, alpha


def precision_posterior_mean(Q2_matmul, y, noise):
    """Posterior mean at the graph nodes in precision form: with K = Q2^-1,
    K (K + noise I)^-1 y = (I + noise Q2)^-1 y  (SURVEY.md Appendix A.8)."""
    def A(v):
        return v + noise * Q2_matmul(v)
    x, it, rn = cg_tight(A, y)
    return x, it, rn


def rmse_nll(error, cov):                                     # test_model.py:20-24
    error = np.asarray(error, np.float64)
    rmse = np.sqrt(np.mean(error ** 2))
    sign, logdet = np.linalg.slogdet(cov)
    iq = error @ np.linalg.solve(cov, error)
    nll = 0.5 * (iq + logdet + error.size * np.log(2 * np.pi)) / error.size
    return rmse, nll


def lanczos_tridiag_f64(matmul, q0, steps):
    """k-step Lanczos with full re-orthogonalisation in float64 (linear_operator.utils.lanczos_tridiag restated;
    third-party, parity unpinned): returns (alpha[steps], beta[steps - 1])."""
    q = np.asarray(q0, np.float64)
    q = q / np.linalg.norm(q)
    Q = [q]
    alpha, beta = [], []
    for j in range(steps):
        w = matmul(Q[j])
        a = float(Q[j] @ w)
        w = w - a * Q[j] - (beta[-1] * Q[j - 1] if j > 0 else 0.0)
        for _ in range(2):
            for qq in Q:
                w = w - (qq @ w) * qq
        alpha.append(a)
        b = float(np.linalg.norm(w))
        if j + 1 < steps:
            if b < 1e-12 * max(abs(a), 1e-300):
                break
            beta.append(b)
            Q.append(w / b)
    return np.array(alpha), np.array(beta[:len(alpha) - 1])


def slq_logdet_same_probes(matmul, Z, steps, fun=None):
    """Stochastic Lanczos quadrature of log det with GIVEN probes (columns of Z, ||z||^2 = n):
    (n / P) sum_p e_1^T log(fun(T_p)) e_1 -- the estimator linear_operator's StochasticLQ evaluates behind
    `inv_quad_logdet(logdet=True)` (train_model.py:68), restated in float64 so that a device estimate made with the
    SAME probes can be compared with it tightly (the Monte-Carlo error cancels)."""
    Z = np.asarray(Z, np.float64)
    n, P = Z.shape
    total = 0.0
    for p in range(P):
        a, b = lanczos_tridiag_f64(matmul, Z[:, p], steps)
        T = np.diag(a) + np.diag(b, 1) + np.diag(b, -1)
        theta, S = np.linalg.eigh(T)
        theta = fun(theta) if fun is not None else theta
        total += float(np.sum(S[0, :] ** 2 * np.log(np.maximum(theta, 1e-300))))
    return n * total / P
