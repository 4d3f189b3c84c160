"""CPU ORACLE (test infrastructure) -- the same operators as oracle/laplacian.py + oracle/precision.py, assembled as
scipy CSR matrices in float64 so that the at-size checks (N = 10k ... 1M) finish in seconds.

Nothing new is defined here: every entry comes from a `LaplacianOracle` (graph_laplacian_operator.py:52-106:
diag, triu = S, degree), the products restate graph_laplacian_operator.py:108-124, precision_matern_operator.py:26-37
and schur_complement_operator.py:26-30.  tests/test_oracle_golden.py pins this file against the matrix-free oracle and
the reference-generated goldens.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def laplacian_sym_csr(lap):
    """L_sym = diag - S - S^T (float64 CSR) from a LaplacianOracle."""
    n = lap.n
    r, c = lap.idx[0], lap.idx[1]
    s = lap.triu.astype(np.float64)
    S = sp.coo_matrix((s, (r, c)), shape=(n, n)).tocsr()
    return (sp.diags(lap.diag.astype(np.float64)) - S - S.T).tocsr()


class SparsePrecision:
    """Q = (2 nu / kappa^2 I + L)^nu (x D for random walk), optionally scaled: float64 matvecs on the CSR of L_sym.
    random walk: L_rw = D^-1/2 L_sym D^1/2, D (tau I + L_rw)^nu = D^1/2 (tau I + L_sym)^nu D^1/2."""

    def __init__(self, lap, nu, lengthscale, scale=1.0):
        self.lap, self.nu, self.n = lap, int(nu), lap.n
        self.L = laplacian_sym_csr(lap)
        self.tau = 2.0 * self.nu / (float(lengthscale) ** 2)
        self.scale = float(scale)
        self.dsq = np.sqrt(lap.degree.astype(np.float64)) if lap.normalization == "randomwalk" else None

    def laplacian_matmul(self, v, transposed=False):
        v = np.asarray(v, np.float64)
        if self.dsq is None:
            return self.L @ v
        d = self.dsq if v.ndim == 1 else self.dsq[:, None]
        return (self.L @ (v / d)) * d if transposed else (self.L @ (v * d)) / d

    def matmul(self, v):
        v = np.asarray(v, np.float64)
        d = None if self.dsq is None else (self.dsq if v.ndim == 1 else self.dsq[:, None])
        out = v * d if d is not None else v
        for _ in range(self.nu):
            out = self.tau * out + self.L @ out
        if d is not None:
            out = out * d
        return self.scale * out

    def posterior_system(self, v, noise):
        """(I + noise Q) v: the precision form of K (K + noise I)^-1."""
        return v + noise * self.matmul(v)

    def solve(self, b, matvec=None, tol=1e-13, maxiter=20000, mask=None):
        """fp64 CG solve with this operator (or `matvec`); mask: restrict to a principal block."""
        b = np.asarray(b, np.float64)
        mv = matvec or self.matmul
        if mask is not None:
            idx = np.nonzero(mask)[0]

            def mv_block(z):
                full = np.zeros(self.n)
                full[idx] = z
                return mv(full)[idx]
            A = spla.LinearOperator((idx.size, idx.size), matvec=mv_block, dtype=np.float64)
        else:
            A = spla.LinearOperator((self.n, self.n), matvec=mv, dtype=np.float64)
        cols = b.reshape(b.shape[0], -1)
        out = np.empty_like(cols)
        for j in range(cols.shape[1]):
            x, info = spla.cg(A, cols[:, j], rtol=tol, atol=0.0, maxiter=maxiter)
            if info != 0:
                raise RuntimeError("oracle CG did not converge (info %d)" % info)
            out[:, j] = x
        return out.reshape(b.shape)

    def schur_matmul(self, v, mask):
        """Q_ll v - Q_lu Q_uu^-1 Q_ul v (schur_complement_operator.py:26-30), inner solve by fp64 CG."""
        mask = np.asarray(mask, bool)
        v = np.asarray(v, np.float64)
        full = np.zeros(self.n)
        full[mask] = v
        tmp = self.matmul(full)
        sol = self.solve(tmp[~mask], mask=~mask)
        full2 = np.zeros(self.n)
        full2[~mask] = sol
        return tmp[mask] - self.matmul(full2)[mask]


def smallest_eigenvalues(lap, m):
    """m smallest eigenvalues of L_sym by dense float64 eigvalsh (N up to ~10k) -- riemann_kernel.py:121-125."""
    L = laplacian_sym_csr(lap).toarray()
    return np.linalg.eigvalsh(L)[:m]
