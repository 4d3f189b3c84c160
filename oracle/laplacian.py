"""CPU ORACLE (test infrastructure) -- diffusion-maps graph Laplacian on upper-tri COO edges.

Literal numpy restatement of manifold_gp/operators/graph_laplacian_operator.py:52-157.
Inputs are what the reference operator is constructed from: edge squared distances
``x`` ([M], the reference names them `x`), ``idx`` ([2,M] i64, row<col), the node count,
the graph bandwidth eps, normalization in {"symmetric","randomwalk"}, self_loops.
``dtype`` float32 reproduces the reference precision; float64 is the converged oracle.
"""
import numpy as np


def _scatter_add(base, index, src):
    out = base.copy()
    np.add.at(out, index, src)
    return out


class LaplacianOracle:
    def __init__(self, x, idx, operator_dimension, graphbandwidth, normalization="randomwalk",
                 self_loops=True, transposed=False, dtype=np.float32):
        self.dtype = dtype
        self.x = np.asarray(x, dtype=dtype)
        self.idx = np.asarray(idx, dtype=np.int64)
        self.n = int(operator_dimension)
        self.eps = dtype(graphbandwidth)
        self.normalization = normalization
        self.self_loops = self_loops
        self.transposed = transposed
        dt = dtype
        r, c = self.idx[0], self.idx[1]
        eps2 = self.eps * self.eps
        # :54-56  W = exp(-x / (4 eps^2))
        self.adjacency_unnorm = np.exp(self.x / (dt(-4) * eps2)).astype(dt)
        # :60-69  Dtilde = [self_loops] + sum_row W + sum_col W
        base = np.ones(self.n, dt) if self_loops else np.zeros(self.n, dt)
        self.degree_unnorm = _scatter_add(_scatter_add(base, r, self.adjacency_unnorm), c,
                                          self.adjacency_unnorm)
        # :73-75  A = W / (Dtilde_i Dtilde_j)
        self.adjacency = (self.adjacency_unnorm /
                          (self.degree_unnorm[r] * self.degree_unnorm[c])).astype(dt)
        # :79-88  D = [self_loops] Dtilde^-2 + sum_row A + sum_col A
        base = (self.degree_unnorm ** dt(-2)).astype(dt) if self_loops else np.zeros(self.n, dt)
        self.degree = _scatter_add(_scatter_add(base, r, self.adjacency), c, self.adjacency)
        # :92-97  diag = (1 - Dtilde^-2 / D) / eps^2   or  1/eps^2
        if self_loops:
            self.diag = ((dt(1) - (self.degree_unnorm ** dt(-2)) * (self.degree ** dt(-1))) /
                         eps2).astype(dt)
        else:
            self.diag = (np.ones(self.n, dt) / eps2).astype(dt)
        # :104-106 S = A / (sqrt(D_i) sqrt(D_j)) / eps^2
        ds = np.sqrt(self.degree).astype(dt)
        self.triu = (self.adjacency / (ds[r] * ds[c]) / eps2).astype(dt)

    # graph_laplacian_operator.py:108-124
    def matmul(self, rhs, transposed=None):
        transposed = self.transposed if transposed is None else transposed
        rhs = np.asarray(rhs, dtype=self.dtype)
        squeeze = rhs.ndim == 1
        if squeeze:
            rhs = rhs[:, None]
        r, c = self.idx[0], self.idx[1]
        dsq = np.sqrt(self.degree).astype(self.dtype)[:, None]
        if self.normalization == "randomwalk":
            vec = rhs / dsq if transposed else rhs * dsq
        else:
            vec = rhs
        out = vec * self.diag[:, None]
        # torch_sparse.spmm(idx, val, n, n, vec) = scatter_add(val * vec[col], row)
        if vec.shape[1] > 64:
            # wide right-hand sides (dense twins of the wrappers: identity blocks of ~1500 columns): the same updates in
            # the same order per output row -- first the entries with this row as `r` in list order, then those with it as
            # `c` -- through a CSR product that accumulates into `out` entry by entry (np.subtract.at takes a minute here)
            out = self._wide_subtract(out, vec)
        else:
            np.subtract.at(out, r, self.triu[:, None] * vec[c])
            np.subtract.at(out, c, self.triu[:, None] * vec[r])
        if self.normalization == "randomwalk":
            out = out * dsq if transposed else out / dsq
        out = out.astype(self.dtype)
        return out[:, 0] if squeeze else out

    def _wide_subtract(self, out, vec):
        from scipy.sparse import _sparsetools
        if getattr(self, "_csr", None) is None:
            r, c = self.idx[0].astype(np.int64), self.idx[1].astype(np.int64)
            rows = np.concatenate([r, c])
            cols = np.concatenate([c, r])
            vals = np.concatenate([-self.triu, -self.triu]).astype(self.dtype)
            order = np.argsort(rows, kind="stable")               # list order kept inside a row, `r` entries first
            indptr = np.zeros(self.n + 1, np.int64)
            np.add.at(indptr, rows + 1, 1)
            self._csr = (np.cumsum(indptr).astype(np.int32), cols[order].astype(np.int32), np.ascontiguousarray(vals[order]))
        ip, ix, vx = self._csr
        y = np.ascontiguousarray(out, dtype=self.dtype)
        x = np.ascontiguousarray(vec, dtype=self.dtype)
        _sparsetools.csr_matvecs(self.n, self.n, x.shape[1], ip, ix, vx, x.ravel(), y.ravel())
        return y

    def dense_symmetric(self):
        """Dense L_sym as assembled at manifold_gp/kernels/riemann_kernel.py:121-124."""
        L = np.zeros((self.n, self.n), self.dtype)
        L[np.arange(self.n), np.arange(self.n)] = self.diag
        np.add.at(L, (self.idx[0], self.idx[1]), -self.triu)
        np.add.at(L, (self.idx[1], self.idx[0]), -self.triu)
        return L

    def dense(self, transposed=None):
        return self.matmul(np.eye(self.n, dtype=self.dtype), transposed)

    # graph_laplacian_operator.py:146-157
    def out_of_sample(self, phi, edge_value, edge_idx):
        dt = self.dtype
        out = np.exp(np.asarray(edge_value, dt) / (dt(-4) * self.eps * self.eps)).astype(dt)
        degree_test = out.sum(axis=1, dtype=dt)
        out = out / (self.degree_unnorm[edge_idx] * degree_test[:, None])
        if self.normalization == "symmetric":
            out = out / (np.sqrt(self.degree)[edge_idx] * np.sqrt(out.sum(axis=1, dtype=dt))[:, None])
        elif self.normalization == "randomwalk":
            out = out / out.sum(axis=1, dtype=dt)[:, None]
        out = out.astype(dt)
        return (out[:, :, None] * np.asarray(phi, dt)[edge_idx]).sum(axis=1, dtype=dt).astype(dt)
