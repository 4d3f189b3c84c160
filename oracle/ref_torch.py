"""CPU ORACLE (test infrastructure) -- the reference's sparse operator restated in torch-CPU.

This is the "reference CPU path" timed by bench.py's cpu_baseline leg (kind "port"): the body
of manifold_gp/operators/graph_laplacian_operator.py:108-124 with torch_sparse.spmm expanded to
the three torch ops it is composed of (index_select -> mul -> index_add_), fp32 values, int64
upper-triangular COO indices, all host cores, driven the way
benchmark/bench_sparse_laplacian.py:15-19,63-64 drives it (v = torch.rand(N), seed 1337).
"""
import torch


class TorchCooLaplacian:
    def __init__(self, idx, triu, diag, degree, normalization="symmetric", transposed=False):
        self.idx = torch.as_tensor(idx, dtype=torch.int64)
        self.triu = torch.as_tensor(triu, dtype=torch.float32)
        self.diag = torch.as_tensor(diag, dtype=torch.float32)
        self.degree = torch.as_tensor(degree, dtype=torch.float32)
        self.normalization = normalization
        self.transposed = transposed
        self.n = self.diag.shape[0]

    def _spmm(self, row, col, vec):
        # torch_sparse.spmm(index, value, m, n, matrix): gather, scale, scatter-add
        out = vec.index_select(0, col) * self.triu.view(-1, 1)
        return torch.zeros_like(vec).index_add_(0, row, out)

    def matmul(self, rhs):                       # graph_laplacian_operator.py:108-124
        squeeze = rhs.dim() == 1
        if squeeze:
            rhs = rhs.view(-1, 1)
        if self.normalization == "randomwalk":
            vec = rhs.contiguous().div(self.degree.pow(0.5).view(-1, 1)) if self.transposed \
                else rhs.contiguous() * self.degree.pow(0.5).view(-1, 1)
        else:
            vec = rhs.contiguous()
        out = vec * self.diag.view(-1, 1)
        out -= self._spmm(self.idx[0], self.idx[1], vec)
        out -= self._spmm(self.idx[1], self.idx[0], vec)
        if self.normalization == "randomwalk":
            out *= self.degree.pow(0.5).view(-1, 1) if self.transposed else self.degree.pow(-0.5).view(-1, 1)
        return out.view(-1) if squeeze else out


class TorchPrecision:
    """precision_matern_operator.py:26-37 on top of TorchCooLaplacian, plus the affine
    posterior-mean system A = I + noise * outputscale * Q used by the CG headline."""

    def __init__(self, lap, nu, lengthscale, outputscale=1.0, noise=0.0):
        self.lap, self.nu, self.ls = lap, int(nu), float(lengthscale)
        self.outputscale, self.noise = float(outputscale), float(noise)

    def q(self, rhs):
        out = rhs.contiguous()
        diag = self.ls * self.ls / (2 * self.nu)
        for _ in range(self.nu):
            out = out + diag * self.lap.matmul(out)
            out = out / diag
        if self.lap.normalization == "randomwalk":
            out = out * (self.lap.degree.view(-1, 1) if out.dim() == 2 else self.lap.degree)
        return out

    def posterior_system(self, rhs):
        return rhs + (self.noise * self.outputscale) * self.q(rhs)


def torch_cg(matmul, b, tol, max_iter, jacobi=None):
    """Plain torch CG with a relative-residual stop (same rule as the HIP solver's tight
    mode): used for the CPU CG baseline timing."""
    x = torch.zeros_like(b)
    r = b.clone()
    z = r * jacobi if jacobi is not None else r
    p = z.clone()
    rz = torch.dot(r, z)
    bn = b.norm()
    it = 0
    for it in range(1, max_iter + 1):
        q = matmul(p)
        alpha = rz / torch.dot(p, q)
        x += alpha * p
        r -= alpha * q
        if r.norm() <= tol * bn:
            break
        z = r * jacobi if jacobi is not None else r
        rz_new = torch.dot(r, z)
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it


def torch_laplacian_from_edges(val, idx, n, eps, self_loops=True):
    """graph_laplacian_operator.py:52-106 in differentiable torch ops (exp, scatter_add, gathers): the
    reference's cached properties as functions of the bandwidth `eps` (a tensor that may require grad).
    Returns (diag [n], triu [M], degree [n]) of L_sym, i.e. what TorchCooLaplacian takes."""
    val = torch.as_tensor(val, dtype=torch.float32)
    idx = torch.as_tensor(idx, dtype=torch.int64)
    w = torch.exp(-val / (4.0 * eps * eps))                                   # :56
    ones = torch.ones(n) if self_loops else torch.zeros(n)
    dt = ones.index_add(0, idx[0], w).index_add(0, idx[1], w)                 # :62-69
    a = w / (dt[idx[0]] * dt[idx[1]])                                         # :75
    self_a = dt.pow(-2) if self_loops else torch.zeros(n)
    deg = self_a.index_add(0, idx[0], a).index_add(0, idx[1], a)              # :81-88
    diag = (1.0 - self_a / deg) / (eps * eps) if self_loops else torch.ones(n) / (eps * eps)   # :94-97
    triu = a / (deg[idx[0]].sqrt() * deg[idx[1]].sqrt()) / (eps * eps)        # :105-106 (stored positive)
    return diag, triu, deg


def dense_model_precision(val, idx, n, eps, kappa, outputscale, noise, nu, normalization="randomwalk", self_loops=True):
    """The model precision of riemann_gp.py:32-39 as a DENSE differentiable float64 matrix (test oracle for gradients):
    L from torch_laplacian_from_edges (graph_laplacian_operator.py:52-124), Q = (2 nu / kappa^2 I + L)^nu (x D for
    random walk, precision_matern_operator.py:26-37), Q2 = outputscale * Q (scale_wrapper_operator.py:27),
    Q3 = Q2 - noise Q2^2 + noise^2 Q2^3 (noise_wrapper_operator.py:22).  eps, kappa, outputscale, noise: 0-d float64
    tensors (may require grad)."""
    val = torch.as_tensor(val, dtype=torch.float64)
    idx = torch.as_tensor(idx, dtype=torch.int64)
    w = torch.exp(-val / (4.0 * eps * eps))
    base = torch.ones(n, dtype=torch.float64) if self_loops else torch.zeros(n, dtype=torch.float64)
    dt = base.index_add(0, idx[0], w).index_add(0, idx[1], w)
    a = w / (dt[idx[0]] * dt[idx[1]])
    self_a = dt.pow(-2) if self_loops else torch.zeros(n, dtype=torch.float64)
    deg = self_a.index_add(0, idx[0], a).index_add(0, idx[1], a)
    diag = (1.0 - self_a / deg) / (eps * eps) if self_loops else torch.ones(n, dtype=torch.float64) / (eps * eps)
    triu = a / (deg[idx[0]].sqrt() * deg[idx[1]].sqrt()) / (eps * eps)
    S = torch.zeros(n, n, dtype=torch.float64).index_put((idx[0], idx[1]), triu, accumulate=True)
    Lsym = torch.diag(diag) - S - S.t()
    if normalization == "randomwalk":
        ds = deg.sqrt()
        L = Lsym * ds.view(1, -1) / ds.view(-1, 1)             # D^-1/2 L_sym D^1/2
    else:
        L = Lsym
    Q = torch.linalg.matrix_power(torch.eye(n, dtype=torch.float64) * (2.0 * nu / (kappa * kappa)) + L, nu)
    if normalization == "randomwalk":
        Q = deg.view(-1, 1) * Q
    Q2 = Q * outputscale
    return Q2 - noise * (Q2 @ Q2) + noise * noise * (Q2 @ Q2 @ Q2)
