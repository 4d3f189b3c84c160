"""GraphLaplacianOperator on MI355X: same constructor, cached properties and hooks as the reference
(manifold_gp/operators/graph_laplacian_operator.py:24-157); the arithmetic runs in libmgp_hip.

  reference                                   here
  ----------------------------------------   -------------------------------------------------
  exp / scatter_add_ x4 / gathers (:52-106)   mgp_laplacian_build: 3 fused gather-only CSR passes
  2 x torch_sparse.spmm + elementwise (:108)  mgp_laplacian_matmul: one fused SpMM launch
  super().diagonalization (:132-144)          mgp_lanczos_smallest (dense symeig kept for tiny N)
  out_of_sample [T,k,m] temporary (:146-157)  fused weights -> gather -> sum kernel
"""
import ctypes

import torch

from .. import _lib
from .._compat import LinearOperator, settings
from .._lib import check, lib, ptr, stream
from ..graph import KnnGraph, LaplacianData, graph_for_coo


def _scalar(t):
    return float(t.reshape(-1)[0].item()) if torch.is_tensor(t) else float(t)


class GraphLaplacianOperator(LinearOperator):
    def __init__(self, x, idx, operator_dimension, graphbandwidth, normalization="randomwalk",
                 self_loops=True, transposed=False, graph=None):
        super().__init__(x, idx=idx, operator_dimension=operator_dimension, graphbandwidth=graphbandwidth,
                         normalization=normalization, self_loops=self_loops, transposed=transposed)
        if normalization not in ("symmetric", "randomwalk"):
            raise ValueError("normalization must be 'symmetric' or 'randomwalk'")
        _lib.require_device(x, idx)
        self.x = x
        self.idx = idx
        self.operator_dimension = int(operator_dimension)
        self.graphbandwidth = graphbandwidth
        self.normalization = normalization
        self.self_loops = self_loops
        self.transposed = transposed
        self._graph = graph if isinstance(graph, KnnGraph) else None
        self._data = None

    # ---- device data (built lazily, once per operator = once per graph bandwidth value)
    @property
    def graph(self):
        if self._graph is None:
            self._graph = graph_for_coo(self.idx, self.x, self.operator_dimension)
        return self._graph

    @property
    def data(self):
        if self._data is None:
            self._data = LaplacianData(self.graph, _scalar(self.graphbandwidth), self.self_loops)
        return self._data

    # ---- the reference's cached properties (graph_laplacian_operator.py:52-106)
    @property
    def adjacency_unnorm_mat(self):
        return self.data.edge_values(0)

    @property
    def degree_unnorm_mat(self):
        return self.data.degree_unnorm

    @property
    def adjacency_mat(self):
        return self.data.edge_values(1)

    @property
    def degree_mat(self):
        return self.data.degree

    @property
    def laplacian_diag(self):
        return self.data.diag

    @property
    def laplacian_triu(self):
        return self.data.edge_values(2)

    def _diagonal(self):
        return self.laplacian_diag

    # ---- hooks
    def _mode(self):
        if self.normalization == "symmetric":
            return 0
        return 2 if self.transposed else 1

    def _hyper_tensors(self):
        return [self.graphbandwidth]

    def _matmul_grad(self, rhs):
        """Differentiable path (autograd.py): gradients wrt rhs and the graph bandwidth."""
        from ..autograd import fused_spmm, node_vector
        d, eps = self.data, self.graphbandwidth
        pre = post = None
        if self.normalization == "randomwalk":
            sq, isq = node_vector(eps, d, "dsqrt"), node_vector(eps, d, "dinvsqrt")
            pre, post = (isq, sq) if self.transposed else (sq, isq)
        return fused_spmm(d, rhs, eps, a=0.0, b=1.0, pre=pre, post=post)

    def _matmul(self, rhs):
        _lib.require_device(rhs)
        from ..autograd import needs_grad
        if needs_grad(rhs, self.graphbandwidth):
            return self._matmul_grad(rhs)
        squeeze = rhs.dim() == 1
        X = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
        d = self.data
        n = self.operator_dimension
        if X.shape[0] != n:
            raise RuntimeError("shape mismatch: operator is %d x %d, rhs has %d rows" % (n, n, X.shape[0]))
        check(lib().mgp_spmm_set_group_hint(self.graph.spmv_lanes), "mgp_spmm_set_group_hint")
        csr = d.csr(wide=X.shape[1] >= 48)
        out = torch.empty_like(X)
        for c0 in range(0, X.shape[1], 256):
            Xc = X if X.shape[1] <= 256 else X[:, c0:c0 + 256].contiguous()
            Yc = out if X.shape[1] <= 256 else torch.empty_like(Xc)
            check(lib().mgp_laplacian_matmul(ctypes.byref(csr), ptr(d.dsqrt), ptr(d.dinvsqrt), self._mode(),
                                             ptr(Xc), Xc.shape[1], ptr(Yc), None, stream()),
                  "mgp_laplacian_matmul")
            if Yc is not out:
                out[:, c0:c0 + 256] = Yc
        return out.squeeze(-1) if squeeze else out

    def _size(self):
        return torch.Size([self.operator_dimension, self.operator_dimension])

    def _transpose_nonbatch(self):
        if self.normalization != "randomwalk":
            return self
        op = GraphLaplacianOperator(self.x, self.idx, self.operator_dimension, self.graphbandwidth,
                                    self.normalization, self.self_loops, not self.transposed, graph=self._graph)
        op._data = self._data
        return op

    def _symmetric_twin(self):
        op = GraphLaplacianOperator(self.x, self.idx, self.operator_dimension, self.graphbandwidth, "symmetric",
                                    self.self_loops, graph=self._graph)
        op._data = self._data        # the CSR values are those of L_sym for both normalisations
        return op

    # ---- graph_laplacian_operator.py:132-144
    def diagonalization(self, method=None, num_modes=None):
        from ..solvers import dense_symeig, lanczos_smallest
        n = self.operator_dimension
        if self.normalization == "symmetric":
            m = num_modes if num_modes is not None and num_modes < n else n
            use_dense = method == "symeig" or (method is None and n <= settings.max_cholesky_size.value())
            if use_dense:
                evals, evecs = dense_symeig(self)
                evals, evecs = evals[:m].clone(), evecs[:, :m].contiguous()
            else:
                evals, evecs, _ = lanczos_smallest(self.data, m)
            evals[0] = 0.0
            return evals, evecs
        evals, evecs = self._symmetric_twin().diagonalization(method, num_modes)
        evecs = evecs * self.degree_mat.pow(-0.5).view(-1, 1)
        evecs = torch.nn.functional.normalize(evecs, p=2, dim=0)
        return evals, evecs

    # ---- graph_laplacian_operator.py:146-157, unfused form kept for API parity; the kernel's
    # feature path uses the fused HIP kernel (kernels/riemann_kernel.py)
    def out_of_sample(self, x, edge_value, edge_idx):
        _lib.require_device(x, edge_value, edge_idx)
        T, k = edge_idx.shape
        m = x.shape[1]
        ones = torch.zeros(m, device=x.device)          # eigenvalues 0 -> sqrt-density is a constant
        Z = torch.empty(T, m, device=x.device, dtype=torch.float32)
        d = self.data
        # big support so that no point is masked, bump decay 0 -> modulation 1
        check(lib().mgp_features_oos(ptr(ones), ptr(_lib.f32c(x)), self.operator_dimension, m, 1, 1.0, d.eps,
                                     0 if self.normalization == "symmetric" else 1, ptr(d.degree_unnorm),
                                     ptr(d.degree), ptr(_lib.f32c(edge_value)),
                                     ptr(edge_idx.to(torch.int32).contiguous()), T, k, 3.0e38 / max(d.eps, 1e-30),
                                     0.0, ptr(Z), stream()), "mgp_features_oos")
        # the fused kernel multiplies by sqrt(N * s_j / sum s) = sqrt(N / m); undo it
        return Z * (m / float(self.operator_dimension)) ** 0.5
