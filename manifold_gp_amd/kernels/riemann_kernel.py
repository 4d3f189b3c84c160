"""RiemannKernel on MI355X: the gpytorch Kernel plugin of manifold_gp/kernels/riemann_kernel.py:25-149
with the same constructor, parameters, attributes and hooks.

  ctor     k-NN graph build  -> HIP exact k-NN + symmetrise kernels          (:40-42)
  eval()   dense N x N eigh  -> HIP Lanczos for the `num_modes` smallest pairs of L_sym; the
           reference's post-processing (lambda_0 = 0, D^-1/2 scaling, column normalisation, applied
           for BOTH normalisations) is kept                                    (:117-130)
  features in-sample / out-of-sample features -> fused HIP kernels              (:132-149)
  forward  low-rank root operators / dense block on the fp32 MFMA              (:79-100)
"""
from abc import abstractmethod

import torch

from .. import _lib
from .._compat import HAVE_GPYTORCH, Kernel, Positive, settings
from .._lib import check, lib, ptr, stream
from ..operators import GraphLaplacianOperator
from ..utils import NearestNeighbors


class SpectralRootOperator:
    """K = Z1 Z2^T kept lazy (what LowRankRootLinearOperator / MatmulLinearOperator are to the
    reference, riemann_kernel.py:92-100).  With linear_operator installed the real classes are
    returned instead."""

    def __init__(self, z1, z2=None):
        self.z1 = z1
        self.z2 = z1 if z2 is None else z2

    root = property(lambda self: self.z1)

    @property
    def shape(self):
        return torch.Size([self.z1.shape[0], self.z2.shape[0]])

    def to_dense(self):
        from ..solvers import kernel_block
        return kernel_block(self.z1, self.z2)

    evaluate = to_dense

    def matmul(self, rhs):
        return self.z1 @ (self.z2.t() @ rhs)

    __matmul__ = matmul

    def diagonal(self):
        from ..solvers import kernel_diag
        return kernel_diag(self.z1, self.z2)


class RiemannKernel(Kernel):
    has_lengthscale = True

    def __init__(self, x, nearest_neighbors=10, laplacian_normalization="symmetric", num_modes=100,
                 bump_scale=1.0, bump_decay=0.01, graphbandwidth_prior=None, graphbandwidth_constraint=None,
                 **kwargs):
        super(RiemannKernel, self).__init__(**kwargs)
        self.knn = NearestNeighbors(x, nlist=1)
        self.nearest_neighbors = nearest_neighbors
        self.edge_index, self.edge_value = self.knn.graph(self.nearest_neighbors, nprobe=1)
        self.laplacian_normalization = laplacian_normalization
        self.num_modes = num_modes
        self.bump_scale = bump_scale
        self.bump_decay = bump_decay
        # residual tolerance of eval()'s eigensolve relative to lambda_max.  1e-6 since round 3: against the reference
        # pipeline's float64 posterior on the dumbbell (tests/golden/dumbbell_posterior.npz) 1e-5 leaves 1.8e-4 in the
        # posterior mean and 5.5e-3 in (K + noise I)^-1 y, 1e-6 leaves 1.3e-6 / 5.2e-5 (tools/probe_posterior.py); where
        # fp32 cannot reach it the solver stops at its residual floor (include/mgp_hip.h)
        self.eigen_tol = 1e-6
        self.keep_eigen_block = False     # True: eval() keeps the solver's whole Rayleigh-Ritz block (guard columns) in .eigen_block
        self.warm_start = True            # eval() starts from the previous eval()'s block while the graph is the same object

        if graphbandwidth_constraint is None:
            graphbandwidth_constraint = Positive()
        self.register_parameter(name="raw_graphbandwidth",
                                parameter=torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, 1)))
        if graphbandwidth_prior is not None:
            if HAVE_GPYTORCH:
                from gpytorch.priors import Prior
                if not isinstance(graphbandwidth_prior, Prior):
                    raise TypeError("Expected gpytorch.priors.Prior but got " + type(graphbandwidth_prior).__name__)
            elif not hasattr(graphbandwidth_prior, "log_prob"):
                raise TypeError("Expected gpytorch.priors.Prior but got " + type(graphbandwidth_prior).__name__)
            self.register_prior("graphbandwidth_prior", graphbandwidth_prior, self._graphbandwidth_param,
                                self._graphbandwidth_closure)
        self.register_constraint("raw_graphbandwidth", graphbandwidth_constraint)

    def _graphbandwidth_param(self, m):
        return m.graphbandwidth

    def _graphbandwidth_closure(self, m, v):
        return m._set_graphbandwidth(v)

    def _set_graphbandwidth(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_graphbandwidth)
        self.initialize(raw_graphbandwidth=self.raw_graphbandwidth_constraint.inverse_transform(value))

    @property
    def graphbandwidth(self):
        return self.raw_graphbandwidth_constraint.transform(self.raw_graphbandwidth)

    @graphbandwidth.setter
    def graphbandwidth(self, value):
        self._set_graphbandwidth(value)

    # ------------------------------------------------------------------ riemann_kernel.py:79-100
    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, **kwargs):
        if last_dim_is_batch:
            x1 = x1.transpose(-1, -2).unsqueeze(-1)
            x2 = x2.transpose(-1, -2).unsqueeze(-1)
        x1_eq_x2 = x1.data_ptr() == x2.data_ptr() and x1.shape == x2.shape or torch.equal(x1, x2)
        z1 = self.features(x1)
        z2 = z1 if x1_eq_x2 else self.features(x2)
        if diag:
            from ..solvers import kernel_diag
            return kernel_diag(z1, z2)
        if HAVE_GPYTORCH:  # pragma: no cover
            from linear_operator.operators import (LowRankRootLinearOperator, MatmulLinearOperator,
                                                   RootLinearOperator)
            if x1_eq_x2:
                return LowRankRootLinearOperator(z1) if z1.size(-1) < z2.size(-2) else RootLinearOperator(z1)
            return MatmulLinearOperator(z1, z2.transpose(-1, -2))
        return SpectralRootOperator(z1, None if x1_eq_x2 else z2)

    @abstractmethod
    def spectral_density(self):
        raise NotImplementedError()

    # ------------------------------------------------------------------ :114-130
    def laplacian(self):
        dev = self.edge_value.device
        return GraphLaplacianOperator(self.edge_value, self.edge_index, self.knn.x.shape[0],
                                      self.graphbandwidth.to(dev), self.laplacian_normalization,
                                      graph=self.knn.knn_graph)

    def eval(self):
        self.laplacian_operator = self.laplacian()
        with torch.no_grad():
            from ..solvers import lanczos_smallest
            data = self.laplacian_operator.data
            n = self.laplacian_operator.operator_dimension
            m = min(self.num_modes, n)
            # warm start (round 5): the reference recomputes the whole decomposition on every eval() (:117-130); with the graph
            # unchanged and the bandwidth moved a little, the previous Rayleigh-Ritz block is a far better start than a random one
            gid = id(self.knn.knn_graph)
            warm = getattr(self, "_eigen_warm", None)
            warm = warm[1] if (self.warm_start and warm is not None and warm[0] == (gid, m)) else None
            out = lanczos_smallest(data, m, tol=self.eigen_tol, return_block=self.keep_eigen_block, warm=warm, keep_warm=self.warm_start)
            self.eigen_info = list(lanczos_smallest.last_info)
            # (a block that ended at the fp32 residual floor -- fewer than m pairs under the tolerance: the wanted modes sit in a
            # cluster -- is no better a start than a random one and keeps the solver from its floor exits: not kept)
            self._eigen_warm = ((gid, m), lanczos_smallest.last_warm) if (self.warm_start and self.eigen_info[2] >= m) else None
            evals, evecs, resid = out[:3]
            self.eigen_block = out[3] if self.keep_eigen_block else None
            self.eigen_residuals = resid
            evals[0] = 0.0                                              # :126
            work = torch.empty(int(lib().mgp_eigvec_postprocess_work_floats(m)), dtype=torch.float32,
                               device=evecs.device)
            check(lib().mgp_eigvec_postprocess(ptr(evecs), n, m, ptr(data.degree), ptr(work), stream()),
                  "mgp_eigvec_postprocess")                             # :127-128
            self.eigval, self.eigvec = evals, evecs
        return super().eval()

    # ------------------------------------------------------------------ :132-149
    def _is_train_inputs(self, x):
        kx = self.knn.x
        if x.shape != kx.shape:
            return False
        if x.data_ptr() == kx.data_ptr():
            return True
        return bool(torch.equal(x, kx))

    def _hyper(self):
        return (float(self.lengthscale.reshape(-1)[0].item()), float(self.graphbandwidth.reshape(-1)[0].item()))

    def features(self, x):
        _lib.require_device(x)
        n, m = self.eigvec.shape
        kappa, eps = self._hyper()
        dev = self.eigvec.device
        if self._is_train_inputs(x):
            Z = torch.empty(n, m, dtype=torch.float32, device=dev)
            check(lib().mgp_features_insample(ptr(self.eigval), ptr(self.eigvec), n, m, int(self.nu), kappa, ptr(Z),
                                              stream()), "mgp_features_insample")
            return Z
        edge_value, edge_index = self.knn.search(x, self.nearest_neighbors)
        T = x.shape[0]
        Z = torch.empty(T, m, dtype=torch.float32, device=dev)
        data = self.laplacian_operator.data
        check(lib().mgp_features_oos(ptr(self.eigval), ptr(self.eigvec), n, m, int(self.nu), kappa, eps,
                                     0 if self.laplacian_normalization == "symmetric" else 1,
                                     ptr(data.degree_unnorm), ptr(data.degree), ptr(edge_value),
                                     ptr(edge_index.to(torch.int32).contiguous()), T, int(self.nearest_neighbors),
                                     float(self.bump_scale), float(self.bump_decay), ptr(Z), stream()),
              "mgp_features_oos")
        return Z
