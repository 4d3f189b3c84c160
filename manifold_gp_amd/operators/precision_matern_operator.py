"""PrecisionMaternOperator: Q = (2 nu / kappa^2 I + L)^nu (x D for random walk), reference
manifold_gp/operators/precision_matern_operator.py:10-53, applied as nu fused SpMM launches."""
import numpy as np
import torch

from .._compat import LinearOperator
from .._lib import host_scalar
from ._descriptor import Descriptor


class PrecisionMaternOperator(LinearOperator):
    def __init__(self, laplacian, nu, lengthscale):
        super().__init__(laplacian, nu=nu, lengthscale=lengthscale)
        self.laplacian = laplacian
        self.nu = nu
        self.lengthscale = lengthscale

    def _descriptor(self):
        d = self.laplacian.data
        sq = d.dsqrt if self.laplacian.normalization == "randomwalk" else None
        # D (tau I + L_rw)^nu = D^1/2 (tau I + L_sym)^nu D^1/2   (precision_matern_operator.py:30-37)
        return Descriptor(d, int(self.nu), host_scalar(self, "lengthscale"), pre=sq, post=sq)

    def _hyper_tensors(self):
        return self.laplacian._hyper_tensors() + [self.lengthscale]

    def _matmul_grad(self, rhs):
        """Differentiable chain: gradients wrt rhs, graph bandwidth and lengthscale."""
        from ..autograd import fused_spmm, node_vector
        lap = self.laplacian
        d, eps = lap.data, lap.graphbandwidth
        ls = self.lengthscale.reshape(()) if torch.is_tensor(self.lengthscale) else torch.tensor(float(self.lengthscale))
        tau = 2.0 * self.nu / ls.to(d.graph.device).square()
        # the same value on the host, in the float32 steps torch takes on the device (scalar / tensor = reciprocal * scalar): the
        # launches below take it by value, and reading `tau` back would be a host wait per launch
        ls32 = np.float32(host_scalar(self, "lengthscale"))
        tau_host = float((np.float32(1.0) / (ls32 * ls32)) * np.float32(2.0 * self.nu))
        sq = node_vector(eps, d, "dsqrt") if lap.normalization == "randomwalk" else None
        out = rhs
        for s in range(self.nu):
            out = fused_spmm(d, out, eps, a=tau, b=1.0, pre=sq if s == 0 else None,
                             post=sq if s == self.nu - 1 else None, a_value=tau_host)
        return out

    def _matmul(self, rhs):
        from ..autograd import needs_grad
        if needs_grad(rhs, *self._hyper_tensors()):
            return self._matmul_grad(rhs)
        return self._descriptor().apply(rhs)

    def _size(self):
        return self.laplacian._size()

    def _transpose_nonbatch(self):
        return self

    def _solve_hip(self, rhs):
        from ..solvers import cg_solve
        return cg_solve(self._descriptor(), rhs)[0]

    def _average_variance(self, num_rand_vec=100):
        """precision_matern_operator.py:45-53: mean_i (Q^-1)_ii over random one-hot columns
        (indices drawn from [0, d-2] with replacement, as the reference does)."""
        d = self.shape[0]
        dev = self.laplacian.x.device
        if num_rand_vec >= d:
            rand_vec = torch.eye(d, device=dev)
        else:
            rand_idx = torch.randint(0, d - 1, (1, num_rand_vec), device=dev)
            rand_vec = torch.zeros(d, num_rand_vec, device=dev).scatter_(0, rand_idx, 1.0)
        return self.inv_quad_logdet(inv_quad_rhs=rand_vec, logdet=False)[0] / rand_vec.shape[1]
