"""NoiseWrapperOperator (manifold_gp/operators/noise_wrapper_operator.py:8-28):
Q (v - s Q (v - s Q v)) = (Q - s Q^2 + s^2 Q^3) v, the 2nd-order Neumann form of (Q^-1 + s I)^-1."""
import torch

from .._compat import LinearOperator


def _scalar(t):
    return float(t.reshape(-1)[0].item()) if torch.is_tensor(t) else float(t)


class NoiseWrapperOperator(LinearOperator):
    def __init__(self, operator, noise):
        super().__init__(operator, noise=noise)
        self.operator = operator
        self.noise = noise

    def _descriptor(self):
        inner = getattr(self.operator, "_descriptor", lambda: None)()
        if inner is None or inner.form != 0:
            return None
        return inner.with_(form=1, noise=_scalar(self.noise))

    def _hyper_tensors(self):
        return getattr(self.operator, "_hyper_tensors", lambda: [])() + [self.noise]

    def _matmul(self, rhs):
        from ..autograd import needs_grad
        if needs_grad(rhs, *self._hyper_tensors()):
            Q = self.operator._matmul                        # noise_wrapper_operator.py:22, differentiable
            rhs = rhs.contiguous()
            return Q(rhs - self.noise * Q(rhs - self.noise * Q(rhs)))
        d = self._descriptor()
        if d is not None:
            return d.apply(rhs)
        s = _scalar(self.noise)
        Q = self.operator._matmul
        rhs = rhs.contiguous()
        return Q(rhs - s * Q(rhs - s * Q(rhs)))

    def _size(self):
        return self.operator._size()

    def _transpose_nonbatch(self):
        return NoiseWrapperOperator(self.operator._transpose_nonbatch(), self.noise)

    def _solve_hip(self, rhs):
        d = self._descriptor()
        if d is not None:
            from ..solvers import cg_solve
            return cg_solve(d, rhs)[0]
        # Not one polynomial chain (a Schur complement underneath): every matvec of this operator is three nested
        # inner solves.  It is the second-order Neumann form of (Q^-1 + s I)^-1: with M = Q^-1 + s I -- ONE solve with
        # the wrapped operator, which the Schur complement answers with a single non-nested CG on the full
        # precision -- the spectrum of M A is 1 + (s q)^3, within [1, 1.07] for s |Q| ~ 0.4.  Preconditioned by M and
        # started from M b the CG needs one or two iterations (8 solves) where the cold, unpreconditioned
        # recurrence took linear_cg's 10+ iterations and the start from M b alone 17 (each 3 nested solves).
        from ..solvers import generic_cg
        s = _scalar(self.noise)
        inner = self.operator

        def M(v):
            with torch.no_grad():
                return inner._solve(v) + s * v
        return generic_cg(self, rhs, x0=M(rhs), precond=M)
