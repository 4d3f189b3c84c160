"""ScaleWrapperOperator (manifold_gp/operators/scale_wrapper_operator.py:12-34): Q * s or Q / s.
When the wrapped operator is a polynomial chain the scale rides in the last SpMM's epilogue."""
import torch

from .._compat import LinearOperator
from .._lib import host_scalar


class ScaleWrapperOperator(LinearOperator):
    def __init__(self, operator, scale, inverse_scale=False):
        super().__init__(operator, scale=scale, inverse_scale=inverse_scale)
        self.operator = operator
        self.scale = scale
        self.inverse_scale = inverse_scale

    def _factor(self):
        s = host_scalar(self, "scale")
        return 1.0 / s if self.inverse_scale else s

    def _descriptor(self):
        inner = getattr(self.operator, "_descriptor", lambda: None)()
        if inner is None or inner.form != 0:
            return None
        return inner.with_(scale=inner.scale * self._factor())

    def _hyper_tensors(self):
        return getattr(self.operator, "_hyper_tensors", lambda: [])() + [self.scale]

    def _matmul(self, rhs):
        from ..autograd import needs_grad
        if needs_grad(rhs, *self._hyper_tensors()):
            out = self.operator._matmul(rhs.contiguous())
            return out / self.scale if self.inverse_scale else out * self.scale
        d = self._descriptor()
        if d is not None:
            return d.apply(rhs)
        return self.operator._matmul(rhs.contiguous()) * self._factor()

    def _size(self):
        return self.operator._size()

    def _transpose_nonbatch(self):
        return ScaleWrapperOperator(self.operator._transpose_nonbatch(), self.scale, self.inverse_scale)

    def _solve_hip(self, rhs):
        d = self._descriptor()
        if d is not None:
            from ..solvers import cg_solve
            return cg_solve(d, rhs)[0]
        return self.operator._solve(rhs) / self._factor()
