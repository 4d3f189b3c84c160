"""NoiseWrapperOperator (manifold_gp/operators/noise_wrapper_operator.py:8-28):
Q (v - s Q (v - s Q v)) = (Q - s Q^2 + s^2 Q^3) v, the 2nd-order Neumann form of (Q^-1 + s I)^-1."""
import torch

from .._compat import LinearOperator
from .._lib import host_scalar


# When the wrapped operator is one polynomial chain the solve below can also run as HIP CG on the cubic polynomial itself
# (form 1 of the descriptor: 3 nu SpMMs per iteration).  Measured on the supervised 60k epoch (tools/lab/ab_neumann.py):
# ~100 such iterations per 12-probe solve, 600 SpMM launches, against ~30 iterations of the better conditioned wrapped
# operator plus two or three series terms: 25 -> 18 ms per epoch.  False restores that CG (A/B runs).
_NEUMANN_FOR_CHAINS = [True]


class NoiseWrapperOperator(LinearOperator):
    def __init__(self, operator, noise):
        super().__init__(operator, noise=noise)
        self.operator = operator
        self.noise = noise

    def _descriptor(self):
        inner = getattr(self.operator, "_descriptor", lambda: None)()
        if inner is None or inner.form != 0:
            return None
        return inner.with_(form=1, noise=host_scalar(self, "noise"))

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
        s = host_scalar(self, "noise")
        Q = self.operator._matmul
        rhs = rhs.contiguous()
        return Q(rhs - s * Q(rhs - s * Q(rhs)))

    def _size(self):
        return self.operator._size()

    def _transpose_nonbatch(self):
        return NoiseWrapperOperator(self.operator._transpose_nonbatch(), self.noise)

    def _solve_hip(self, rhs):
        d = self._descriptor()
        if d is not None and not _NEUMANN_FOR_CHAINS[0]:
            from ..solvers import cg_solve
            return cg_solve(d, rhs)[0]
        # With q = the wrapped operator, t = s q and M = q^-1 + s I (ONE solve with the wrapped operator -- a Schur
        # complement, every matvec of which is a nested solve, answers it with a single non-nested CG on the full
        # precision; a polynomial chain with HIP CG on the chain itself):
        #     A = q - s q^2 + s^2 q^3 = q (1 + t^3) / (1 + t)     =>     A^-1 = M (1 + t^3)^-1 = M (1 - t^3 + t^6 - ...)
        # The noise model needs |t| < 1 anyway (it is the Neumann form of (q^-1 + s I)^-1); in training |t| ~ 0.4,
        # t^3 ~ 0.07.  x_J = M (b - t^3 b + ... +- t^3J b) has the TRUE residual b - A x_J = -+ t^3(J+1) b, i.e. the norm
        # of the next term, which is estimated from the ratio of the last two: the series stops as soon as that estimate
        # is under the CG tolerance (one host read per TERM -- a term is three Schur matvecs, each a nested solve -- where
        # the preconditioned CG this replaces synchronised every iteration and spent two more solves with M on the same
        # accuracy).  A ratio >= 1 (noise model outside its range) falls back to that CG.
        from .. import _lib
        from .._compat import settings
        from ..solvers import generic_cg
        s = host_scalar(self, "noise")
        inner = self.operator
        tol = float(settings.cg_tolerance.value())

        def M(v):
            with torch.no_grad():
                return inner._solve(v) + s * v

        squeeze = rhs.dim() == 1
        B = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
        with torch.no_grad():
            bn = B.norm(dim=0).clamp_min(1e-30)
            y, term, prev = B.clone(), B, 1.0
            ok = False
            for _ in range(12):
                for _k in range(3):
                    term = s * inner._matmul(term)
                term = -term
                rel = float((term.norm(dim=0) / bn).mean())
                ratio = rel / prev
                if not (ratio < 0.9):
                    break
                y = y + term
                if rel * ratio / (1.0 - ratio) < tol:
                    ok = True
                    break
                prev = rel
            if ok:
                x = M(y)
                return x.squeeze(-1) if squeeze else x
        if d is not None:
            from ..solvers import cg_solve
            return cg_solve(d, rhs)[0]
        return generic_cg(self, rhs, x0=M(rhs), precond=M)
