"""Plugin-surface compatibility layer.

The reference's operators subclass `linear_operator.LinearOperator` and its kernels subclass
`gpytorch.kernels.Kernel`.  When those packages are importable the classes
of this package subclass the real ones, so they drop into existing ExactGP models.  When they are
absent (this image) a small protocol base with the SAME hook names (`_matmul`, `_size`,
`_transpose_nonbatch`, `_diagonal`, `forward`, `eval`, ...) and the same public entry points
(`matmul`, `solve`, `inv_quad_logdet`, `diagonal`, `to_dense`, `diagonalization`, `.T`) is used,
so that the parity tests read like the reference's own tests either way.
"""
import contextlib
import math
import warnings

import torch

try:  # pragma: no cover - not installed in the build image
    import linear_operator as _lo
    from linear_operator.operators import LinearOperator as _RealLinearOperator
    HAVE_LINEAR_OPERATOR = True
except Exception:  # ModuleNotFoundError in this image
    _lo = None
    _RealLinearOperator = None
    HAVE_LINEAR_OPERATOR = False

try:  # pragma: no cover
    import gpytorch as _gpytorch
    HAVE_GPYTORCH = True
except Exception:
    _gpytorch = None
    HAVE_GPYTORCH = False


# ------------------------------------------------------------------------------ settings
class _Setting:
    """gpytorch.settings-style context manager holding one value."""
    _default = None
    _gp_name = None

    def __init__(self, value):
        self._new = value
        self._old = None

    @classmethod
    def value(cls):
        if HAVE_GPYTORCH and cls._gp_name is not None and "_value" not in cls.__dict__:
            return getattr(_gpytorch.settings, cls._gp_name).value()
        return cls.__dict__.get("_value", cls._default)

    def __enter__(self):
        cls = type(self)
        self._old = cls.__dict__.get("_value", None)
        cls._value = self._new
        return self

    def __exit__(self, *exc):
        cls = type(self)
        if self._old is None:
            del cls._value
        else:
            cls._value = self._old
        return False


class settings:
    """The subset of gpytorch.settings the reference touches (train_model.py:21,54,66,
    test_model.py:11, graph_laplacian_operator.py:133), with linear_operator's defaults
    (SURVEY.md Appendix B)."""

    class max_cholesky_size(_Setting):
        _default, _gp_name = 800, "max_cholesky_size"

    class cg_tolerance(_Setting):
        _default, _gp_name = 1.0, "cg_tolerance"

    class eval_cg_tolerance(_Setting):
        _default, _gp_name = 0.01, "eval_cg_tolerance"

    class max_cg_iterations(_Setting):
        _default, _gp_name = 1000, "max_cg_iterations"

    class max_root_decomposition_size(_Setting):
        _default, _gp_name = 100, "max_root_decomposition_size"

    class num_trace_samples(_Setting):
        _default, _gp_name = 10, "num_trace_samples"

    class max_lanczos_quadrature_iterations(_Setting):
        _default, _gp_name = 20, "max_lanczos_quadrature_iterations"

    class cg_jacobi_preconditioner(_Setting):
        """Extension: Jacobi-preconditioned CG in the HIP solver (reference: unpreconditioned)."""
        _default = False

    class cg_stop_mode(_Setting):
        """0 = linear_cg's stopping rule (default, reference behaviour); 1 = per-column relative
        residual <= tolerance (tight mode used for parity against converged oracles)."""
        _default = 0


# ------------------------------------------------------------------------------ LinearOperator
class _ProtocolLinearOperator:
    """Minimal stand-in for linear_operator.LinearOperator (same hooks, same entry points)."""

    def __init__(self, *args, **kwargs):
        self._args = args
        self._kwargs = kwargs

    # hooks the subclasses implement
    def _matmul(self, rhs):
        raise NotImplementedError

    def _size(self):
        raise NotImplementedError

    def _transpose_nonbatch(self):
        raise NotImplementedError

    def _diagonal(self):
        raise NotImplementedError

    # public surface
    @property
    def shape(self):
        return self._size()

    def size(self, dim=None):
        s = self._size()
        return s if dim is None else s[dim]

    def dim(self):
        return 2

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        for a in list(self._args) + list(self._kwargs.values()):
            if torch.is_tensor(a):
                return a.device
            if isinstance(a, _ProtocolLinearOperator):
                return a.device
        return torch.device("cpu")

    def representation(self):
        return tuple(a for a in self._args if torch.is_tensor(a))

    def matmul(self, other):
        if other.dim() == 1:
            return self._matmul(other.unsqueeze(-1)).squeeze(-1)
        return self._matmul(other)

    __matmul__ = matmul

    @property
    def T(self):
        return self._transpose_nonbatch()

    mT = T

    def transpose(self, d0, d1):
        return self._transpose_nonbatch()

    def t(self):
        return self._transpose_nonbatch()

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self._diagonal()

    def to_dense(self):
        n = self._size()[-1]
        out = []
        step = 256
        for c0 in range(0, n, step):
            c1 = min(n, c0 + step)
            eye = torch.zeros(n, c1 - c0, device=self.device, dtype=torch.float32)
            eye[torch.arange(c0, c1, device=self.device), torch.arange(c1 - c0, device=self.device)] = 1.0
            out.append(self._matmul(eye))
        return torch.cat(out, dim=1)

    def evaluate(self):
        return self.to_dense()

    def diagonalization(self, method=None):
        from .solvers import dense_symeig
        return dense_symeig(self)

    def __add__(self, other):
        raise NotImplementedError("lazy sums are provided by linear_operator; not needed on the hot path")


class _HipEntryPoints:
    """The solver entry points of the operators, implemented on the HIP solvers (solvers.py).

    They are defined HERE, ahead of the base class in the MRO, for both bases.  With the real
    linear_operator.LinearOperator underneath this keeps `solve` / `inv_quad_logdet` on the HIP CG, the block
    Lanczos log-determinant and the surrogate gradients of solvers.inv_quad_logdet instead of linear_operator's
    own linear_cg / `_bilinear_derivative` machinery, whose calling convention of `_solve` (a (solves, tridiagonals)
    pair when num_tridiag > 0) the operators honour as well (solvers.solve_with_tridiag) but which has never run
    against these classes: the image has no linear_operator (INTEGRATION.md section 2 says what is verified)."""

    def solve(self, right_tensor, left_tensor=None):
        squeeze = right_tensor.dim() == 1
        rhs = right_tensor.unsqueeze(-1) if squeeze else right_tensor
        sol = self._solve(rhs)
        if left_tensor is not None:
            sol = left_tensor @ sol
        return sol.squeeze(-1) if squeeze else sol

    def _solve(self, rhs, preconditioner=None, num_tridiag=0):
        # `_solve_hip(rhs)`: the operator's own HIP solve (one CG on its polynomial chain, block elimination for the
        # Schur complement, ...); operators without one take the generic CG over `_matmul`
        from .solvers import generic_cg, solve_with_tridiag
        hip = getattr(self, "_solve_hip", None)
        sol = hip(rhs) if hip is not None else generic_cg(self, rhs)
        return solve_with_tridiag(self, sol, rhs, num_tridiag)

    def inv_quad_logdet(self, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
        from .solvers import inv_quad_logdet
        return inv_quad_logdet(self, inv_quad_rhs, logdet, reduce_inv_quad)

    def inv_quad(self, inv_quad_rhs, reduce_inv_quad=True):
        return self.inv_quad_logdet(inv_quad_rhs, False, reduce_inv_quad)[0]

    def logdet(self):
        return self.inv_quad_logdet(None, True)[1]


if HAVE_LINEAR_OPERATOR:  # pragma: no cover
    class LinearOperator(_HipEntryPoints, _RealLinearOperator):
        pass
else:
    class LinearOperator(_HipEntryPoints, _ProtocolLinearOperator):
        pass


# ------------------------------------------------------------------------------ Kernel
class _Positive:
    """gpytorch.constraints.Positive: softplus transform."""

    def transform(self, raw):
        return torch.nn.functional.softplus(raw)

    def inverse_transform(self, value):
        # inverse softplus, stable for large values
        return value + torch.log(-torch.expm1(-value))


class _ProtocolKernel(torch.nn.Module):
    """Stand-in for gpytorch.kernels.Kernel: lengthscale parameter with a Positive constraint,
    `initialize(**hypers)`, `register_constraint/prior`, `__call__` -> forward."""
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None,
                 lengthscale_prior=None, lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self._constraints = {}
        self._priors = {}
        if self.has_lengthscale:
            self.register_parameter("raw_lengthscale", torch.nn.Parameter(torch.zeros(*batch_shape, 1, 1)))
            self.register_constraint("raw_lengthscale", lengthscale_constraint or _Positive())

    def register_parameter(self, name, parameter=None, param=None):
        # gpytorch.Module names the argument `parameter`, torch.nn.Module names it `param`
        return super().register_parameter(name, parameter if parameter is not None else param)

    def register_constraint(self, name, constraint):
        self._constraints[name] = constraint
        setattr(self, name + "_constraint_obj", constraint)

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        self._priors[name] = (prior, param_or_closure, setting_closure)

    def __getattr__(self, name):
        if name.endswith("_constraint") and name.startswith("raw_"):
            cons = self.__dict__.get("_constraints", {})
            if name[: -len("_constraint")] in cons:
                return cons[name[: -len("_constraint")]]
        return super().__getattr__(name)

    @property
    def lengthscale(self):
        return self._constraints["raw_lengthscale"].transform(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        self.initialize(lengthscale=value)

    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            if not torch.is_tensor(val):
                val = torch.as_tensor(val, dtype=torch.float32)
            if name.startswith("raw_"):
                p = getattr(self, name)
                p.data.copy_(val.to(p).expand_as(p))
            else:
                raw = getattr(self, "raw_" + name)
                cons = self._constraints["raw_" + name]
                raw.data.copy_(cons.inverse_transform(val.to(raw)).expand_as(raw))
        return self

    def __call__(self, x1, x2=None, diag=False, **params):
        if x2 is None:
            x2 = x1
        return self.forward(x1, x2, diag=diag, **params)


if HAVE_GPYTORCH:  # pragma: no cover
    Kernel = _gpytorch.kernels.Kernel
    Positive = _gpytorch.constraints.Positive
else:
    Kernel = _ProtocolKernel
    Positive = _Positive


def warn_once(msg, _seen=set()):
    if msg not in _seen:
        _seen.add(msg)
        warnings.warn(msg)
