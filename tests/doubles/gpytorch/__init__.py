"""TEST DOUBLE (see tests/doubles/README.md) -- gpytorch.kernels.Kernel's registration contract, constraints, priors,
settings.  riemann_kernel.py:28-63 of the reference is what it has to carry."""
import types

import torch

from linear_operator import settings as _lo_settings

__version__ = "0.0-double"


class Module(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self._priors = {}
        self._constraints_names = {}

    def register_parameter(self, name, parameter):
        super().register_parameter(name, parameter)

    def register_constraint(self, param_name, constraint):
        if param_name not in self._parameters:
            raise RuntimeError("Attempting to register constraint for nonexistent parameter.")
        self.add_module(param_name + "_constraint", constraint)

    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if not callable(param_or_closure) and param_or_closure not in self._parameters:
            raise AttributeError("Unknown parameter %s" % param_or_closure)
        self.add_module(name, prior)
        self._priors[name] = (prior, param_or_closure, setting_closure)

    def initialize(self, **kwargs):
        for name, val in kwargs.items():
            if isinstance(val, int):
                val = float(val)
            if not hasattr(self, name):
                raise AttributeError("Unknown parameter {p} for {c}".format(p=name, c=self.__class__.__name__))
            elif name not in self._parameters and name not in self._buffers:
                setattr(self, name, val)
            elif torch.is_tensor(val):
                p = self.__getattr__(name)
                p.data.copy_(val.to(p).expand_as(p))
            else:
                self.__getattr__(name).data.fill_(val)
        return self


class _Positive(torch.nn.Module):
    def transform(self, raw):
        return torch.nn.functional.softplus(raw)

    def inverse_transform(self, value):
        return value + torch.log(-torch.expm1(-value))


class _Prior(Module):
    def log_prob(self, x):
        raise NotImplementedError


class _NormalPrior(_Prior):
    def __init__(self, loc, scale):
        super().__init__()
        self.loc, self.scale = float(loc), float(scale)

    def log_prob(self, x):
        return torch.distributions.Normal(self.loc, self.scale).log_prob(x)


class _Kernel(Module):
    has_lengthscale = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self._batch_shape = batch_shape
        self.ard_num_dims = ard_num_dims
        self.eps = eps
        if self.has_lengthscale:
            dims = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter(name="raw_lengthscale",
                                    parameter=torch.nn.Parameter(torch.zeros(*self.batch_shape, 1, dims)))
            if lengthscale_prior is not None:
                self.register_prior("lengthscale_prior", lengthscale_prior, lambda m: m.lengthscale,
                                    lambda m, v: m._set_lengthscale(v))
            self.register_constraint("raw_lengthscale", lengthscale_constraint if lengthscale_constraint is not None else _Positive())

    @property
    def batch_shape(self):
        return self._batch_shape

    @property
    def lengthscale(self):
        return self.raw_lengthscale_constraint.transform(self.raw_lengthscale) if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_lengthscale(value)

    def _set_lengthscale(self, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value).to(self.raw_lengthscale)
        self.initialize(raw_lengthscale=self.raw_lengthscale_constraint.inverse_transform(value))

    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, **params):
        raise NotImplementedError

    def __call__(self, x1, x2=None, diag=False, last_dim_is_batch=False, **params):
        # the real class wraps the call in a LazyEvaluatedKernelTensor whose evaluation calls forward(x1, x2, ...)
        return self.forward(x1, x1 if x2 is None else x2, diag=diag, last_dim_is_batch=last_dim_is_batch, **params)


kernels = types.SimpleNamespace(Kernel=_Kernel)
constraints = types.SimpleNamespace(Positive=_Positive)
priors = types.ModuleType("gpytorch.priors")
priors.Prior = _Prior
priors.NormalPrior = _NormalPrior
settings = _lo_settings

import sys as _sys  # noqa: E402
_sys.modules[__name__ + ".priors"] = priors
