"""TEST DOUBLE -- settings as context managers with `.value()` (linear_operator.settings / gpytorch.settings)."""


class _Value:
    _global_value = None

    def __init__(self, value):
        self._new, self._old = value, None

    @classmethod
    def value(cls):
        return cls._global_value

    def __enter__(self):
        cls = type(self)
        self._old, cls._global_value = cls._global_value, self._new
        return self

    def __exit__(self, *exc):
        type(self)._global_value = self._old
        return False


class max_cholesky_size(_Value):
    _global_value = 800


class cg_tolerance(_Value):
    _global_value = 1.0


class eval_cg_tolerance(_Value):
    _global_value = 0.01


class max_cg_iterations(_Value):
    _global_value = 1000


class max_root_decomposition_size(_Value):
    _global_value = 100


class num_trace_samples(_Value):
    _global_value = 10


class max_lanczos_quadrature_iterations(_Value):
    _global_value = 20
