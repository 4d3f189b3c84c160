"""Drop-in for manifold_gp.kernels (manifold_gp/kernels/__init__.py:3-5)."""
from .riemann_matern_kernel import RiemannMaternKernel

__all__ = ["RiemannMaternKernel"]
