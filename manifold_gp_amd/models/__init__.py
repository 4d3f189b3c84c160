"""Drop-in for manifold_gp.models (manifold_gp/models/__init__.py): the RiemannGP model whose
posterior / precision entry points are the consumers of the hot path."""
from .riemann_gp import EuclideanGP, GaussianLikelihood, RiemannGP, ScaleKernel

__all__ = ["RiemannGP", "ScaleKernel", "GaussianLikelihood", "EuclideanGP"]
