"""Drop-in for manifold_gp.operators (manifold_gp/operators/__init__.py:3-15)."""
from .graph_laplacian_operator import GraphLaplacianOperator
from .precision_matern_operator import PrecisionMaternOperator
from .scale_wrapper_operator import ScaleWrapperOperator
from .noise_wrapper_operator import NoiseWrapperOperator
from .schur_complement_operator import SchurComplementOperator

__all__ = [
    "GraphLaplacianOperator",
    "PrecisionMaternOperator",
    "ScaleWrapperOperator",
    "NoiseWrapperOperator",
    "SchurComplementOperator",
]
