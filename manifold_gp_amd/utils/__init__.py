"""Hot-path subset of manifold_gp.utils (manifold_gp/utils/__init__.py:3-18): NearestNeighbors and
bump_function.  Dataset loaders, plotting and the training harness are out of scope (SURVEY.md
section 2)."""
from .nearest_neighbors import NearestNeighbors
from .torch_utils import bump_function

__all__ = ["NearestNeighbors", "bump_function"]
