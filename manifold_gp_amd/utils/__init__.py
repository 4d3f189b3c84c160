"""Hot-path subset of manifold_gp.utils (manifold_gp/utils/__init__.py:3-18): NearestNeighbors,
bump_function, the training loops `manifold_informed_train` (precision form) / `vanilla_train` and the metrics
caller `test_model`.  Dataset loaders and plotting are out of scope."""
from .nearest_neighbors import NearestNeighbors
from .torch_utils import bump_function
from .test_model import test_model
from .train_model import manifold_informed_train, vanilla_train

__all__ = ["NearestNeighbors", "bump_function", "manifold_informed_train", "vanilla_train", "test_model"]
