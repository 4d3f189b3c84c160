"""TEST DOUBLE (see tests/doubles/README.md) -- the slice of linear_operator's call contract the plugin classes use."""
from . import operators, settings  # noqa: F401
from .operators import LinearOperator  # noqa: F401

__version__ = "0.0-double"
