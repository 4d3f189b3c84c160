"""manifold_gp_amd -- MI355X-native implementation of manifold-gp's sparse graph-Laplacian GP
inference path behind the reference's plugin surface:

    manifold_gp_amd.kernels    <->  manifold_gp.kernels    (RiemannMaternKernel)
    manifold_gp_amd.operators  <->  manifold_gp.operators  (GraphLaplacianOperator, ...)
    manifold_gp_amd.utils      <->  manifold_gp.utils      (NearestNeighbors, bump_function)
    manifold_gp_amd.models     <->  manifold_gp.models     (RiemannGP: precision / hybrid posterior)

All arithmetic runs in hand-written gfx950 HIP kernels (manifold_gp_amd/csrc) loaded through the
C-ABI of include/mgp_hip.h; there is no CPU fallback.  `install_as_manifold_gp()` aliases the three
sub-packages into `sys.modules` under the reference's names so that existing model code
(`from manifold_gp.kernels import RiemannMaternKernel`) picks them up unchanged.
"""
import sys

from . import _compat
from ._compat import settings
from . import kernels, operators, utils, models  # noqa: F401

__version__ = "0.1.0"


def install_as_manifold_gp(force=False):
    """Register this package's sub-packages under the reference's module names."""
    import types
    if "manifold_gp" in sys.modules and not force:
        base = sys.modules["manifold_gp"]
    else:
        base = types.ModuleType("manifold_gp")
        base.__path__ = []
        sys.modules["manifold_gp"] = base
    for name, mod in (("kernels", kernels), ("operators", operators), ("utils", utils), ("models", models)):
        sys.modules["manifold_gp." + name] = mod
        setattr(base, name, mod)
    return base
