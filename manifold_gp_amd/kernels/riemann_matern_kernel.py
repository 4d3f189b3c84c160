"""RiemannMaternKernel (manifold_gp/kernels/riemann_matern_kernel.py:10-25): Matern spectral density
on the graph spectrum and the factory of the Matern precision operator."""
from .riemann_kernel import RiemannKernel
from ..operators import PrecisionMaternOperator


class RiemannMaternKernel(RiemannKernel):
    has_lengthscale = True

    def __init__(self, nu=2, **kwargs):
        super().__init__(**kwargs)
        self.nu = nu

    def spectral_density(self):
        ls = self.lengthscale.to(self.eigval.device)
        return (2 * self.nu / ls.square() + self.eigval).pow(-self.nu)

    def precision(self):
        return PrecisionMaternOperator(self.laplacian(), self.nu, self.lengthscale)
