"""RiemannGP on MI355X -- the model class of manifold_gp/models/riemann_gp.py:10-75.

Same constructor, `precision`, `modulation`, `posterior`, `posterior_mean / covar / stddev` and
`base_kernel` as the reference.  The reference inherits the posterior from gpytorch.models.ExactGP
(prediction strategy for a LowRankRootAddedDiagLinearOperator = Woodbury on the m x m root, SURVEY.md
Appendix B); gpytorch is not in this image, so the same algebra is spelled out here on device:

    K = s Z Z^T,  C = (n/s) I + Z^T Z                       (m x m, fp64, Cholesky)
    (K + n I)^-1 v = (v - Z C^-1 Z^T v) / n
    mean(x*)  = c + s Z* Z^T (K + n I)^-1 (y - c)
    cov(x*)   = s Z* [I - (s/n)(G - G C^-1 G)] Z*^T,  G = Z^T Z      (kernel block on the fp32 MFMA)

with Z = kernel.features(train_x), Z* = kernel.features(x*) (fused out-of-sample HIP kernel), s the
output scale, n the noise.  The hybrid posterior (riemann_gp.py:45-75) blends a Euclidean base model
with weight 1 - bump(distance to the nearest training point).
"""
import torch

from .. import _lib
from .._compat import HAVE_GPYTORCH, Positive
from ..operators import NoiseWrapperOperator, ScaleWrapperOperator, SchurComplementOperator
from ..utils import bump_function


class GaussianLikelihood(torch.nn.Module):
    """The one attribute of gpytorch.likelihoods.GaussianLikelihood the path reads: `noise`."""

    def __init__(self, noise=1e-2):
        super().__init__()
        self._constraint = Positive()
        self.raw_noise = torch.nn.Parameter(self._constraint.inverse_transform(torch.tensor([float(noise)])))

    @property
    def noise(self):
        return self._constraint.transform(self.raw_noise)

    @noise.setter
    def noise(self, value):
        with torch.no_grad():
            self.raw_noise.copy_(self._constraint.inverse_transform(torch.as_tensor([float(value)]).to(self.raw_noise)))


class ScaleKernel(torch.nn.Module):
    """gpytorch.kernels.ScaleKernel as far as RiemannGP uses it: `base_kernel`, `outputscale`."""

    def __init__(self, base_kernel, outputscale=1.0):
        super().__init__()
        self.base_kernel = base_kernel
        self._constraint = Positive()
        self.raw_outputscale = torch.nn.Parameter(self._constraint.inverse_transform(torch.tensor(float(outputscale))))

    @property
    def outputscale(self):
        return self._constraint.transform(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        with torch.no_grad():
            self.raw_outputscale.copy_(self._constraint.inverse_transform(torch.as_tensor(float(value)).to(self.raw_outputscale)))

    def features(self, x):
        return self.base_kernel.features(x) * self.outputscale.sqrt().to(x.device)


class _Posterior:
    """mean / covariance / stddev of a Gaussian posterior at the test inputs."""

    def __init__(self, mean, covar):
        self.mean, self.covariance_matrix = mean, covar

    @property
    def stddev(self):
        return self.covariance_matrix.diagonal().clamp_min(0).sqrt()


class RiemannGP(torch.nn.Module):
    def __init__(self, train_x, train_y, likelihood, kernel, labeled=None):
        super().__init__()
        _lib.require_device(train_x, train_y)
        self.train_inputs = (train_x,)
        self.train_targets = train_y
        self.likelihood = likelihood
        self.covar_module = kernel
        self.labeled = labeled
        self.mean_constant = torch.nn.Parameter(torch.zeros(()))     # gpytorch.means.ConstantMean
        self._cache = None

    # ------------------------------------------------------------------ riemann_gp.py:23-39
    def eval(self):
        self.base_kernel.eval()
        self._cache = None
        return super().eval()

    @property
    def base_kernel(self):
        return self.covar_module.base_kernel if hasattr(self.covar_module, "base_kernel") else self.covar_module

    def precision(self, noise=True):
        opt = self.base_kernel.precision()
        if self.labeled is not None:
            opt = SchurComplementOperator(opt, self.labeled)
        if hasattr(self.covar_module, "outputscale"):
            opt = ScaleWrapperOperator(opt, self.covar_module.outputscale)
        if noise:
            opt = NoiseWrapperOperator(opt, self.likelihood.noise)
        return opt

    # ------------------------------------------------------------------ riemann_gp.py:41-43
    def modulation(self, x):
        edge_value, _ = self.base_kernel.knn.search(x, 1)
        k = self.base_kernel
        return bump_function(edge_value.sqrt().squeeze(-1), k.bump_scale * float(k.graphbandwidth.detach().reshape(-1)[0]), k.bump_decay)

    # ------------------------------------------------------------------ the ExactGP prediction, spelled out
    def _scale_noise(self):
        s = float(self.covar_module.outputscale.detach()) if hasattr(self.covar_module, "outputscale") else 1.0
        return s, float(self.likelihood.noise.detach().reshape(-1)[0])

    def _train_cache(self):
        if self._cache is None:
            x, y = self.train_inputs[0], self.train_targets
            s, n = self._scale_noise()
            Z = self.base_kernel.features(x)
            c = float(self.mean_constant.detach())
            from ..solvers import woodbury
            wd = woodbury(Z, y - c, s, n)                              # (K + n I)^-1 (y - c) on the m x m root
            G, Lc, alpha = wd["G"], wd["Lc"], wd["alpha"]
            m = G.shape[0]
            # mean(x*) = c + Z* w,  w = s Z^T alpha = s (Z^T v - G t) / n = t   [(G + (n/s) I) t = Z^T v]
            w = wd["t"].squeeze(-1)
            # cov(x*) = s Z* M Z*^T with M = I - (s/n)(G - G C^-1 G)
            M = torch.eye(m, dtype=torch.float64, device=G.device) - (s / n) * (G - G @ torch.cholesky_solve(G, Lc))
            self._cache = dict(Z=Z, w=w.float(), M=(0.5 * (M + M.t())).float(), s=s, n=n, c=c, alpha=alpha)
        return self._cache

    def __call__(self, x):
        """Posterior of the latent function at x (eval mode), as `self(x)` does for an ExactGP."""
        from ..solvers import kernel_block
        _lib.require_device(x)
        cache = self._train_cache()
        Zs = self.base_kernel.features(x)
        mean = cache["c"] + Zs @ cache["w"]
        covar = kernel_block(Zs @ cache["M"], Zs, cache["s"])         # T x T block on the MFMA
        return _Posterior(mean, 0.5 * (covar + covar.t()))

    def _noisy(self, post, noise):
        eye = torch.eye(post.mean.shape[0], device=post.mean.device)
        return _Posterior(post.mean, post.covariance_matrix + noise * eye)

    # ------------------------------------------------------------------ riemann_gp.py:45-75
    def posterior(self, x, noisy_posterior=False, base_model=None):
        geom = self(x)
        self.posterior_geom = self._noisy(geom, self._scale_noise()[1]) if noisy_posterior else geom
        for name in ("posterior_base", "base_scale"):
            if hasattr(self, name):
                delattr(self, name)
        if base_model is not None:
            base = base_model(x)
            if noisy_posterior and hasattr(base_model, "likelihood"):
                base = self._noisy(base, float(base_model.likelihood.noise.detach().reshape(-1)[0]))
            self.posterior_base = base
            self.base_scale = 1 - self.modulation(x)
        return self

    @property
    def posterior_mean(self):
        mean = self.posterior_geom.mean.clone()
        if hasattr(self, "posterior_base"):
            mean += self.base_scale * self.posterior_base.mean
        return mean

    @property
    def posterior_covar(self):
        covar = self.posterior_geom.covariance_matrix.clone()
        if hasattr(self, "posterior_base"):
            covar += torch.outer(self.base_scale, self.base_scale) * self.posterior_base.covariance_matrix
        return covar

    @property
    def posterior_stddev(self):
        stddev = self.posterior_geom.stddev.clone()
        if hasattr(self, "posterior_base"):
            stddev += self.base_scale * self.posterior_base.stddev
        return stddev


class EuclideanGP(torch.nn.Module):
    """Exact GP with a constant mean and an RBF kernel on the ambient coordinates: the `base_model` of
    the hybrid posterior (benchmark/*.py of the reference build a gpytorch ExactGP for it).  Dense
    Cholesky through torch: sized for the labelled subset, not the hot path."""

    def __init__(self, train_x, train_y, likelihood, lengthscale=1.0, outputscale=1.0, mean=0.0):
        super().__init__()
        self.train_x, self.train_y, self.likelihood = train_x, train_y, likelihood
        self.lengthscale, self.outputscale, self.mean = float(lengthscale), float(outputscale), float(mean)
        self._chol = None

    def _k(self, a, b):
        d2 = torch.cdist(a.double(), b.double()).square()
        return self.outputscale * torch.exp(-0.5 * d2 / self.lengthscale ** 2)

    def __call__(self, x):
        n = float(self.likelihood.noise.detach().reshape(-1)[0])
        if self._chol is None:
            K = self._k(self.train_x, self.train_x)
            self._chol = torch.linalg.cholesky(K + n * torch.eye(K.shape[0], dtype=K.dtype, device=K.device))
            self._alpha = torch.cholesky_solve((self.train_y.double() - self.mean).unsqueeze(-1), self._chol).squeeze(-1)
        Ks = self._k(x, self.train_x)
        mean = self.mean + Ks @ self._alpha
        cov = self._k(x, x) - Ks @ torch.cholesky_solve(Ks.t(), self._chol)
        return _Posterior(mean.float(), cov.float())
