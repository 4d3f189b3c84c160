"""test_model (manifold_gp/utils/test_model.py:10-29): RMSE and negative log likelihood of the (hybrid)
posterior at test points.  The reference reads both off `posterior_covar.inv_quad_logdet` (gpytorch, Cholesky
at these sizes); here the T x T posterior covariance (T = test points) is factored in fp64 on the device.
The settings arguments are accepted for signature parity: nothing iterative is left at this size."""
import math

import torch


def test_model(model, input, output, noisy_test=False, base_model=None, max_cholesky=800, cg_tolerance=1e-2,
               cg_iterations=1000):
    with torch.no_grad():
        model.likelihood.eval()
        model.eval()
        if base_model is not None:
            model.posterior(input, noisy_posterior=noisy_test, base_model=base_model)
        else:
            model.posterior(input, noisy_posterior=noisy_test)
        error = output - model.posterior_mean                                   # :20
        rmse = error.square().mean().sqrt()                                     # :21
        cov = model.posterior_covar.double()
        cov = 0.5 * (cov + cov.t())
        e = error.double().unsqueeze(-1)
        L, info = torch.linalg.cholesky_ex(cov)
        jitter = 0.0
        while int(info) != 0:                   # a noise-free posterior at in-sample points is singular to fp32
            jitter = 1e-8 * float(cov.diagonal().mean()) if jitter == 0.0 else jitter * 10.0
            if jitter > 1e-2 * float(cov.diagonal().mean()):
                raise RuntimeError("test_model: posterior covariance is not positive definite")
            L, info = torch.linalg.cholesky_ex(cov + jitter * torch.eye(cov.shape[0], dtype=cov.dtype, device=cov.device))
        inv_quad = torch.cholesky_solve(e, L).mul(e).sum()
        logdet = 2.0 * L.diagonal().log().sum()
        nll = 0.5 * (inv_quad + logdet + error.size(-1) * math.log(2 * math.pi)) / error.size(-1)   # :23-24
        return rmse, nll.to(rmse.dtype)


test_model.__test__ = False      # not a pytest case
