"""manifold_informed_train on MI355X -- manifold_gp/utils/train_model.py:49-113.

Same signature, same loss (0.5 * [y^T Q y - logdet Q + N log 2 pi] / N with Q = model.precision(), minus
the log-priors), same output-scale normalisation before / after the loop.  Every term runs on the HIP
path: `precision_operator.matmul` is the differentiable fused SpMM chain (autograd.py), `inv_quad_logdet`
is the dense Cholesky below `max_cholesky` and stochastic Lanczos + surrogate gradients above it
(solvers.inv_quad_logdet), `_average_variance` is a multi-right-hand-side HIP CG.
`vanilla_train` (gpytorch's ExactMarginalLogLikelihood on the spectral kernel) is a caller of gpytorch,
not of this path, and is not provided.
"""
import math

import torch

from .._compat import settings


def _named_priors(model):
    fn = getattr(model, "named_priors", None)
    return fn() if callable(fn) else ()


def manifold_informed_train(model, optimizer, max_iter=100, tolerance=1e-2, update_norm=None, num_rand_vec=100,
                            max_cholesky=800, cg_tolerance=1e-2, cg_max_iter=1000, scheduler=None, verbose=False):
    model.train()
    model.likelihood.train()

    def ctx():
        return (settings.max_cholesky_size(max_cholesky), settings.cg_tolerance(cg_tolerance),
                settings.max_cg_iterations(cg_max_iter))

    def average_variance():
        a, b, c = ctx()
        with torch.no_grad(), a, b, c:
            return model.covar_module.base_kernel.precision()._average_variance(num_rand_vec=num_rand_vec)

    if hasattr(model.covar_module, "outputscale"):
        model.covar_module.outputscale = model.covar_module.outputscale.detach() / average_variance()

    epoch = 0
    prev_loss = 1e6
    num_data = model.train_targets.shape[0]
    loss = torch.zeros(())
    while epoch <= max_iter:
        optimizer.zero_grad()
        precision_operator = model.precision()
        a, b, c = ctx()
        with a, b, c:
            y = model.train_targets
            loss = 0.5 * (torch.dot(y, precision_operator.matmul(y.view(-1, 1)).squeeze(-1))
                          - precision_operator.inv_quad_logdet(logdet=True)[1]
                          + num_data * math.log(2 * math.pi))
            for _, module, prior, closure, _ in _named_priors(model):
                loss = loss - prior.log_prob(closure(module)).sum()
            loss = loss / num_data

        if verbose:
            lr = scheduler.get_last_lr()[0] if scheduler is not None and hasattr(scheduler, "get_last_lr") \
                else optimizer.param_groups[0]["lr"]
            msg = ["Iteration: %d, Loss: %0.3f, Lr: %s" % (epoch, loss.item(), lr),
                   "Noise Variance: %0.3f" % model.likelihood.noise.item()]
            if hasattr(model.covar_module, "outputscale"):
                msg += ["Signal Variance: %0.3f" % model.covar_module.outputscale.item()]
            msg += ["Lengthscale: %0.3f, Graphbandwidth: %0.3f" % (model.base_kernel.lengthscale.item(),
                                                                   model.base_kernel.graphbandwidth.item())]
            print(",\t".join(msg))

        loss.backward()
        optimizer.step()
        if scheduler is not None:
            scheduler.step(loss)

        epoch += 1
        if abs(loss.item() - prev_loss) <= tolerance:
            break

        if update_norm is not None and epoch % (update_norm + 1) == 0:
            print("Update covariance normalization at epoch: ", epoch)
            model.covar_module.outputscale = 1.0 / average_variance()

    if hasattr(model.covar_module, "outputscale"):
        model.covar_module.outputscale = model.covar_module.outputscale.detach() * average_variance()
    return loss.item()
