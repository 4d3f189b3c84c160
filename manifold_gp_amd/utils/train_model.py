"""manifold_informed_train on MI355X -- manifold_gp/utils/train_model.py:49-113.

Same signature, same loss (0.5 * [y^T Q y - logdet Q + N log 2 pi] / N with Q = model.precision(), minus
the log-priors), same output-scale normalisation before / after the loop.  Every term runs on the HIP
path: `precision_operator.matmul` is the differentiable fused SpMM chain (autograd.py), `inv_quad_logdet`
is the dense Cholesky below `max_cholesky` and stochastic Lanczos + surrogate gradients above it
(solvers.inv_quad_logdet), `_average_variance` is a multi-right-hand-side HIP CG.
`vanilla_train` (train_model.py:10-46) maximises the exact marginal likelihood of the spectral kernel
K = s Z Z^T + noise I, Z = sqrt(N S(lambda; kappa) / sum S) (.) Phi.  The reference gets it from gpytorch's
ExactMarginalLogLikelihood (Woodbury on the m x m root); without gpytorch the same closed form is written
out here: the N-sized quantities Phi^T Phi, Phi^T y, Phi^T 1 are formed once on the device, every epoch is
m x m fp64 algebra under autograd (noise, output scale, length scale, constant mean; the eigenpairs -- and
with them the graph bandwidth -- are held fixed exactly as in the reference, which computes them under
no_grad in eval()).
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

    # every epoch runs tens of multi-column solves on one graph: have its nearest-neighbour chain order built now (once, a
    # host walk: 24 ms at 60k; None where it does not pay) -- solves of 8 columns and more then iterate on the matrix in that
    # order (solvers.CHAIN_SOLVE_MIN_C), which a run with fewer than 48 normalisation columns would otherwise never trigger
    graph = getattr(getattr(model.base_kernel, "knn", None), "knn_graph", None)
    if graph is not None and hasattr(graph, "wide_relabelled"):
        graph.wide_relabelled()

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


def exact_mll_lowrank(model):
    """-(1/N) log p(y) for K = s Z Z^T + noise I (what gpytorch's ExactMarginalLogLikelihood returns, negated),
    differentiable wrt noise / output scale / length scale / constant mean."""
    kern = model.base_kernel
    if getattr(kern, "eigvec", None) is None:
        training = model.training
        kern.eval()                      # eigenpairs of the current graph bandwidth (no_grad, as the reference)
        model.train(training)
    cache = getattr(model, "_vanilla_cache", None)
    x, y = model.train_inputs[0], model.train_targets
    insample = kern._is_train_inputs(x)

    def scale():
        # Z = A * sqrt(N S / sum S): S the spectral density, divided by (1 - eps^2 lambda)^2 for inputs that are
        # not the graph's own points (riemann_kernel.py:134-136 / :143-145)
        S = kern.spectral_density().double().reshape(-1)
        if not insample:
            eps = kern.graphbandwidth.detach().double().reshape(-1)[0].to(S.device)
            S = S / (1.0 - eps * eps * kern.eigval.double()).square()
        return (float(kern.eigvec.shape[0]) * S / S.sum()).sqrt()

    if cache is None or cache["eigvec"] is not kern.eigvec or cache["x"] is not x:
        with torch.no_grad():
            A = kern.eigvec if insample else (kern.features(x).double() / scale()).float()
            yd = y.double()
            m = A.shape[1]
            if m + 2 <= 512:
                # one fp64-accumulating HIP Gram pass over [A | y | 1]: A^T A, A^T y, A^T 1 (mgp_gram_f64)
                from ..solvers import gram_f64
                Gb = gram_f64(torch.cat([A.float(), y.float().view(-1, 1), torch.ones_like(y).float().view(-1, 1)], 1))
                G0, by, b1 = 0.5 * (Gb[:m, :m] + Gb[:m, :m].t()), Gb[:m, m].clone(), Gb[:m, m + 1].clone()
            else:
                Ad = A.double()
                G0, by, b1 = Ad.t() @ Ad, Ad.t() @ yd, Ad.t() @ torch.ones_like(yd)
            cache = dict(eigvec=kern.eigvec, x=x, G0=G0, by=by, b1=b1, yy=torch.dot(yd, yd), y1=yd.sum(), n=float(yd.shape[0]))
        model._vanilla_cache = cache
    N = cache["n"]
    d = scale()
    s = model.covar_module.outputscale.double().reshape(()) if hasattr(model.covar_module, "outputscale") \
        else torch.ones((), dtype=torch.float64, device=d.device)
    noise = model.likelihood.noise.double().reshape(())
    c = model.mean_constant.double() if hasattr(model, "mean_constant") else torch.zeros((), dtype=torch.float64, device=d.device)
    G = d.view(-1, 1) * cache["G0"] * d.view(1, -1)
    m = G.shape[0]
    C = G + (noise / s) * torch.eye(m, dtype=torch.float64, device=G.device)
    Lc = torch.linalg.cholesky(C)
    zty = d * (cache["by"] - c * cache["b1"])
    rr = cache["yy"] - 2.0 * c * cache["y1"] + c * c * N
    t = torch.cholesky_solve(zty.view(-1, 1), Lc).view(-1)
    quad = (rr - torch.dot(zty, t)) / noise
    logdet = (N - m) * torch.log(noise) + m * torch.log(s) + 2.0 * torch.log(torch.diagonal(Lc)).sum()
    return (0.5 * (quad + logdet + N * math.log(2 * math.pi)) / N).float()


def vanilla_train(model, optimizer, max_iter=100, max_cholesky=800, tolerance=1e-2, cg_tolerance=1e-2, cg_max_iter=1000,
                  scheduler=None, verbose=False):
    model.train()
    model.likelihood.train()
    epoch = 0
    prev_loss = 1e6
    loss = torch.zeros(())
    while epoch <= max_iter:
        optimizer.zero_grad()
        loss = exact_mll_lowrank(model)
        if verbose:
            lr = scheduler.get_last_lr()[0] if scheduler is not None and hasattr(scheduler, "get_last_lr") \
                else optimizer.param_groups[0]["lr"]
            msg = ["Iteration: %d, Loss: %0.3f, Lr: %s" % (epoch, loss.item(), lr),
                   "Noise Variance: %0.3f" % model.likelihood.noise.item()]
            if hasattr(model.covar_module, "outputscale"):
                msg += ["Signal Variance: %0.3f" % model.covar_module.outputscale.item()]
            msg += ["Lengthscale: %0.3f" % model.base_kernel.lengthscale.item()]
            print(",\t".join(msg))
        loss.backward()
        optimizer.step()
        if scheduler is not None:
            scheduler.step(loss)
        epoch += 1
        if abs(loss.item() - prev_loss) <= tolerance:       # prev_loss is never updated in the reference either (:16, :40)
            break
    return loss.item()
