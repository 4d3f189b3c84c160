"""Hyper-parameter gradient path.

The reference differentiates its loss by autograd through the torch ops of
`GraphLaplacianOperator._matmul` (manifold_gp/operators/graph_laplacian_operator.py:108-124; default
`_bilinear_derivative` of linear_operator), pinned by test/_test_functions.py:59-104 (`test_grad`,
`test_ml`).  Here the differentiable primitive is the fused HIP SpMM itself:

    Y = cb * base + co * post (.) (a * xs + b * (diag (.) xs - S xs)),   xs = pre (.) X

with gradients for X, the scalars a, b, co, cb, the row vectors pre / post / base and the graph
bandwidth eps.  The eps-gradient needs no SDDMM: the Laplacian's forward-mode tangent
(mgp_laplacian_tangent -> d_vals, d_diag, three row passes) turns it into one more fused SpMV,
d(h^T L xs)/d eps = h^T L' xs.  Node vectors that depend on eps (sqrt(D), 1/sqrt(D), D) are exposed
through `node_vector`, whose backward is a dot with their tangents.  The precision / wrapper operators
compose these primitives with ordinary torch scalar arithmetic when gradients are requested.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


FUSED_BACKWARD_SUMS = [True]     # False: the torch form of the backward pass's reductions (A/B runs, tests)


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


def _spmm(data, X, a, b, pre, post, base, cb, co, tangent=False):
    """Plain (non-differentiable) launch of the fused SpMM on the Laplacian or its eps-tangent."""
    g = data.graph
    check(lib().mgp_spmm_set_group_hint(g.spmv_lanes), "mgp_spmm_set_group_hint")
    if tangent:
        t = data.tangent()
        csr = g.csr_with(t.d_vals, t.d_diag, t.d_vals_t)
    else:
        csr = data.csr(wide=X.shape[1] >= 48)
    X = _lib.f32c(X)
    out = torch.empty_like(X)
    C = X.shape[1]
    for c0 in range(0, C, 256):
        sl = slice(c0, min(C, c0 + 256))
        Xc = X if C <= 256 else X[:, sl].contiguous()
        Bc = None if base is None else (base if C <= 256 else base[:, sl].contiguous())
        Yc = out if C <= 256 else torch.empty_like(Xc)
        check(lib().mgp_spmm_fused(ctypes.byref(csr), ptr(Xc), Xc.shape[1], ptr(Yc), float(a), float(b), ptr(pre),
                                   ptr(post), ptr(Bc), float(cb), float(co), None, None, stream()), "mgp_spmm_fused")
        if Yc is not out:
            out[:, sl] = Yc
    return out


class _NodeVector(torch.autograd.Function):
    """eps -> one of the node vectors of LaplacianData; backward = <g, d vector / d eps>."""

    @staticmethod
    def forward(ctx, eps, data, name):
        ctx.data, ctx.name = data, name
        ctx.eps_shape = eps.shape
        return getattr(data, name).detach()

    @staticmethod
    def backward(ctx, g):
        t = ctx.data.tangent()
        d = getattr(t, "d_" + ctx.name)
        return (g * d).sum().reshape(ctx.eps_shape), None, None


def node_vector(eps, data, name):
    """name in {"dsqrt", "dinvsqrt", "degree", "degree_unnorm", "diag"}."""
    return _NodeVector.apply(eps, data, name)


_CONST0 = {}


def _as0(t, like=None):
    # a python scalar becomes a CPU tensor: no host-to-device copy, and reading it back is free (the kernels take their
    # coefficients by value; a device tensor here would be copied up only to be synchronised down again).  The handful of
    # constants the operators pass (0, 1) are made once: a tensor construction per coefficient per launch was 36 of them per epoch
    if torch.is_tensor(t):
        return t
    v = float(t)
    c = _CONST0.get(v)
    if c is None:
        if len(_CONST0) > 64:
            _CONST0.clear()
        c = _CONST0[v] = torch.tensor(v)
    return c


def _host_values(*ts):
    """Python floats of 0-d tensors with at most ONE device synchronisation (for those that live on the GPU)."""
    dev = [t for t in ts if t.device.type != "cpu"]
    if len(dev) > 1:
        vals = iter(torch.stack([t.detach().reshape(()).float() for t in dev]).tolist())
    else:
        vals = iter([t.item() for t in dev])
    return [float(next(vals)) if t.device.type != "cpu" else float(t) for t in ts]


class _FusedSpmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, eps, a, b, co, cb, pre, post, base, data, known=None):
        ctx.data = data
        ctx.save_for_backward(X, eps, a, b, co, cb,
                              pre if pre is not None else torch.empty(0), post if post is not None else torch.empty(0),
                              base if base is not None else torch.empty(0))
        ctx.has = (pre is not None, post is not None, base is not None)
        # (known: host values the caller already holds for some of a, b, co, cb -- a device scalar derived from a hyper-parameter
        # whose host copy exists, _lib.host_scalar -- so that the launch needs no read-back of its own)
        kn = known or (None, None, None, None)
        need = [t for t, k in zip((a, b, co, cb), kn) if k is None]
        got = iter(_host_values(*need)) if need else iter(())
        av, bv, cov, cbv = ctx.vals = tuple(float(k) if k is not None else next(got) for k in kn)   # read once, reused by backward
        return _spmm(data, X, av, bv, pre, post, base, cbv, cov)

    @staticmethod
    def backward(ctx, g):
        X, eps, a, b, co, cb, pre, post, base = ctx.saved_tensors
        has_pre, has_post, has_base = ctx.has
        data = ctx.data
        pre = pre if has_pre else None
        post = post if has_post else None
        base = base if has_base else None
        g = _lib.f32c(g)
        av, bv, cov, cbv = ctx.vals
        need = ctx.needs_input_grad
        xs = X * pre.view(-1, 1) if pre is not None else X
        h = g * cov
        if post is not None:
            h = h * post.view(-1, 1)
        gX = geps = ga = gb = gco = gcb = gpre = gpost = gbase = None
        lx = gxs = dlx = None
        want_pre = need[6] and pre is not None
        want_post = need[7] and post is not None
        if need[3] or need[4] or want_post:
            lx = _spmm(data, xs, 0.0, 1.0, None, None, None, 0.0, 1.0)          # L xs (recomputed)
        if need[0] or want_pre:
            gxs = _spmm(data, h, av, bv, None, None, None, 0.0, 1.0)            # a h + b L h (L symmetric)
            if need[0]:
                gX = gxs * pre.view(-1, 1) if pre is not None else gxs
        if need[1]:
            dlx = _spmm(data, xs, 0.0, 1.0, None, None, None, 0.0, 1.0, tangent=True)   # L' xs
        if FUSED_BACKWARD_SUMS[0] and (need[1] or need[2] or want_pre or want_post):
            # <h, L' xs>, <h, xs>, the row sums of gxs (.) X and of g (.) (a xs + b L xs) in ONE launch (mgp_spmm_backward_sums)
            n, C = g.shape
            nb = lib().mgp_spmm_backward_blocks(n)
            part = torch.empty(nb, 2, dtype=torch.float32, device=g.device)
            gpre = torch.empty(n, dtype=torch.float32, device=g.device) if want_pre else None
            gpost = torch.empty(n, dtype=torch.float32, device=g.device) if want_post else None
            Xc = _lib.f32c(X)
            check(lib().mgp_spmm_backward_sums(n, C, ptr(h), ptr(dlx), ptr(xs if (need[2] or want_post) else None), ptr(gxs if want_pre else None),
                                               ptr(Xc if want_pre else None), ptr(g if want_post else None), ptr(lx if want_post else None),
                                               float(av), float(bv), float(cov), ptr(part), ptr(gpre), ptr(gpost), stream()),
                  "mgp_spmm_backward_sums")
            tot = part.sum(0)
            if need[1]:
                geps = (bv * tot[0]).reshape(eps.shape)
            if need[2]:
                ga = tot[1].reshape(a.shape).to(a.device)
        else:
            if want_pre:
                gpre = (gxs * X).sum(1)
            if need[1]:
                geps = (bv * (h * dlx).sum()).reshape(eps.shape)
            if need[2]:
                ga = (h * xs).sum().reshape(a.shape).to(a.device)
            if want_post:
                gpost = (g * (av * xs + bv * lx)).sum(1) * cov
        if need[3]:
            gb = (h * lx).sum().reshape(b.shape).to(b.device)
        if need[4]:
            t = av * xs + bv * lx
            gco = ((g * t * post.view(-1, 1)).sum() if post is not None else (g * t).sum()).reshape(co.shape).to(co.device)
        if need[5]:
            gcb = ((g * base).sum() if base is not None else torch.zeros((), device=g.device)).reshape(cb.shape).to(cb.device)
        if need[8] and base is not None:
            gbase = g * cbv
        return gX, geps, ga, gb, gco, gcb, gpre, gpost, gbase, None, None


def fused_spmm(data, X, eps, a=0.0, b=1.0, co=1.0, cb=0.0, pre=None, post=None, base=None, a_value=None):
    """Differentiable fused SpMM on the Laplacian held by `data` (built at the current eps).  a_value: the host value of `a` when
    the caller has it (no read-back)."""
    squeeze = X.dim() == 1
    Xc = _lib.f32c(X.unsqueeze(-1) if squeeze else X)
    if base is not None and base.dim() == 1:
        base = base.unsqueeze(-1)
    eps = eps if torch.is_tensor(eps) else torch.tensor(float(eps), device=Xc.device)
    known = None if a_value is None else (float(a_value), None, None, None)
    out = _FusedSpmm.apply(Xc, eps, _as0(a, Xc), _as0(b, Xc), _as0(co, Xc), _as0(cb, Xc), pre, post, base, data, known)
    return out.squeeze(-1) if squeeze else out
