"""Stochastic Lanczos quadrature log-determinant.

What `inv_quad_logdet(logdet=True)` does in linear_operator for N > max_cholesky_size
(manifold_gp/utils/train_model.py:68): with Rademacher probes z_p (||z||^2 = N) and the k-step Lanczos
tridiagonal T_p = Q_p^T A Q_p started at z_p / ||z_p||,

    logdet(A) ~= (N / P) * sum_p  e_1^T log(T_p) e_1 = (N / P) * sum_p sum_i tau_{p,i}^2 log(theta_{p,i}).

The Lanczos runs on device (mgp_lanczos_tridiag: operator chain = fused SpMM launches, full
re-orthogonalisation); the k x k tridiagonal eigenproblems are solved on the host in fp64.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._compat import settings
from ._lib import check, lib, ptr, stream


def lanczos_tridiag(desc, q0, steps):
    """(alpha[steps], beta[steps]) of the Lanczos tridiagonal of the operator `desc` started at q0."""
    _lib.require_device(q0)
    op = desc.struct()
    wb = lib().mgp_lanczos_tridiag_workspace_bytes(ctypes.byref(op), int(steps))
    work = _lib.workspace(wb, "lanczos_tridiag", q0.device)
    alpha = (ctypes.c_float * steps)()
    beta = (ctypes.c_float * steps)()
    check(lib().mgp_lanczos_tridiag(ctypes.byref(op), ptr(_lib.f32c(q0)), int(steps), alpha, beta, None, ptr(work),
                                    work.numel(), stream()), "mgp_lanczos_tridiag")
    return np.array(alpha, dtype=np.float64), np.array(beta, dtype=np.float64)


def lanczos_tridiag_block(desc, Q0, steps):
    """P independent Lanczos runs at once (mgp_lanczos_tridiag_block): Q0 [n, P], P <= 16.
    Returns alpha [steps, P], beta [steps, P] (float64 numpy)."""
    _lib.require_device(Q0)
    Q0 = _lib.f32c(Q0)
    P = Q0.shape[1]
    op = desc.struct()
    wb = lib().mgp_lanczos_tridiag_block_workspace_bytes(ctypes.byref(op), P, int(steps))
    if wb == 0:
        raise RuntimeError("mgp_lanczos_tridiag_block: unsupported shape (P = %d, steps = %d)" % (P, steps))
    work = _lib.workspace(wb, "lanczos_tridiag", Q0.device)
    alpha = (ctypes.c_float * (steps * P))()
    beta = (ctypes.c_float * (steps * P))()
    check(lib().mgp_lanczos_tridiag_block(ctypes.byref(op), ptr(Q0), P, int(steps), alpha, beta, ptr(work), work.numel(),
                                          stream()), "mgp_lanczos_tridiag_block")
    return (np.array(alpha, dtype=np.float64).reshape(steps, P), np.array(beta, dtype=np.float64).reshape(steps, P))


def _generic_device(operator):
    op = operator
    while op is not None:
        for name in ("mask", "scale", "noise"):
            t = getattr(op, name, None)
            if torch.is_tensor(t) and t.is_cuda:
                return t.device
        base = getattr(op, "base", None)
        if base is not None and hasattr(base, "laplacian"):
            return base.laplacian.x.device
        op = getattr(op, "operator", None) or base
    return torch.device("cuda:0")


_PROBES = {}


def rademacher_probes(n, num_probes, seed, device):
    """[n, num_probes] matrix of +-1 from a seeded CPU generator, kept on the device: the probes of one (n, count,
    seed) are the same at every call by construction, and generating + uploading 60k x 12 of them took longer
    (~15 ms) than the 20-step block Lanczos that consumes them (9 ms)."""
    key = (int(n), int(num_probes), int(seed), str(device))
    Z = _PROBES.get(key)
    if Z is None:
        if len(_PROBES) >= 8:
            _PROBES.pop(next(iter(_PROBES)))
        gen = torch.Generator(device="cpu").manual_seed(seed)
        Z = (torch.randint(0, 2, (n, num_probes), generator=gen).float() * 2 - 1).to(device)
        _PROBES[key] = Z
    return Z


def _quadrature_log(alpha, beta, fun=None):
    k = len(alpha)
    # an (almost) zero beta means the Krylov space is exhausted: truncate there
    cut = k
    for j in range(k - 1):
        if not np.isfinite(beta[j]) or abs(beta[j]) < 1e-6 * max(abs(alpha[j]), 1e-30):
            cut = j + 1
            break
    T = np.diag(alpha[:cut]) + np.diag(beta[:cut - 1], 1) + np.diag(beta[:cut - 1], -1)
    theta, S = np.linalg.eigh(T)
    theta = fun(theta) if fun is not None else theta
    theta = np.maximum(theta, 1e-30)
    return float(np.sum(S[0, :] ** 2 * np.log(theta)))


def _quadrature_log_sum(a, b, fun=None):
    """sum over the probes (columns of a / b [steps, P]) of e_1^T log(fun(T_p)) e_1 (fun = identity by default): one
    batched eigh for the probes whose Krylov space is not exhausted, the scalar routine for the (rare) others."""
    k, P = a.shape
    ok = np.isfinite(b[:k - 1]).all(0) & (np.abs(b[:k - 1]) >= 1e-6 * np.maximum(np.abs(a[:k - 1]), 1e-30)).all(0) if k > 1 \
        else np.ones(P, bool)
    total = 0.0
    idx = np.nonzero(ok)[0]
    if idx.size:
        T = np.zeros((idx.size, k, k))
        r = np.arange(k)
        T[:, r, r] = a[:, idx].T
        if k > 1:
            T[:, r[:-1], r[1:]] = b[:k - 1, idx].T
            T[:, r[1:], r[:-1]] = b[:k - 1, idx].T
        theta, S = np.linalg.eigh(T)
        theta = fun(theta) if fun is not None else theta
        total += float(np.sum(S[:, 0, :] ** 2 * np.log(np.maximum(theta, 1e-30))))
    for p in np.nonzero(~ok)[0]:
        total += _quadrature_log(a[:, p], b[:, p], fun)
    return total


HIP_GENERIC_LANCZOS = [True]   # False: the torch form of the step below at every shape (A/B runs, tests; it stays for P > 16)


def _lanczos_block_generic(operator, Z, steps):
    """The same P independent Lanczos runs for an operator that is NOT one polynomial chain (wrappers around
    a Schur complement): torch vector algebra around `operator.matmul` on [n, P] blocks -- every matmul
    underneath is still HIP launches (the Schur matvec runs its inner HIP CG with P right-hand sides)."""
    # The basis lives as Qt [P, steps + 1, n] (one [steps + 1, n] matrix per probe), so that the two passes of
    # Gram-Schmidt against q_0 .. q_j are four batched matrix-vector products per step whatever j is, and alpha / beta
    # stay on the device until the end.  (A Python loop over the basis vectors -- a product, a column sum and an update
    # each -- was 4 (j + 1) small launches per step, 840 per 20-step run, plus two host reads per step: ~9 ms of an
    # 80 ms semi-supervised epoch.)
    n, P = Z.shape
    wb = lib().mgp_blz_workspace_bytes(n, P, int(steps)) if (HIP_GENERIC_LANCZOS[0] and Z.is_cuda) else 0
    if wb:
        # (round 5) the vector algebra of a step as ONE call of seven launches (mgp_blz_step: the kernels of the block Lanczos
        # over a descriptor) instead of ~20 torch ops on [n, 12] blocks -- after every product, a CG solve that ends in a host
        # wait, the device sat idle while the host issued them: ~2.5 ms of a 57 ms semi-supervised epoch
        Z = _lib.f32c(Z)
        work = _lib.workspace(wb, "blz_generic", Z.device)
        wp, st = ptr(work), stream()
        check(lib().mgp_blz_begin(ptr(Z), n, P, int(steps), wp, wb, st), "mgp_blz_begin")
        off0 = int(lib().mgp_blz_q(n, P, int(steps), 0, wp, wb)) - work.data_ptr()
        Qall = work[off0:off0 + (steps + 1) * n * P * 4].view(torch.float32).view(steps + 1, n, P)
        for j in range(steps):
            W = _lib.f32c(operator.matmul(Qall[j]))
            if W.data_ptr() == Qall[j].data_ptr():
                W = W.clone()
            check(lib().mgp_blz_step(ptr(W), n, P, int(steps), j, wp, wb, st), "mgp_blz_step")
        alpha = (ctypes.c_float * (steps * P))()
        beta = (ctypes.c_float * (steps * P))()
        check(lib().mgp_blz_end(n, P, int(steps), alpha, beta, wp, wb, st), "mgp_blz_end")
        return (np.array(alpha, dtype=np.float64).reshape(steps, P), np.array(beta, dtype=np.float64).reshape(steps, P))
    Qt = torch.empty(P, steps + 1, n, dtype=Z.dtype, device=Z.device)
    Qt[:, 0, :] = (Z / Z.norm(dim=0, keepdim=True).clamp_min(1e-30)).t()
    A = torch.zeros(steps, P, dtype=Z.dtype, device=Z.device)
    B = torch.zeros(steps, P, dtype=Z.dtype, device=Z.device)
    for j in range(steps):
        W = operator.matmul(Qt[:, j, :].t().contiguous())
        Wt = W.t().contiguous().unsqueeze(-1)                 # [P, n, 1]
        Qj = Qt[:, :j + 1, :]                                  # [P, j + 1, n]
        a = torch.zeros(P, dtype=Z.dtype, device=Z.device)
        for _ in range(2):                                     # classical Gram-Schmidt against q_0 .. q_j, twice
            h = torch.bmm(Qj, Wt)                              # [P, j + 1, 1]
            Wt = Wt - torch.bmm(Qj.transpose(1, 2), h)
            a = a + h[:, j, 0]
        b = Wt.squeeze(-1).norm(dim=1)
        A[j], B[j] = a, b
        Qt[:, j + 1, :] = Wt.squeeze(-1) / b.clamp_min(1e-30).unsqueeze(1)
    return A.double().cpu().numpy(), B.double().cpu().numpy()


INVERSE_LANCZOS = [True]      # False: Lanczos over the Schur complement itself (nested solves), as in round 1


class _InverseOperator:
    """v -> op^-1 v as the `matmul` the generic block Lanczos calls."""

    def __init__(self, op):
        self.op = op
        self.shape = op.shape

    def matmul(self, V):
        return self.op._solve(V)


def _inverse_lanczos_target(target):
    """target = [ScaleWrapper of] a SchurComplementOperator over a chain that factorises: its inverse is cheap."""
    from .operators.scale_wrapper_operator import ScaleWrapperOperator
    from .operators.schur_complement_operator import SchurComplementOperator
    inner = target.operator if isinstance(target, ScaleWrapperOperator) else target
    if not isinstance(inner, SchurComplementOperator):
        return None
    from .solvers import _factorisable
    desc = getattr(inner.base, "_descriptor", lambda: None)()
    if desc is None or not _factorisable(desc, {}):
        return None
    return _InverseOperator(target)


def slq_logdet(operator, num_probes=None, steps=None, seed=1337):
    desc = getattr(operator, "_descriptor", lambda: None)()
    if desc is None:
        n = operator.shape[0]
        num_probes = settings.num_trace_samples.value() if num_probes is None else num_probes
        num_probes = -(-num_probes // 4) * 4     # 4 / 8 / 12 / 16 columns: the 16-byte-row SpMM kernel (30 vs 43 us)
        steps = min(n, 20 if steps is None else steps)
        gen = torch.Generator(device="cpu").manual_seed(seed)
        dev = operator.device if hasattr(operator, "device") else None
        Z = rademacher_probes(n, num_probes, seed, _generic_device(operator))
        # A noise wrapper is the polynomial p(Q) = Q - s Q^2 + s^2 Q^3 of what it wraps: log det p(Q) = tr log p(Q) is
        # a spectral function of Q itself, so the Lanczos runs go over Q (ONE nested solve per step for a Schur
        # complement underneath, not three) and the quadrature evaluates log p at the Ritz values.
        from .operators.noise_wrapper_operator import NoiseWrapperOperator
        fun, target = None, operator
        if isinstance(operator, NoiseWrapperOperator):
            sn = float(operator.noise.reshape(-1)[0].item()) if torch.is_tensor(operator.noise) else float(operator.noise)
            target = operator.operator
            fun = lambda th: th - sn * th * th + sn * sn * th * th * th     # noqa: E731
        # A Schur complement underneath: every matvec S v hides a CG on Q_uu (not a product of sparse factors), while
        # S^-1 v = [Q^-1 (v; 0)]_l is ONE solve with the full precision, which factorises (solvers._factorised_solve).
        # tr log f(S) is just as much a spectral function of W = S^-1 (eigenvalues mu = 1 / theta), so the Lanczos runs
        # go over W and the quadrature evaluates log f(1 / mu) at its Ritz values: half the SpMMs per step.
        inv = _inverse_lanczos_target(target) if INVERSE_LANCZOS[0] else None
        with torch.no_grad():
            if inv is not None:
                a, b = _lanczos_block_generic(inv, Z, steps)
                f0 = fun if fun is not None else (lambda th: th)
                total = _quadrature_log_sum(a, b, lambda mu: f0(1.0 / np.maximum(mu, 1e-30)))
            else:
                a, b = _lanczos_block_generic(target, Z, steps)
                total = _quadrature_log_sum(a, b, fun)
        return torch.tensor(n * total / num_probes, dtype=torch.float32, device=Z.device)
    n = desc.n
    num_probes = settings.num_trace_samples.value() if num_probes is None else num_probes
    steps = min(n, 20 if steps is None else steps)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    dev = desc.data.graph.device
    total = 0.0
    # forms 1 / 2 are polynomials of the form-0 chain Q2 (Q2 - s Q2^2 + s^2 Q2^3, I + s Q2): Lanczos over Q2 -- a third
    # of the SpMVs per step for the noise wrapper -- and log p at the Ritz values
    fun = None
    if desc.form in (1, 2):
        sn = float(desc.noise)
        fun = (lambda th: th - sn * th * th + sn * sn * th * th * th) if desc.form == 1 else (lambda th: 1.0 + sn * th)
        desc = desc.with_(form=0)
    if steps + 1 <= 48:
        # all probes as columns of one block (batches of <= 16, a multiple of 4 columns so that the SpMM
        # takes its 16-byte-row kernel): the runs are independent, the launch count drops by the batch size
        num_probes = -(-num_probes // 4) * 4
        Z = rademacher_probes(n, num_probes, seed, dev)
        for c0 in range(0, num_probes, 16):
            a, b = lanczos_tridiag_block(desc, Z[:, c0:c0 + 16].contiguous(), steps)
            total += _quadrature_log_sum(a, b, fun)
    else:
        for _ in range(num_probes):
            z = (torch.randint(0, 2, (n,), generator=gen).float() * 2 - 1).to(dev)
            a, b = lanczos_tridiag(desc, z, steps)
            total += _quadrature_log(a, b, fun)
    return torch.tensor(n * total / num_probes, dtype=torch.float32, device=dev)
