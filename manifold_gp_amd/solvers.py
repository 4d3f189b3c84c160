"""Host drivers for the iterative pieces of the path: CG solves, inv_quad_logdet, eigensolves and
the GP posterior.  All arithmetic is in libmgp_hip; this file only sizes workspaces, reads the
gpytorch-style settings and mirrors the call conventions of linear_operator that the reference
relies on."""
import ctypes
import math
import warnings

import torch

from . import _lib
from ._compat import settings
from ._lib import CgParamsT, LanczosParamsT, check, lib, ptr, stream


# ------------------------------------------------------------------------------ CG
RELABEL_SOLVES = [True]        # iterative solves on a graph with locality-ordered tiles run on P A P^T (CgPlan.__init__)
# Column count from which a solve iterates on the CHAIN-relabelled matrix (graph.KnnGraph.wide_relabelled).  48 and more: the
# matrix-core tile SpMM, whose work is the distinct columns of a 16-row tile -- these solves BUILD the chain order where it pays.
# From CHAIN_SOLVE_MIN_C columns on the 64-row tile kernels gain too (their dictionaries shrink the same way: the 12-column SpMM on
# the 60k graph 18.5 -> 15.9 us) and the two permutations of the right-hand side / the solution (~5 us each at 60k x 12) are a
# fraction of one iteration: these solves USE the chain order when a wide product has built it (in training `_average_variance`
# and `eval()` have) and never start the host walk themselves.  Below, a solve is too short to pay for the permutations
# (tools/lab/chain_c1.py: C = 1 53 -> 69 us).
CHAIN_SOLVE_MIN_C = [8]


class CgPlan:
    """A reusable HIP CG solver for one operator descriptor and column count: owns the workspace
    and the captured iteration graph (mgp_cg_plan_* in include/mgp_hip.h)."""

    def __init__(self, desc, C, tol=None, max_iter=None, min_iter=None, stop_mode=None, jacobi=None,
                 check_every=0, use_graph=True, refine=0):
        # A graph handed over without locality carries tiles over a locality order (graph.build_tiles_auto).  On the CSR as
        # given, every SpMV of the solve gathers x and scatters y through that order at 4-byte granularity and every vector
        # update works in the caller's order (1M swiss roll in random order: 134 us per SpMV against 81 for the same points
        # handed over Z-ordered).  The plan therefore iterates on P A P^T -- the relabelled CSR + dictionaries the tile view
        # already is, node vectors permuted once per operator -- and permutes the right-hand side in and the solution out:
        # two gathers of n C floats per solve against hundreds of SpMVs.  Caller-order `rhs` in, caller-order result out
        # (graph_laplacian_operator.py:108-124).  RELABEL_SOLVES[0] = False: iterate in the caller's order (A/B, tests).
        self.C = int(C)
        self._xout = None
        desc = self._bind(desc)
        dev = desc.data.graph.device
        stop_mode = settings.cg_stop_mode.value() if stop_mode is None else stop_mode
        self.params = CgParamsT(
            float(settings.cg_tolerance.value() if tol is None else tol),
            int(settings.max_cg_iterations.value() if max_iter is None else max_iter),
            int((10 if stop_mode == 0 else 0) if min_iter is None else min_iter),
            int(stop_mode), int(check_every), int(bool(use_graph)), int(refine))
        self._jacobi = bool(settings.cg_jacobi_preconditioner.value() if jacobi is None else jacobi)
        self.minv = desc.jacobi() if self._jacobi else None
        wb = lib().mgp_cg_workspace_bytes(ctypes.byref(self.op), self.C)
        if wb == 0:
            raise RuntimeError("mgp_cg_workspace_bytes: unsupported operator / column count %d" % C)
        self.work = torch.empty(wb, dtype=torch.uint8, device=dev)
        self.handle = ctypes.c_void_p(0)
        check(lib().mgp_cg_plan_create(ctypes.byref(self.op), self.C, ptr(self.minv), ctypes.byref(self.params),
                                       ptr(self.work), self.work.numel(), stream(), ctypes.byref(self.handle)),
              "mgp_cg_plan_create")
        self.iters = 0
        self._xview = None
        self._iters, self._status = ctypes.c_int32(0), ctypes.c_int32(0)
        self._iters_ref, self._status_ref = ctypes.byref(self._iters), ctypes.byref(self._status)
        self._resid = (ctypes.c_float * self.C)()
        self._solve_fn, self._applies_fn = lib().mgp_cg_plan_solve, lib().mgp_cg_plan_last_applies
        self.status = 0

    def _bind(self, desc):
        """The descriptor the plan iterates on (relabelled where that pays) + its C struct."""
        self._rg = None
        if RELABEL_SOLVES[0]:
            # 48 columns and more run on the matrix-core tile SpMM: there the chain-relabelled matrix where it pays (round 5)
            rdesc, rg = desc.relabelled(wide=self.C >= 48, chain_if_built=self.C >= CHAIN_SOLVE_MIN_C[0])
            if rdesc is not None:
                desc, self._rg = rdesc, rg
        self.desc = desc
        self.op = desc.struct(wide=self.C >= 48)       # 48 columns and more: the matrix-core tile SpMM (graph.MtPlan)
        return desc

    def rebind(self, desc):
        """Point the plan at another operator of the same structure (mgp_cg_plan_rebind: the same graph at the next epoch's
        hyper-parameters) instead of building a new plan: workspace, flags and executable graphs are kept.  False when the
        library refuses (different structure): the caller builds a new plan."""
        old = (self.desc, self.op, self.minv, self._rg)
        had_rg = self._rg is not None
        desc = self._bind(desc)
        if (self._rg is not None) != had_rg:
            self.desc, self.op, self.minv, self._rg = old
            return False
        minv = desc.jacobi() if self._jacobi else None
        rc = lib().mgp_cg_plan_rebind(self.handle, ctypes.byref(self.op), ptr(minv))
        if rc != 0:
            self.desc, self.op, self.minv, self._rg = old
            return False
        self.minv = minv            # (the captured graphs still name the previous operator's arrays; they are re-recorded before
        return True                 # their next launch and never read through the old pointers -- nothing to keep alive)

    # read on demand: a solve is ~60 us, every ctypes call / list conversion on its way back is a visible fraction
    @property
    def applies(self):
        """operator applies the last solve actually ran (mgp_cg_plan_last_applies)"""
        return self._applies_fn(self.handle)

    @property
    def resid(self):
        return list(self._resid)

    @property
    def complex_shift(self):
        """True when the plan solves its system through the complex factorisation I + c B^2 = (I + i sigma B)(I - i sigma B)
        (COCG on the complex symmetric factor, include/mgp_hip.h: mgp_cg_set_complex_shift): `iters` then counts COCG
        iterations of ONE four-column product with B each."""
        return bool(lib().mgp_cg_plan_is_complex_shift(self.handle))

    def solution_view(self):
        """The plan's own solution buffer as a tensor view (no copy; overwritten by the next solve).  NOTE: when the plan iterates
        on the relabelled matrix P A P^T (`self._rg` set: a graph handed over without locality), the rows of this view are in THAT
        order -- row p is node `self._rg.order[p]` of the caller's -- ; `solve()` and `solution64_view()` return caller-order
        copies."""
        if self._xview is None:                 # the buffer never moves: one view for the life of the plan
            off = int(lib().mgp_cg_plan_x(self.handle)) - self.work.data_ptr()
            nb = self.desc.n * self.C * 4
            self._xview = self.work[off:off + nb].view(torch.float32).view(self.desc.n, self.C)
        return self._xview

    def solution64_view(self):
        """The float64 solution a refined solve accumulated (mgp_cg_plan_x64): view [n, C], overwritten by the
        next solve.  Its float32 rounding is what solve() returns."""
        p = lib().mgp_cg_plan_x64(self.handle)
        if not p or self.params.max_refine <= 0:
            raise RuntimeError("no float64 solution: the plan was created without refinement")
        off = int(p) - self.work.data_ptr()
        nb = self.desc.n * self.C * 8
        v = self.work[off:off + nb].view(torch.float64).view(self.desc.n, self.C)
        return v if self._rg is None else self._rg.unpermute(v)      # (a copy, in the caller's order)

    def solve(self, B, out=None, copy=True):
        if B.device.type != "cuda":
            _lib.require_device(B)
        if B.dtype != torch.float32 or not B.is_contiguous():
            B = _lib.f32c(B)
        assert B.shape == (self.desc.n, self.C)
        if self._rg is not None:
            return self._solve_relabelled(B, out, copy)
        if copy:
            X = torch.empty_like(B) if out is None else out
        else:
            X = None
        # (the ctypes out-parameters live in the plan: a solve is ~65 us, per-call allocations are visible)
        rc = self._solve_fn(self.handle, B.data_ptr(), X.data_ptr() if X is not None else None, self._iters_ref,
                            self._resid, self._status_ref)
        if rc != 0:
            check(rc, "mgp_cg_plan_solve")
        if X is None:
            X = self.solution_view()
        self.iters, self.status = self._iters.value, self._status.value
        return X

    def _solve_relabelled(self, B, out, copy):
        """The solve on P A P^T: right-hand side gathered into the locality order, solution gathered back."""
        rg = self._rg
        Br = rg.permute(B)
        rc = self._solve_fn(self.handle, Br.data_ptr(), None, self._iters_ref, self._resid, self._status_ref)
        if rc != 0:
            check(rc, "mgp_cg_plan_solve")
        self.iters, self.status = self._iters.value, self._status.value
        if copy:
            X = torch.empty_like(B) if out is None else out
        else:                                   # copy=False: a plan-owned buffer, overwritten by the next solve
            if self._xout is None:
                self._xout = torch.empty_like(B)
            X = self._xout
        return rg.unpermute(self.solution_view(), out=X)

    def solve_repeated(self, B, times):
        """A^-times B: `times` solves in a row, each on the previous one's solution (the factors of Q = (tau I + L)^nu are the
        same operator).  On a relabelled plan the right-hand side is permuted in ONCE and the result out once, and nothing but
        the library calls sits between two solves (the host wait that ends a solve leaves the device idle until the next launch
        arrives: two gathers and their Python less per factor).  Returns (X, total iterations, worst status)."""
        if times == 1:
            X = self.solve(B)
            return X, self.iters, self.status
        if B.device.type != "cuda":
            _lib.require_device(B)
        if B.dtype != torch.float32 or not B.is_contiguous():
            B = _lib.f32c(B)
        assert B.shape == (self.desc.n, self.C)
        rg = self._rg
        cur = rg.permute(B) if rg is not None else B
        if getattr(self, "_chain_tmp", None) is None:
            self._chain_tmp = (torch.empty_like(cur), torch.empty_like(cur))      # solve k lands in [k % 2], solve k + 1 reads it
        its, worst = 0, 0
        for k in range(times):
            last = k == times - 1
            dst = None if last else self._chain_tmp[k % 2].data_ptr()
            rc = self._solve_fn(self.handle, cur.data_ptr(), dst, self._iters_ref, self._resid, self._status_ref)
            if rc != 0:
                check(rc, "mgp_cg_plan_solve")
            its += self._iters.value
            worst = max(worst, self._status.value)
            if self._status.value == 3:
                break
            if not last:
                cur = self._chain_tmp[k % 2]
        self.iters, self.status = its, worst
        X = self.solution_view()
        return (rg.unpermute(X) if rg is not None else X.clone()), its, worst

    def close(self):
        if self.handle:
            if lib().mgp_cg_plan_poisoned(self.handle):
                # a solve timed out (dead peer): queued work still references the plan's buffers -- keep them alive, free
                # nothing, synchronise nothing; the process is expected to exit non-zero (include/mgp_hip.h)
                _lib.leak(self.__dict__.copy())
            lib().mgp_cg_plan_destroy(self.handle)
            self.handle = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 8
REBIND_PLANS = [True]          # lab / test switch: False = a new plan for every new operator value (rounds 1-4)


def _cached_plan(desc, C, kw):
    """Plans own a workspace, host-mapped flags and captured hipGraphs (~0.9 ms of host time over a plan's life): they are
    kept per operator STRUCTURE -- graph, nu, form, which of pre / post are present, column count, solver settings, stream --
    and pointed at the current operator VALUES (bandwidth, length scale, scale, noise, the pre / post vectors) when those
    changed since the plan's last use (CgPlan.rebind): the nested CG of a Schur-complement matvec and `_average_variance` reuse
    one plan within an epoch, and the next epoch -- new hyper-parameters, same graph -- reuses it again."""
    # a plan is bound to the stream that was current when it was created (its launches, its graph replays and the
    # copy of X are ordered there only): a solve issued under another torch.cuda.stream gets its own plan.
    # (graph / data uids are assigned monotonically -- an id() could be reused by a new object once the old one is gone; the
    # pre / post pointers in the value key are safe because a plan keeps its descriptor, hence those tensors, alive)
    g = getattr(desc.data, "graph", None)
    skey = (getattr(g, "uid", None) or id(g), desc.nu, desc.form, desc.pre is not None, desc.post is not None,
            int(C), settings.cg_tolerance.value(), settings.max_cg_iterations.value(), settings.cg_stop_mode.value(),
            settings.cg_jacobi_preconditioner.value(), tuple(sorted(kw.items())),
            int(torch.cuda.current_stream(desc.data.graph.device).cuda_stream))
    vkey = (getattr(desc.data, "uid", None) or id(desc.data), desc.kappa, desc.scale, desc.noise,
            desc.pre.data_ptr() if desc.pre is not None else 0, desc.post.data_ptr() if desc.post is not None else 0)
    if not REBIND_PLANS[0]:
        skey = skey + vkey
    plan = _PLAN_CACHE.get(skey)
    if plan is not None and plan._vkey != vkey:
        if plan.rebind(desc):
            plan._vkey = vkey
        else:
            _PLAN_CACHE.pop(skey).close()
            plan = None
    if plan is None:
        if len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE))).close()
        plan = CgPlan(desc, C, **kw)
        plan._vkey = vkey
        _PLAN_CACHE[skey] = plan
    return plan


def clear_plan_cache():
    while _PLAN_CACHE:
        _PLAN_CACHE.popitem()[1].close()


FACTORISED_SOLVES = [True]      # lab / test switch: False = CG on the whole chain Q = (tau I + L)^nu even when it factorises
# Each factor of the first round is solved to tol / (divisor * nu).  divisor 2 leaves the whole system under tol when the
# later factors' residuals are not amplified (direct solves with Q in supervised training, the 100 one-hot columns of
# _average_variance: one round, worst true residual 0.3-0.6 tol); the Schur complement's solves (right-hand sides padded with
# zeros on the unlabelled rows) needed a second round 53 times out of 65 per epoch at 2 and never at 4: 42 -> 29 iterations
# per solve, semi-supervised epoch 92 -> 80 ms on one box (tools/lab/prof_training.py).  So the divisor starts at
# FACTOR_TOL_DIVISOR and a (graph, column count) whose solve needed a second round is solved with twice that from then on.
FACTOR_TOL_DIVISOR = [2.0]
_FACTOR_DIVISOR_OF = {}
FACTOR_ROUNDS_LOG = None        # lab: a list here collects (rounds, iterations, worst true residual / tol) per factorised solve


def _factorisable(desc, kw):
    """Q = scale D^1/2 (tau I + L_sym)^nu D^1/2 (randomwalk) or scale (tau I + L_sym)^nu (symmetric), nu >= 2, no mask folded
    into pre / post, no refinement rounds asked for: Q^-1 is nu solves with B = tau I + L_sym."""
    if not FACTORISED_SOLVES[0] or desc.form != 0 or int(desc.nu) < 2 or kw.get("refine", 0):
        return False
    if desc.pre is None and desc.post is None:
        return True
    sq = getattr(desc.data, "dsqrt", None)
    return (desc.pre is not None and desc.post is not None and sq is not None
            and desc.pre.data_ptr() == sq.data_ptr() and desc.post.data_ptr() == sq.data_ptr())


def _factorised_solve(desc, B, kw):
    """nu sequential CG solves with B = tau I + L_sym instead of one with B^nu: cond(B) = cond(Q)^(1/nu), and an iteration
    costs ONE SpMM instead of nu.  Measured on the 60k graph, nu = 2, 12 Gaussian columns (tools/lab/cg_iters.py):
    tol 1e-2: 94 iterations / 5.3 ms -> 11 + 13 iterations / 1.5 ms with a TRUE residual of 4.5e-3 instead of 9.8e-3;
    tol 1e-6: 328 iterations / 17.7 ms (true residual 1.7e-5: the fp32 recurrence on Q stalls there) -> 26 + 27
    iterations / 3.0 ms (5.1e-6).
    With r_k the residual of factor k the residual of the whole system is r_1 + B r_2 + B^2 r_3 ...: each factor is solved
    to tol / (2 nu), which leaves the whole under tol / 2 on well conditioned graphs, but B amplifies the later factors'
    residuals by up to cond(B)^(k-1) (dumbbell, eps = 0.05, nu = 3: 5 tol).  So the TRUE residual b - Q x is formed (one
    apply) and, while it is above tol, a correction Q d = r is solved the same way (to the looser tolerance that remains)
    and added: at most three rounds, usually one, and `resid` is the true relative residual.  A (graph, column count) that
    needed a second round starts tighter the next time (FACTOR_TOL_DIVISOR above)."""
    import math
    nu = int(desc.nu)
    dB = desc.with_(nu=1, kappa=desc.kappa / math.sqrt(nu), scale=1.0, pre=None, post=None)
    kw = dict(kw)
    kw.pop("jacobi", None)                      # diag(B) = tau + diag(L_sym) is nearly constant: nothing to gain
    tol = kw.get("tol", None)
    tol = float(settings.cg_tolerance.value() if tol is None else tol)
    stop_mode = kw.get("stop_mode", None)
    stop_mode = int(settings.cg_stop_mode.value() if stop_mode is None else stop_mode)
    dinv = None if desc.pre is None else desc.data.dinvsqrt.view(-1, 1)
    bn = B.norm(dim=0).clamp_min(1e-30)
    X, R, its, rel = None, B, 0, None
    want, prev_worst = tol, float("inf")
    hint = (id(getattr(desc.data, "graph", None)), int(B.shape[1]))
    div0 = _FACTOR_DIVISOR_OF.get(hint, FACTOR_TOL_DIVISOR[0])
    for rnd in range(3):
        kw["tol"] = max(want, 1e-7) / ((div0 if rnd == 0 else 8.0) * nu)
        Y = R if dinv is None else (R * dinv).contiguous()
        plan = _cached_plan(dB, Y.shape[1], kw)
        Y, its_k, status = plan.solve_repeated(Y, nu)       # (a fresh tensor)
        its += its_k
        if status == 3:
            raise RuntimeError("NaNs encountered in CG")
        if dinv is not None:
            Y = Y * dinv
        if desc.scale != 1.0:
            Y = Y / desc.scale
        X = Y if X is None else X + Y
        R = B - desc.apply(X)
        rel = R.norm(dim=0) / bn
        worst = float(rel.mean() if stop_mode == 0 else rel.max())
        if not (worst > tol) or not math.isfinite(worst):
            break
        if rnd > 0 and worst > 0.5 * prev_worst:  # fp32 floor of the true residual (|A||x| eps32 / |b|): no point in going on
            break
        prev_worst = worst
        want = min(0.5, tol / worst)              # relative to the new right-hand side R
    if rnd > 0 and div0 < 4.0 * FACTOR_TOL_DIVISOR[0]:
        if len(_FACTOR_DIVISOR_OF) > 256:
            _FACTOR_DIVISOR_OF.clear()
        _FACTOR_DIVISOR_OF[hint] = 2.0 * div0
    if FACTOR_ROUNDS_LOG is not None:
        FACTOR_ROUNDS_LOG.append((rnd + 1, its, worst / tol))
    if worst > max(tol, 2e-5):
        warnings.warn("factorised CG solve: true residual %.3g above the tolerance %.3g after %d rounds" % (worst, tol, rnd + 1))
    return X, its, [float(v) for v in rel.tolist()]


def cg_solve(desc, rhs, **kw):
    """Solve A X = rhs with the HIP CG.  Returns (X, iterations, relative residuals).  A form-0 chain with nu >= 2 is
    solved factor by factor (_factorised_solve); iterations then counts all factors' iterations (one SpMM each) and the
    residuals are those of the last factor."""
    squeeze = rhs.dim() == 1
    B = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
    kw = dict(kw)
    factorise = kw.pop("factorise", True)          # factorise=False: CG on the whole chain (comparisons, tests)
    if factorise and _factorisable(desc, kw) and B.shape[1] <= 256:
        X, its, res = _factorised_solve(desc, B, kw)
        return (X.squeeze(-1) if squeeze else X), its, res
    outs, its, res = [], 0, []
    for c0 in range(0, B.shape[1], 256):
        Bc = B if B.shape[1] <= 256 else B[:, c0:c0 + 256].contiguous()
        plan = _cached_plan(desc, Bc.shape[1], kw)
        outs.append(plan.solve(Bc))
        its = max(its, plan.iters)
        res += plan.resid
        if plan.status == 2:
            warnings.warn("CG did not converge in %d iterations (residuals up to %.3g)"
                          % (plan.iters, max(plan.resid)))
        elif plan.status == 3:
            raise RuntimeError("NaNs encountered in CG")          # linear_cg raises on NaN too
    X = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    return (X.squeeze(-1) if squeeze else X), its, res


def generic_cg(operator, rhs, tol=None, max_iter=None, x0=None, precond=None):
    """linear_cg's recurrence in torch ops for operators that are not one polynomial chain (e.g.
    wrappers around a Schur complement).  Every `_matmul` underneath is still a HIP launch.
    x0: optional initial guess; precond: optional callable v -> M v with M ~ A^-1 (preconditioned CG).  With either
    the stopping rule (mean relative residual < tol) is tested from the start -- linear_cg's minimum of 10
    iterations belongs to its cold, unpreconditioned start from zero."""
    _lib.require_device(rhs)
    tol = settings.cg_tolerance.value() if tol is None else tol
    max_iter = settings.max_cg_iterations.value() if max_iter is None else max_iter
    squeeze = rhs.dim() == 1
    B = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
    bn = B.norm(dim=0, keepdim=True).clamp_min(1e-10)
    Bn = B / bn
    min_iter = 10 if (x0 is None and precond is None) else 0
    if x0 is None:
        x = torch.zeros_like(Bn)
        r = Bn.clone()
    else:
        x = _lib.f32c(x0.unsqueeze(-1) if squeeze else x0) / bn
        r = Bn - operator._matmul(x)
    if min_iter == 0 and r.norm(dim=0).mean().item() < tol:
        x = x * bn
        return x.squeeze(-1) if squeeze else x
    z = precond(r) if precond is not None else r
    p = z.clone()
    rz = (r * z).sum(0, keepdim=True)
    for it in range(1, max_iter + 1):
        q = operator._matmul(p)
        pq = (p * q).sum(0, keepdim=True)
        alpha = rz / pq.where(pq.abs() > 1e-30, torch.full_like(pq, 1e-30))
        x = x + alpha * p
        r = r - alpha * q
        if it >= min_iter and r.norm(dim=0).mean().item() < tol:
            break
        z = precond(r) if precond is not None else r
        rz_new = (r * z).sum(0, keepdim=True)
        p = z + (rz_new / rz.where(rz.abs() > 1e-30, torch.full_like(rz, 1e-30))) * p
        rz = rz_new
    x = x * bn
    return x.squeeze(-1) if squeeze else x


def solve_with_tridiag(operator, sol, rhs, num_tridiag):
    """linear_operator's convention for `_solve(rhs, preconditioner, num_tridiag)`: the solution alone when
    num_tridiag == 0, else (solution, T) with T [num_tridiag, k, k] the Lanczos tridiagonals of the operator started
    at the first `num_tridiag` columns of rhs (what linear_cg assembles from its CG coefficients for the stochastic
    Lanczos quadrature; k = max_lanczos_quadrature_iterations).  The tridiagonals come from the HIP block Lanczos."""
    if not num_tridiag:
        return sol
    from .slq import _lanczos_block_generic, lanczos_tridiag_block
    k = min(int(settings.max_lanczos_quadrature_iterations.value()), operator.shape[-1])
    Z = _lib.f32c(rhs[:, :num_tridiag])
    desc = getattr(operator, "_descriptor", lambda: None)()
    with torch.no_grad():
        if desc is not None and k + 1 <= 48:
            parts = [lanczos_tridiag_block(desc, Z[:, c0:c0 + 16].contiguous(), k) for c0 in range(0, num_tridiag, 16)]
            a = torch.cat([torch.from_numpy(p[0]) for p in parts], 1)
            b = torch.cat([torch.from_numpy(p[1]) for p in parts], 1)
        else:
            an, bn = _lanczos_block_generic(operator, Z, k)
            a, b = torch.from_numpy(an), torch.from_numpy(bn)
    T = torch.zeros(num_tridiag, k, k, dtype=torch.float32, device=rhs.device)
    idx = torch.arange(k, device=rhs.device)
    T[:, idx, idx] = a.t().float().to(rhs.device)
    if k > 1:
        off = b[:k - 1].t().float().to(rhs.device)
        T[:, idx[:-1], idx[1:]] = off
        T[:, idx[1:], idx[:-1]] = off
    return sol, T


# ------------------------------------------------------------------------------ inv_quad / logdet
def inv_quad_logdet(operator, inv_quad_rhs=None, logdet=False, reduce_inv_quad=True):
    """linear_operator's inv_quad_logdet as the reference uses it (precision_matern_operator.py:53,
    train_model.py:68, test_model.py:23): dense Cholesky of to_dense() when N <= max_cholesky_size,
    iterative otherwise (HIP CG for the quadratic form; stochastic Lanczos quadrature for logdet).

    Gradients (training): the dense branch is differentiable end to end (to_dense()
    runs the differentiable fused SpMM).  The iterative branch returns the iterative VALUE plus a
    zero-valued surrogate that carries the gradient, the way linear_operator's InvQuadLogdet does:
        d logdet A = E_z[(A^-1 z)^T dA z],      d (b^T A^-1 b) = -(A^-1 b)^T dA (A^-1 b)
    with the solves detached and one differentiable operator application each."""
    from .autograd import needs_grad
    n = operator.shape[-1]
    inv_quad = None
    logdet_term = None
    dense_ok = n <= settings.max_cholesky_size.value()
    grad = needs_grad(*getattr(operator, "_hyper_tensors", lambda: [])())
    chol = None
    if dense_ok:
        A = operator.to_dense()
        A = 0.5 * (A + A.t())
        chol = torch.linalg.cholesky(A.double())
    if inv_quad_rhs is not None:
        rhs = inv_quad_rhs if inv_quad_rhs.dim() == 2 else inv_quad_rhs.unsqueeze(-1)
        if chol is not None:
            sol = torch.cholesky_solve(rhs.double(), chol).float()
            iq = (rhs * sol).sum(0)
        else:
            with torch.no_grad():
                sol = operator.solve(rhs.detach())
            if rhs.requires_grad and torch.is_grad_enabled():
                # value b^T A^-1 b with d/db = 2 A^-1 b (sol detached): the dense branch differentiates the same way
                iq = 2.0 * (rhs * sol).sum(0) - (rhs.detach() * sol).sum(0)
            else:
                iq = (rhs * sol).sum(0)
            if grad:
                sur = -(sol * operator.matmul(sol)).sum(0)
                iq = iq + (sur - sur.detach())
        inv_quad = iq.sum() if reduce_inv_quad else iq
    if logdet:
        if chol is not None:
            logdet_term = (2.0 * chol.diagonal().log().sum()).float()
        else:
            from .slq import slq_logdet
            with torch.no_grad():
                logdet_term = slq_logdet(operator)
            if grad:
                # probes rounded up to a multiple of 4: the 4 / 8 / 12 / 16-column SpMM kernel (16-byte rows,
                # DPP reductions) is ~2x faster per launch than the generic one, and more probes only help
                P = -(-settings.num_trace_samples.value() // 4) * 4
                from .slq import rademacher_probes
                Z = rademacher_probes(n, P, 4321, logdet_term.device)
                with torch.no_grad():
                    S = operator.solve(Z)
                sur = (S * operator.matmul(Z)).sum() / P
                logdet_term = logdet_term + (sur - sur.detach())
    return inv_quad, logdet_term


def dense_symeig(operator):
    """`symeig` branch of LinearOperator.diagonalization: eigh of to_dense() (small N only)."""
    A = operator.to_dense()
    evals, evecs = torch.linalg.eigh(0.5 * (A + A.t()))
    return evals, evecs


# ------------------------------------------------------------------------------ Lanczos
class EigenFloorWarning(UserWarning):
    """mgp_lanczos_smallest returned MGP_OK with info[2] < m (residual floor / unseparable guards)."""


def lanczos_smallest(lap_data, m, tol=1e-5, max_basis=0, degree=0, max_restarts=60, seed=1337, return_block=False, warm=None,
                     keep_warm=False):
    """m smallest eigenpairs of L_sym (CSR in `lap_data`) by the HIP filtered block iteration.
    Returns (evals[m] device, evecs[n,m] device, resid[m] host list); with return_block=True a fourth item
    dict(evals [b] device, evecs [n, b] device, resid [b] list): the whole Rayleigh-Ritz block the solver ended with,
    guard columns included (what the independent float64 check of the spectral stage starts from).
    warm: the `lanczos_smallest.last_warm` of an earlier call on the same graph (same sparsity pattern, same m): the solve starts
    from that block instead of a random one (mgp_lanczos_smallest_warm).  keep_warm=True leaves this call's block there."""
    g = lap_data.graph
    dev = g.device
    check(lib().mgp_spmm_set_group_hint(g.spmv_lanes), "mgp_spmm_set_group_hint")
    # A graph whose nodes arrive without locality carries tiles over a locality order (graph.build_tiles_auto).
    # The block iteration gathers an X row per entry: on the CSR as given those rows are scattered over a
    # block of n x b floats (HBM-bound, 4.3 ms per 84-column SpMM at N = 1M); on the SAME matrix relabelled
    # by that order (P L P^T: same spectrum, eigenvectors permuted back below) they sit in cache.
    order = None
    # (round 5: for the block products the matrix relabelled by the graph's nearest-neighbour CHAIN order where that shrinks the
    # dense 16-row tiles -- graph.KnnGraph.wide_relabelled --, else the graph's own locality order)
    rel = getattr(lap_data, "wide_relabelled", getattr(lap_data, "relabelled", lambda: None))()
    if rel is not None:
        order = rel.graph.order
        csr = rel.csr(wide=True)
    else:
        csr = lap_data.csr(wide=True)
    prm = LanczosParamsT(int(max_basis), int(degree), int(max_restarts), float(tol), int(seed))
    wb = lib().mgp_lanczos_workspace_bytes(g.n, int(m), ctypes.byref(prm))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    evals = (ctypes.c_float * m)()
    resid = (ctypes.c_float * m)()
    info = (ctypes.c_int32 * 4)()
    evecs = torch.empty(g.n, m, dtype=torch.float32, device=dev)
    b = int(lib().mgp_lanczos_block_size(int(m), ctypes.byref(prm))) if (return_block or keep_warm or warm is not None) else 0
    b = min(b, g.n)
    bev = (ctypes.c_float * b)() if b else None
    bres = (ctypes.c_float * b)() if b else None
    bvec = torch.empty(g.n, b, dtype=torch.float32, device=dev) if b else None
    if warm is not None and (warm["block"].shape != (g.n, b) or warm["m"] != int(m) or warm["ordered"] != (order is not None)):
        warm = None                                          # another shape / another row order: cold start
    if warm is not None:
        wev = (ctypes.c_float * b)(*warm["evals"])
        rc = lib().mgp_lanczos_smallest_warm(ctypes.byref(csr), int(m), ctypes.byref(prm), evals, ptr(evecs), resid, info,
                                             bev, ptr(bvec), bres, ptr(warm["block"]), wev, ptr(work), work.numel(), stream())
    else:
        rc = lib().mgp_lanczos_smallest_ex(ctypes.byref(csr), int(m), ctypes.byref(prm), evals, ptr(evecs), resid, info,
                                           bev, ptr(bvec) if b else None, bres, ptr(work), work.numel(), stream())
    if keep_warm and b:
        # (in the SOLVER's row order -- the relabelled one when the graph has a locality order --: what the next call hands back)
        lanczos_smallest.last_warm = dict(block=bvec.clone() if order is not None and return_block else bvec, evals=list(bev), m=int(m),
                                          ordered=order is not None)
    if rc == -4:      # MGP_ERR_NOT_CONVERGED: the best block is returned, residuals say how good it is
        warnings.warn("eigensolver stopped after %d rounds with %d/%d pairs below tol (max residual %.3g)"
                      % (info[0], info[2], m, max(resid)))
    else:
        check(rc, "mgp_lanczos_smallest")
        if info[2] < m:
            # MGP_OK with fewer than m pairs under tol: the solver stopped at the residual floor of fp32 arithmetic (or
            # could not separate the wanted block from its guards at the degree cap).  Not silent: the reference runs a
            # dense eigh here (riemann_kernel.py:124), the caller must see how far from `tol` the block is
            warnings.warn("eigensolver stopped at the fp32 residual floor after %d rounds with %d/%d pairs below tol "
                          "(max residual %.3g, tol %.3g of lambda_max; residuals are in the third return value)"
                          % (info[0], info[2], m, max(resid), tol), EigenFloorWarning)
    ev = torch.tensor(list(evals), dtype=torch.float32, device=dev)
    lanczos_smallest.last_info = list(info)
    if order is not None:
        out = torch.empty_like(evecs)
        out[order] = evecs                                   # row p of the relabelled problem is node order[p]
        evecs = out
        if b:
            outb = torch.empty_like(bvec)
            outb[order] = bvec
            bvec = outb
    if return_block:
        return ev, evecs, list(resid), dict(evals=torch.tensor(list(bev), dtype=torch.float32, device=dev), evecs=bvec,
                                            resid=list(bres))
    return ev, evecs, list(resid)


# ------------------------------------------------------------------------------ posterior
def lowrank_apply(Z, X, alpha, beta):
    """alpha * Z (Z^T X) + beta * X on device (K + sigma^2 I with K = outputscale Z Z^T)."""
    _lib.require_device(Z, X)
    squeeze = X.dim() == 1
    Xc = _lib.f32c(X.unsqueeze(-1) if squeeze else X)
    Z = _lib.f32c(Z)
    n, m = Z.shape
    Y = torch.empty_like(Xc)
    wb = lib().mgp_lowrank_workspace_bytes(m, Xc.shape[1])
    work = _lib.workspace(wb, "lowrank", Z.device)
    check(lib().mgp_lowrank_apply(ptr(Z), n, m, ptr(Xc), Xc.shape[1], float(alpha), float(beta), ptr(Y), ptr(work),
                                  work.numel(), stream()), "mgp_lowrank_apply")
    return Y.squeeze(-1) if squeeze else Y


def lowrank_cg(Z, y, outputscale, noise, tol=1e-6, max_iter=500):
    """(outputscale Z Z^T + noise I)^-1 y by CG on device -- the '(K + sigma^2 I) x = y' solve of
    the spectral kernel (reference: gpytorch's Woodbury path)."""
    x = torch.zeros_like(y)
    r = y.clone()
    p = r.clone()
    rz = torch.dot(r, r)
    bn = rz.sqrt()
    it = 0
    for it in range(1, max_iter + 1):
        q = lowrank_apply(Z, p, outputscale, noise)
        alpha = rz / torch.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rz_new = torch.dot(r, r)
        if rz_new.sqrt() <= tol * bn:
            break
        p = r + (rz_new / rz) * p
        rz = rz_new
    return x, it


def gram_f64(A):
    """A^T A for a tall fp32 block A [n, b <= 512], fp64 accumulation on the device (mgp_gram_f64) -> f64 [b, b]."""
    _lib.require_device(A)
    A = _lib.f32c(A)
    n, b = A.shape
    G = torch.empty(b, b, dtype=torch.float64, device=A.device)
    wb = lib().mgp_gram_workspace_bytes(n, b)
    work = _lib.workspace(wb, "gram", A.device)
    check(lib().mgp_gram_f64(ptr(A), n, b, ptr(G), ptr(work), work.numel(), stream()), "mgp_gram_f64")
    return G


def woodbury(Z, y, outputscale, noise):
    """The pieces of (outputscale Z Z^T + noise I)^-1 y on the m x m root, the way gpytorch evaluates a
    LowRankRootAddedDiagLinearOperator: G = Z^T Z and Z^T y from ONE fp64-accumulating HIP
    Gram pass over [Z | y] (mgp_gram_f64; rocBLAS' dgemm takes 53 ms for the 1M x 50 shape, 3 ms at 60k x 100),
    the m x m system in fp64 through torch (POTRF on a 100 x 100 matrix), the solution rows summed in fp64
    (mgp_lowrank_residual).  Returns dict(G [m,m] f64, Lc, ZTy [m,C] f64, t [m,C] f64, alpha [n,C] f32)."""
    _lib.require_device(Z, y)
    Z = _lib.f32c(Z)
    n, m = Z.shape
    Y = _lib.f32c(y.reshape(n, -1))
    C = Y.shape[1]
    eye = torch.eye(m, dtype=torch.float64, device=Z.device)
    if m + C <= 512 and m * C <= 6144:
        Gb = gram_f64(torch.cat([Z, Y], 1))
        G = 0.5 * (Gb[:m, :m] + Gb[:m, :m].t())
        ZTy = Gb[:m, m:].contiguous()
        Lc = torch.linalg.cholesky(G + (noise / outputscale) * eye)
        t = torch.cholesky_solve(ZTy, Lc).contiguous()
        alpha = torch.empty_like(Y)
        check(lib().mgp_lowrank_residual(ptr(Z), n, m, ptr(t), ptr(Y), C, 1.0 / float(noise), ptr(alpha), stream()),
              "mgp_lowrank_residual")
    else:                                   # very wide right-hand sides: library GEMMs in fp64
        Zd = Z.double()
        G = Zd.t() @ Zd
        ZTy = Zd.t() @ Y.double()
        Lc = torch.linalg.cholesky(G + (noise / outputscale) * eye)
        t = torch.cholesky_solve(ZTy, Lc)
        alpha = ((Y.double() - Zd @ t) / noise).float()
    return dict(G=G, Lc=Lc, ZTy=ZTy, t=t, alpha=alpha.reshape(y.shape))


def lowrank_solve(Z, y, outputscale, noise):
    """(outputscale Z Z^T + noise I)^-1 y by Woodbury on the m x m root -- the direct form of lowrank_cg."""
    return woodbury(Z, y, outputscale, noise)["alpha"]


def kernel_block(Z1, Z2, scale=1.0):
    """K = scale * Z1 Z2^T on the fp32 MFMA."""
    _lib.require_device(Z1, Z2)
    Z1, Z2 = _lib.f32c(Z1), _lib.f32c(Z2)
    K = torch.empty(Z1.shape[0], Z2.shape[0], dtype=torch.float32, device=Z1.device)
    check(lib().mgp_kernel_block(ptr(Z1), Z1.shape[0], ptr(Z2), Z2.shape[0], Z1.shape[1], float(scale), ptr(K),
                                 stream()), "mgp_kernel_block")
    return K


def kernel_diag(Z1, Z2, scale=1.0):
    _lib.require_device(Z1, Z2)
    Z1, Z2 = _lib.f32c(Z1), _lib.f32c(Z2)
    out = torch.empty(Z1.shape[0], dtype=torch.float32, device=Z1.device)
    check(lib().mgp_kernel_diag(ptr(Z1), ptr(Z2), Z1.shape[0], Z1.shape[1], float(scale), ptr(out), stream()),
          "mgp_kernel_diag")
    return out
