"""SchurComplementOperator (manifold_gp/operators/schur_complement_operator.py:12-36):
Q_ll - Q_lu Q_uu^-1 Q_ul on the labelled nodes.

The reference builds three MaskedLinearOperators per matvec and runs linear_cg on Q_uu.  Here the
four blocks are the SAME polynomial chain with a 0/1 mask folded into its input / output row
scalings, acting on full-length zero-padded vectors, and the inner solve is the HIP CG."""
import torch

from .. import _lib
from .._compat import LinearOperator


class _SchurMatmul(torch.autograd.Function):
    """S(theta) v with gradients.  With w = Q_uu^-1 Q_ul v and the full-length vectors V = [v; -w],
    U = [g; -z] (z = Q_uu^-1 Q_ul g), u^T S v = U^T Q V and, because w and z are stationary points of
    that bilinear form, d(g^T S v)/d theta = U^T (dQ/d theta) V with U, V held fixed: one more inner solve
    and one differentiable precision matmul in the backward pass, no differentiation through the CG.
    The rhs gradient is S g (S is symmetric for the symmetric precision operators of the reference)."""

    @staticmethod
    def forward(ctx, v, op, *hyper):
        from .._compat import settings
        # the backward pass runs outside the caller's settings context: keep the solver settings of the
        # forward pass for the inner solve of the backward pass
        ctx.cg = dict(tol=settings.cg_tolerance.value(), max_iter=settings.max_cg_iterations.value(),
                      stop_mode=settings.cg_stop_mode.value())
        res, V = op._apply_parts(v, **ctx.cg)
        ctx.op = op
        ctx.save_for_backward(V, *hyper)
        return res

    @staticmethod
    def backward(ctx, g):
        op = ctx.op
        V = ctx.saved_tensors[0]
        hyper = ctx.saved_tensors[1:]
        Sg, U = op._apply_parts(_lib.f32c(g), **ctx.cg)
        grads = [None] * len(hyper)
        need = [i for i, h in enumerate(hyper) if h.requires_grad and ctx.needs_input_grad[2 + i]]
        if need:
            with torch.enable_grad():
                val = (U * op.base._matmul(V)).sum()
                out = torch.autograd.grad(val, [hyper[i] for i in need], allow_unused=True)
            for i, gi in zip(need, out):
                grads[i] = gi
        return (Sg if ctx.needs_input_grad[0] else None, None, *grads)


class SchurComplementOperator(LinearOperator):
    def __init__(self, base, mask):
        super().__init__(base, mask)
        self.base = base
        self.mask = mask
        self._fmask = None
        self._lidx = None

    def _masks(self):
        if self._fmask is None:
            dev = self.base.laplacian.x.device if hasattr(self.base, "laplacian") else self.mask.device
            m = self.mask.to(dev)
            self._fmask = (m.float().contiguous(), (~m).float().contiguous())
            self._lidx = torch.nonzero(m, as_tuple=False).squeeze(-1)
        return self._fmask

    def _hyper_tensors(self):
        return getattr(self.base, "_hyper_tensors", lambda: [])()

    def _masked_blocks(self, desc, ml, mu):
        """The three masked descriptors of one matvec, kept while the base operator's values do not change: the
        mask is folded into NEW pre / post vectors, and the CG plan cache keys on their addresses -- rebuilt per
        call, every inner solve of a Lanczos / CG loop created (and evicted) a plan (0.5 ms each way)."""
        key = (id(desc.data), desc.nu, desc.kappa, desc.scale, desc.form, desc.noise,
               desc.pre.data_ptr() if desc.pre is not None else 0, desc.post.data_ptr() if desc.post is not None else 0)
        if getattr(self, "_blocks_key", None) != key:
            self._blocks = (desc.masked(col_mask=ml), desc.masked(row_mask=mu, col_mask=mu),
                            desc.masked(row_mask=ml, col_mask=mu))
            self._blocks_key = key
            self._blocks_owner = desc          # keeps desc.data / pre / post alive: their ids and addresses cannot be recycled
        return self._blocks

    def _apply_parts(self, v, **cg_kw):
        """(S v [k, C], V [n, C]) with V = [v on the labelled nodes; -Q_uu^-1 Q_ul v on the others]."""
        from ..solvers import cg_solve
        with torch.no_grad():
            ml, mu = self._masks()
            desc = self.base._descriptor()
            n = desc.n
            full = torch.zeros(n, v.shape[1], device=v.device, dtype=torch.float32)
            full[self._lidx] = v
            d_l, d_uu, d_lu = self._masked_blocks(desc, ml, mu)
            tmp = d_l.apply(full)                                      # Q[:, l] v           (:27)
            # (Jacobi-preconditioned: the diagonal of Q = D^1/2 (tau I + L)^nu D^1/2 varies with the degrees; 35 -> 26
            # iterations per inner solve at C4's size.  The reference's linear_cg runs unpreconditioned -- same answer)
            sol = cg_solve(d_uu, tmp * mu.view(-1, 1), jacobi=True, **cg_kw)[0]     # Q_uu^-1 (:28)
            out = d_lu.apply(sol)                                      # Q_lu (.)            (:29)
            res = (tmp - out)[self._lidx]                              # (:30)
            return res, full - sol * mu.view(-1, 1)

    def _matmul(self, rhs):
        from ..autograd import needs_grad
        _lib.require_device(rhs)
        squeeze = rhs.dim() == 1
        v = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
        hyper = self._hyper_tensors()
        if needs_grad(rhs, *hyper):
            res = _SchurMatmul.apply(v, self, *hyper)
        else:
            res = self._apply_parts(v)[0]
        return res.squeeze(-1) if squeeze else res

    def _solve_hip(self, rhs):
        """S^-1 b without nesting: by block elimination the labelled part of Q^-1 [b; 0] IS S^-1 b, so one
        HIP CG on the full precision replaces a CG on S whose every matvec hides another CG on Q_uu
        (SURVEY.md section 8f-3)."""
        from ..solvers import cg_solve
        _lib.require_device(rhs)
        squeeze = rhs.dim() == 1
        v = _lib.f32c(rhs.unsqueeze(-1) if squeeze else rhs)
        self._masks()
        desc = self.base._descriptor()
        full = torch.zeros(desc.n, v.shape[1], device=v.device, dtype=torch.float32)
        full[self._lidx] = v
        sol = cg_solve(desc, full, jacobi=True)[0][self._lidx]
        return sol.squeeze(-1) if squeeze else sol

    def _size(self):
        k = int(self.mask.sum())
        return torch.Size([k, k])

    def _transpose_nonbatch(self):
        return SchurComplementOperator(self.base._transpose_nonbatch(), self.mask)
