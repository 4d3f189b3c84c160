"""Host-side description of a precision-family operator as ONE chain of fused SpMMs
(mgp_operator_t in include/mgp_hip.h):  A = form(scale * diag(post) (tau I + L_sym)^nu diag(pre))."""
import ctypes
from dataclasses import dataclass, replace
from typing import Optional

import torch

from .. import _lib
from .._lib import OperatorT, check, lib, ptr, stream


@dataclass
class Descriptor:
    data: object                       # graph.LaplacianData (CSR of L_sym + node vectors)
    nu: int
    kappa: float
    pre: Optional[torch.Tensor] = None
    post: Optional[torch.Tensor] = None
    scale: float = 1.0
    form: int = 0                      # 0: Q2, 1: Q2 - s Q2^2 + s^2 Q2^3, 2: I + s Q2
    noise: float = 0.0

    @property
    def n(self):
        return self.data.graph.n

    def with_(self, **kw):
        return replace(self, **kw)

    def masked(self, row_mask=None, col_mask=None):
        """diag(row_mask) A diag(col_mask) for form 0 (MaskedLinearOperator blocks)."""
        assert self.form == 0
        pre, post = self.pre, self.post
        if col_mask is not None:
            pre = col_mask if pre is None else pre * col_mask
        if row_mask is not None:
            post = row_mask if post is None else post * row_mask
        return replace(self, pre=pre, post=post)

    def relabelled(self, wide=False, chain_if_built=False):
        """(descriptor of P A P^T, RelabelledGraph) when the graph's tiles follow a locality order -- the form the iterative
        solvers run on: same operator on permuted vectors -- else (None, None).  wide: for products of 48 columns and more, the
        chain-relabelled matrix where the graph has one (graph.LaplacianData.wide_relabelled).  chain_if_built: the same, but only
        when the graph's chain order EXISTS already (a wide product built it: its host walk, 24 ms at 60k, is not worth starting
        for a solve of a few columns)."""
        if not wide and chain_if_built and getattr(getattr(self.data, "graph", None), "_wide_relabelled", None) is not None:
            wide = True
        name = "wide_relabelled" if wide else "relabelled"
        rel = getattr(self.data, name, getattr(self.data, "relabelled", lambda: None))()
        if rel is None:
            return None, None
        return replace(self, data=rel, pre=rel.permuted(self.pre), post=rel.permuted(self.post)), rel.graph

    def struct(self, wide=False):
        d = self.data
        g = d.graph
        check(lib().mgp_spmm_set_group_hint(g.spmv_lanes), "mgp_spmm_set_group_hint")
        op = OperatorT()
        op.L = d.csr(wide) if wide else d.csr()
        op.pre = self.pre.data_ptr() if self.pre is not None else None
        op.post = self.post.data_ptr() if self.post is not None else None
        op.nu = int(self.nu)
        op.kappa = float(self.kappa)
        op.scale = float(self.scale)
        op.form = int(self.form)
        op.noise = float(self.noise)
        return op

    def apply(self, X):
        """Y = A X for X [n, C] fp32 on device (C chunks of 256)."""
        _lib.require_device(X)
        squeeze = X.dim() == 1
        X = _lib.f32c(X.unsqueeze(-1) if squeeze else X)
        if X.shape[1] >= 48 and hasattr(self.data, "wide_relabelled"):
            # 48 columns and more: the matrix-core tile SpMM, on the chain-relabelled matrix where the graph has one (half the tile
            # image and the work, graph.KnnGraph.wide_relabelled): rows permuted in and out around the product
            rg = self.data.graph.wide_relabelled() if hasattr(self.data.graph, "wide_relabelled") else None
            if rg is not None:
                rdesc, rg = self.relabelled(wide=True)
                return rg.unpermute(rdesc.apply(rg.permute(X)))
        op = self.struct(wide=X.shape[1] >= 48)
        out = torch.empty_like(X)
        C = X.shape[1]
        for c0 in range(0, C, 256):
            Xc = X if C <= 256 else X[:, c0:c0 + 256].contiguous()
            Yc = out if C <= 256 else torch.empty_like(Xc)
            wb = lib().mgp_operator_workspace_bytes(ctypes.byref(op), Xc.shape[1])
            work = _lib.workspace(wb, "operator", X.device)
            check(lib().mgp_operator_apply(ctypes.byref(op), ptr(Xc), Xc.shape[1], ptr(Yc), ptr(work), work.numel(),
                                           stream()), "mgp_operator_apply")
            if Yc is not out:
                out[:, c0:c0 + 256] = Yc
        return out.squeeze(-1) if squeeze else out

    def jacobi(self):
        op = self.struct()
        minv = torch.empty(self.n, dtype=torch.float32, device=self.data.graph.device)
        check(lib().mgp_operator_jacobi(ctypes.byref(op), ptr(minv), stream()), "mgp_operator_jacobi")
        return minv
