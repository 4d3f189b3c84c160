"""Row-partitioned multi-GPU CG (one process per GPU, RCCL over xGMI).

The reference has no distributed code.  Two forms live here:

* `PcgPlan` / `PartitionedOperator` (csrc/pcg.hip) -- the form meant to scale: rows AND vectors partitioned, the
  pipelined CG recurrence, ONE grouped RCCL all-gather per iteration (the w slices and the dot partials), ghost
  layers so that the nu SpMVs of an operator apply need no exchange in between, the iteration loop in a hipGraph.
  `virtual_pcg_solve` runs the same plans for several virtual ranks in one process on one GPU (shared buffers stand
  in for the all-gather): the partition logic is tested without a multi-GPU box.  `solve_columns_sharded` covers
  the other multi-GPU axis of the path: independent right-hand sides (the 100 one-hot columns of
  `_average_variance`, SLQ probes) are dealt to the ranks and solved with the single-GPU plan -- no data-path
  collective at all, one all-gather of the results.
* `DistCgPlan` (csrc/operator.hip, cg.hip) -- round 1's form: rank p owns the CSR rows [p*n_loc, (p+1)*n_loc),
  every vector is REPLICATED at the global length, each SpMM launch computes the local row slice and one grouped
  all-gather per SpMV assembles the output plus the dot partials; the Chronopoulos-Gear vector updates are
  replicated.  Kept for multi-column solves; it cannot speed up the vector work.

In both, all ranks take bit-identical convergence decisions and issue identical collective sequences.

This module holds the host logic: the partition, padding the graph with isolated nodes so that it
divides evenly, slicing the local operator, creating the RCCL communicator from a unique id
distributed through torch.distributed, and a backend-agnostic reference of the same algorithm
(`distributed_cg_reference`) that the CPU tests run under gloo.
"""
import ctypes
import json
import os
import sys
import time

import torch

from . import _lib
from ._lib import CgParamsT, OperatorT, check, lib, ptr, stream


# ------------------------------------------------------------------------------ partition
class RowPartition:
    """Contiguous, equal row blocks: n_loc = ceil(n / world) rounded up to `align`; the global
    vector length is n_pad = world * n_loc (rows >= n are isolated padding nodes)."""

    def __init__(self, n, world, align=64):
        if n <= 0 or world <= 0:
            raise ValueError("n and world must be positive")
        self.n, self.world = int(n), int(world)
        n_loc = -(-self.n // self.world)
        self.n_loc = -(-n_loc // align) * align
        self.n_pad = self.n_loc * self.world

    def range(self, rank):
        r0 = rank * self.n_loc
        return r0, r0 + self.n_loc

    def owned(self, rank):
        """Real (un-padded) rows of `rank`: [r0, min(r1, n))."""
        r0, r1 = self.range(rank)
        return r0, max(r0, min(r1, self.n))

    def pad(self, v):
        if v.shape[0] == self.n_pad:
            return v
        out = torch.zeros((self.n_pad,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
        out[: self.n] = v
        return out

    def unpad(self, v):
        return v[: self.n]


def pad_graph(graph, n_pad):
    """KnnGraph over n nodes -> the same edges over n_pad >= n nodes (extra nodes isolated)."""
    from .graph import KnnGraph
    if n_pad == graph.n:
        return graph
    extra = n_pad - graph.n
    rowptr = torch.cat([graph.rowptr, graph.rowptr[-1:].expand(extra)])
    # row-order tiles only: a rank's slice must be whole tiles of the global graph (local_csr)
    from .graph import build_tiles
    tiles = build_tiles(n_pad, rowptr, graph.col, graph.nnz) if graph.col.is_cuda else None
    g = KnnGraph(n_pad, graph.tri_row, graph.tri_col, graph.tri_val, rowptr, graph.col, graph.d2, graph.eid, tiles=tiles)
    g.spmv_lanes = graph.spmv_lanes
    return g


def local_csr(lap_data, part, rank):
    """Row slice of the padded CSR of `lap_data` (tensors kept alive by the returned dict)."""
    g = lap_data.graph
    r0, r1 = part.range(rank)
    e0, e1 = int(g.rowptr[r0].item()), int(g.rowptr[r1].item())
    rowptr = (g.rowptr[r0:r1 + 1] - e0).contiguous()
    # an empty slice still needs valid (16-byte aligned) pointers
    col = g.col[e0:e1] if e1 > e0 else torch.zeros(4, dtype=torch.int32, device=g.device)
    vals = lap_data.vals[e0:e1] if e1 > e0 else torch.zeros(4, dtype=torch.float32, device=g.device)
    diag = lap_data.diag[r0:r1].contiguous()
    tiles = None
    gt = getattr(g, "tiles", None)
    if gt is not None and gt.get("rowid") is None and e1 > e0 and r0 % gt["rows"] == 0 and (r1 - r0) % gt["rows"] == 0:
        # the slice's tiles are whole tiles of the global graph: offsets into tile_cols stay absolute
        tiles = dict(gt, tile_ptr=gt["tile_ptr"][r0 // gt["rows"]:r1 // gt["rows"] + 1].contiguous(), lid=gt["lid"][e0:e1])
    return dict(n_loc=part.n_loc, rowptr=rowptr, col=col, vals=vals, diag=diag, e0=e0, e1=e1, ncols=g.n, tiles=tiles)


def local_operator_struct(desc, part, rank):
    """mgp_operator_t for the local rows of a Descriptor built on the padded graph."""
    loc = local_csr(desc.data, part, rank)
    op = desc.struct()
    op.L = _lib.csr_struct(loc["n_loc"], loc["rowptr"], loc["col"], loc["vals"], loc["diag"], loc["ncols"],
                           tiles=loc["tiles"])
    return op, loc


# ------------------------------------------------------------------------------ communicator
_COMM = {}


def init_comm(rank, world):
    """RCCL communicator of libmgp_hip: rank 0 creates the unique id, torch.distributed moves it."""
    import torch.distributed as dist
    key = (rank, world)
    if key in _COMM:
        return _COMM[key]
    nbytes = lib().mgp_dist_unique_id_bytes()
    buf = ctypes.create_string_buffer(nbytes)
    if rank == 0:
        check(lib().mgp_dist_unique_id(buf), "mgp_dist_unique_id")
    payload = [bytes(buf.raw)]
    if world > 1:
        dist.broadcast_object_list(payload, src=0)
    comm = ctypes.c_void_p(0)
    check(lib().mgp_dist_init(rank, world, payload[0], ctypes.byref(comm)), "mgp_dist_init")
    _COMM[key] = comm
    return comm


def comm_info(comm, world, group=None):
    """What RCCL reports about libmgp_hip's communicator on EVERY rank, gathered to all (torch.distributed): dict with
    comm_count / user_rank / device lists indexed by torch rank.  The bench's N > 1 line carries it as `rccl_ranks`: a scale
    record then shows by itself that the data path ran over an RCCL communicator of the size it claims."""
    import torch.distributed as dist
    c, r, d = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    check(lib().mgp_dist_comm_info(comm, ctypes.byref(c), ctypes.byref(r), ctypes.byref(d)), "mgp_dist_comm_info")
    mine = [int(c.value), int(r.value), int(d.value)]
    if world > 1:
        allr = [None] * world
        dist.all_gather_object(allr, mine, group=group)
    else:
        allr = [mine]
    return dict(comm_count=[a[0] for a in allr], user_rank=[a[1] for a in allr], device=[a[2] for a in allr],
                consistent=all(a[0] == world for a in allr) and sorted(a[1] for a in allr) == list(range(world)))


class DistCgPlan:
    """HIP CG over a row-partitioned operator (mgp_cg_plan_create_dist)."""

    def __init__(self, desc, part, rank, comm, C=1, tol=1e-6, max_iter=1000, stop_mode=1, check_every=4):
        self.desc, self.part, self.rank, self.C = desc, part, rank, int(C)
        dev = desc.data.graph.device
        self.op, self._loc = local_operator_struct(desc, part, rank)
        self.params = CgParamsT(float(tol), int(max_iter), 10 if stop_mode == 0 else 0, int(stop_mode),
                                int(check_every), 0, 0)
        wb = lib().mgp_cg_dist_workspace_bytes(ctypes.byref(self.op), self.C, part.world)
        self.work = torch.empty(wb, dtype=torch.uint8, device=dev)
        self.handle = ctypes.c_void_p(0)
        check(lib().mgp_cg_plan_create_dist(ctypes.byref(self.op), self.C, None, ctypes.byref(self.params), comm,
                                            rank, part.world, ptr(self.work), self.work.numel(), stream(),
                                            ctypes.byref(self.handle)), "mgp_cg_plan_create_dist")
        self.iters, self.status, self.resid = 0, 0, None

    def solve(self, B_pad):
        """B_pad: [n_pad, C] replicated right-hand side.  Returns a view of the replicated solution."""
        assert B_pad.shape == (self.part.n_pad, self.C)
        iters, status = ctypes.c_int32(0), ctypes.c_int32(0)
        resid = (ctypes.c_float * self.C)()
        check(lib().mgp_cg_plan_solve(self.handle, ptr(B_pad), None, ctypes.byref(iters), resid,
                                      ctypes.byref(status)), "mgp_cg_plan_solve")
        self.iters, self.status, self.resid = iters.value, status.value, list(resid)
        off = int(lib().mgp_cg_plan_x(self.handle)) - self.work.data_ptr()
        nb = self.part.n_pad * self.C * 4
        return self.work[off:off + nb].view(torch.float32).view(self.part.n_pad, self.C)

    def close(self):
        if self.handle:
            if lib().mgp_cg_plan_poisoned(self.handle):
                _lib.leak(self.__dict__.copy())            # timed-out solve: keep every buffer alive (solvers.CgPlan.close)
            lib().mgp_cg_plan_destroy(self.handle)
            self.handle = ctypes.c_void_p(0)


def apply_partitioned(desc, part, rank, comm, X_pad):
    """Y = A X with partitioned rows (mgp_operator_apply_part); X_pad [n_pad, C] replicated."""
    op, loc = local_operator_struct(desc, part, rank)
    X = _lib.f32c(X_pad)
    Y = torch.empty_like(X)
    work = torch.empty(4 * (X.numel() * 4 + 512) + 1024, dtype=torch.uint8, device=X.device)
    check(lib().mgp_operator_apply_part(ctypes.byref(op), comm, rank, part.world, ptr(X), X.shape[1], ptr(Y),
                                        ptr(work), work.numel(), stream()), "mgp_operator_apply_part")
    return Y


# ------------------------------------------------------------------------------ partitioned pipelined CG
def ghost_layers(graph, r0, r1, layers):
    """Neighbour layers of the row block [r0, r1): layer k = nodes referenced by the rows of (block + layers < k)
    that are not in them yet.  Returns a list of `layers` int64 tensors of global ids (ascending)."""
    n = graph.n
    dev = graph.col.device
    entry_row = getattr(graph, "_entry_row", None)
    if entry_row is None:
        counts = (graph.rowptr[1:] - graph.rowptr[:-1]).long()
        entry_row = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int32), counts)
        graph._entry_row = entry_row
    inc = torch.zeros(n, dtype=torch.bool, device=dev)
    inc[r0:r1] = True
    newest = inc.clone()
    out = []
    for _ in range(layers):
        sel = newest[entry_row.long()]
        cols = graph.col[sel].long()
        ref = torch.zeros(n, dtype=torch.bool, device=dev)
        ref[cols] = True
        newest = ref & ~inc
        out.append(torch.nonzero(newest, as_tuple=False).squeeze(-1))
        inc |= newest
    return out


class PartitionedOperator:
    """One rank's view of a precision-family operator (forms 0 / 2) for the partitioned pipelined CG: the tile view
    of the padded graph over the row order [own rows, ghost layer 1, ..., ghost layer nu - 1, rest], the number of
    rows each launch of the SpMV chain covers, and the mgp_operator_t over it.  `desc.data` must be the LaplacianData
    of the PADDED graph (pad_graph), the same on every rank."""

    def __init__(self, desc, part, rank):
        from .graph import build_tiles
        if desc.form not in (0, 2):
            raise ValueError("the partitioned solver takes single-chain operators (forms 0 and 2)")
        d = desc.data
        g = d.graph
        if g.n != part.n_pad:
            raise ValueError("the operator must live on the padded graph (pad_graph): %d != %d" % (g.n, part.n_pad))
        self.desc, self.part, self.rank = desc, part, rank
        r0, r1 = part.range(rank)
        nu = int(desc.nu)
        if part.world == 1:
            # one rank: every launch covers all rows; the graph's own tile view (row order or its locality order)
            self.tiles, self.vals_t = g.tiles, d.vals_t
            if self.tiles is None:
                raise RuntimeError("the partitioned solver needs the tile view of the graph")
            self.launch_rows = [g.n] * nu
            self.ghosts = [torch.empty(0, dtype=torch.int64, device=g.device) for _ in range(nu - 1)]
        else:
            self.ghosts = ghost_layers(g, r0, r1, nu - 1)
            inc = torch.zeros(g.n, dtype=torch.bool, device=g.device)
            inc[r0:r1] = True
            for gl in self.ghosts:
                inc[gl] = True
            rest = torch.nonzero(~inc, as_tuple=False).squeeze(-1)
            order = torch.cat([torch.arange(r0, r1, device=g.device)] + self.ghosts + [rest]).to(torch.int32)
            self.tiles = build_tiles(g.n, g.rowptr, g.col, g.nnz, order=order)
            if self.tiles is None:
                raise RuntimeError("no tile view for this row order (a tile exceeds the LDS budget)")
            self.vals_t = d.vals.index_select(0, self.tiles["emap"])
            tr = self.tiles["rows"]
            sizes = [part.n_loc]
            for gl in self.ghosts:
                sizes.append(sizes[-1] + int(gl.numel()))
            # launch s produces its output on own rows + (nu - 1 - s) ghost layers
            self.launch_rows = [min(g.n, -(-sizes[nu - 1 - s] // tr) * tr) for s in range(nu)]
        self.op = desc.struct()
        self.op.L = _lib.csr_struct(g.n, g.rowptr, g.col, d.vals, d.diag, tiles=self.tiles, tile_vals=self.vals_t)
        self.ghost_rows = sum(int(gl.numel()) for gl in self.ghosts)


class PcgPlan:
    """Partitioned pipelined CG (mgp_pcg_plan_*): one rank of an RCCL job (`comm`), a single GPU (world 1), or a
    virtual rank sharing `shared` with its siblings (comm None, world > 1; driven by `virtual_pcg_solve`)."""

    def __init__(self, desc, part, rank, comm=None, shared=None, tol=1e-6, max_iter=1000, stop_mode=1, check_every=8,
                 use_graph=True, refine=0, recurrence="pipelined"):
        """recurrence: "pipelined" (one collective per iteration) or "chronopoulos-gear" (two; robust on ill-conditioned
        systems).  refine > 0: the pipelined solve becomes the inner solver of up to `refine` rounds of iterative refinement on
        the TRUE residual (csrc/pcg.hip, "Attainable accuracy"); a refined solve reports status 1 once the true relative
        residual is <= 2 tol (its fp32 evaluation scatters by about that factor around the tolerance; `resid` holds the
        value reached); status 4 = the recurrence stagnated (no refinement)."""
        g0 = desc.data.graph
        if part.world > 1 and getattr(g0, "has_locality_order", lambda: False)():
            # the row partition cuts the node order into contiguous blocks: on an order without locality (the library had to
            # pick a locality order for this graph's tiles) every block's ghost layer is most of the graph.  Correct, slow.
            import warnings
            warnings.warn("partitioned CG over a node order without locality: permute the points into a locality order "
                          "(graph.morton_order / bfs_order) before building the graph, as bench.py does for --s5-order random")
        self.pop = PartitionedOperator(desc, part, rank)
        self.part, self.rank = part, rank
        dev = desc.data.graph.device
        self.params = CgParamsT(float(tol), int(max_iter), 10 if stop_mode == 0 else 0, int(stop_mode), int(check_every),
                                int(bool(use_graph)), int(refine))
        nu = int(desc.nu)
        self._rows = (ctypes.c_int64 * nu)(*self.pop.launch_rows)
        wb = lib().mgp_pcg_workspace_bytes(part.n_pad, part.n_loc, part.world)
        self.work = torch.zeros(wb, dtype=torch.uint8, device=dev)
        self.shared = shared
        self.handle = ctypes.c_void_p(0)
        r0, _ = part.range(rank)
        self.recurrence = {"pipelined": 0, "chronopoulos-gear": 1}[recurrence]
        check(lib().mgp_pcg_plan_create(ctypes.byref(self.pop.op), self._rows, r0, part.n_loc, part.n, comm, rank, part.world,
                                        ptr(shared), self.recurrence, ctypes.byref(self.params), ptr(self.work),
                                        self.work.numel(), stream(), ctypes.byref(self.handle)), "mgp_pcg_plan_create")
        self.iters, self.status, self.resid = 0, 0, None

    @staticmethod
    def shared_buffer(part, device):
        """The buffers virtual ranks share (gathered w and dot partials, double-buffered)."""
        return torch.zeros(int(lib().mgp_pcg_shared_floats(part.n_pad, part.n_loc, part.world)), dtype=torch.float32,
                           device=device)

    def x_view(self):
        off = int(lib().mgp_pcg_plan_x(self.handle)) - self.work.data_ptr()
        return self.work[off:off + self.part.n_pad * 4].view(torch.float32)

    def solve(self, B_pad):
        """B_pad [n_pad] on every rank.  Returns this rank's rows of the solution [n_loc] (a view)."""
        B = _lib.f32c(B_pad.reshape(-1))
        assert B.shape[0] == self.part.n_pad
        iters, status, resid = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_float(0.0)
        check(lib().mgp_pcg_plan_solve(self.handle, ptr(B), None, ctypes.byref(iters), ctypes.byref(resid),
                                       ctypes.byref(status)), "mgp_pcg_plan_solve")
        self.iters, self.status, self.resid = iters.value, status.value, resid.value
        r0, r1 = self.part.range(self.rank)
        return self.x_view()[r0:r1]

    def enqueue(self, phase, par=0, B=None):
        check(lib().mgp_pcg_plan_enqueue(self.handle, int(phase), int(par), ptr(B)), "mgp_pcg_plan_enqueue")

    def poll(self):
        iters, status, resid = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_float(0.0)
        rc = lib().mgp_pcg_plan_poll(self.handle, ctypes.byref(iters), ctypes.byref(resid), ctypes.byref(status))
        if rc == 1:
            return None
        check(rc, "mgp_pcg_plan_poll")
        self.iters, self.status, self.resid = iters.value, status.value, resid.value
        return self.iters, self.status, self.resid

    def close(self):
        if self.handle:
            if lib().mgp_pcg_plan_poisoned(self.handle):
                _lib.leak(self.__dict__.copy())            # timed-out solve: keep every buffer alive (solvers.CgPlan.close)
            lib().mgp_pcg_plan_destroy(self.handle)
            self.handle = ctypes.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def virtual_pcg_solve(desc, part, B_pad, tol=1e-6, max_iter=1000, stop_mode=1, check_every=8, recurrence="pipelined"):
    """The partitioned solve with `part.world` VIRTUAL ranks in this process on one GPU: one plan per rank, all sharing
    the gathered-w / partial buffers (each rank writes its slice: the all-gather is the identity), phases enqueued in
    lock step on the current stream.  Exactly the kernels, row orders, ghost layers and decisions of the RCCL job.
    Returns (x [n_pad] assembled from every rank's own rows, iterations, status, plans' ghost row counts)."""
    dev = B_pad.device
    B = _lib.f32c(B_pad.reshape(-1))
    shared = PcgPlan.shared_buffer(part, dev)
    plans = [PcgPlan(desc, part, r, comm=None, shared=shared, tol=tol, max_iter=max_iter, stop_mode=stop_mode,
                     check_every=check_every, recurrence=recurrence) for r in range(part.world)]
    try:
        for pl in plans:
            pl.enqueue(0, B=B)
        it = 0
        res = None
        while res is None and it <= max_iter + 2 * check_every:
            for _ in range(check_every):
                for pl in plans:
                    pl.enqueue(1, par=it & 1)
                if plans[0].recurrence == 1:           # the update half sits behind the (virtual) gather of the partials
                    for pl in plans:
                        pl.enqueue(2, par=it & 1)
                it += 1
            torch.cuda.synchronize()
            polled = [pl.poll() for pl in plans]
            if all(p is not None for p in polled):
                assert all(p == polled[0] for p in polled), "virtual ranks disagree: %r" % (polled,)
                res = polled[0]
        x = torch.empty(part.n_pad, device=dev)
        for r, pl in enumerate(plans):
            r0, r1 = part.range(r)
            x[r0:r1] = pl.x_view()[r0:r1]
        ghosts = [pl.pop.ghost_rows for pl in plans]
    finally:
        for pl in plans:
            pl.close()
    if res is None:
        raise RuntimeError("virtual ranks undecided after %d iterations" % it)
    return x, res[0], res[1], ghosts


def solve_columns_sharded(desc, B, rank, world, group=None, solver=None, **kw):
    """Independent right-hand sides dealt to the ranks: columns [rank::world] are solved here with the single-GPU
    plan (solvers.cg_solve), the results are all-gathered once.  No collective inside the solves: this is how the
    multi-column workloads of the path (`_average_variance`, SLQ probes) use several GPUs.  B [n, C] replicated; C need
    not be a multiple of `world` (ranks with one column less send a zero column that nobody reads).
    solver: callable (desc, B_cols, **kw) -> (X, iterations, residuals), default solvers.cg_solve (the HIP path; the
    gloo CPU tests pass the oracle's solver to exercise the dealing / gathering on its own)."""
    import torch.distributed as dist
    if solver is None:
        from .solvers import cg_solve as solver
    C = B.shape[1]
    mine = list(range(rank, C, world))
    per = -(-C // world)
    Xl = torch.zeros(B.shape[0], per, device=B.device, dtype=B.dtype)
    its = 0
    if mine:
        X, its, _ = solver(desc, B[:, mine].contiguous(), **kw)
        Xl[:, :len(mine)] = X
    if world == 1:
        return Xl[:, :C], its
    parts = [torch.empty_like(Xl) for _ in range(world)]
    dist.all_gather(parts, Xl, group=group)
    out = torch.empty_like(B)
    for r in range(world):
        cols = list(range(r, C, world))
        out[:, cols] = parts[r][:, :len(cols)]
    return out, its


def distributed_pcg_reference(local_apply, b_loc, part, rank, tol=1e-10, max_iter=1000, group=None):
    """The algorithm of csrc/pcg.hip in plain torch, backend-agnostic (the gloo CPU tests run it with the oracle as
    local operator): partitioned vectors, pipelined recurrence, ONE all-gather per iteration carrying the w slice
    and this rank's two dot partials; every rank derives the same alpha / beta / decision from the gathered sums.
    local_apply(v_full) -> rows [r0, r1) of A v.  b_loc: this rank's rows of b.  Returns (x_loc, iterations)."""
    import torch.distributed as dist
    n_loc = part.n_loc

    def gather(w_loc, g, d):
        send = torch.cat([w_loc, torch.stack([g, d]).to(w_loc)])
        if part.world == 1:
            recv = send.unsqueeze(0)
        else:
            buf = [torch.empty_like(send) for _ in range(part.world)]
            dist.all_gather(buf, send, group=group)
            recv = torch.stack(buf)
        return recv[:, :n_loc].reshape(-1), recv[:, n_loc].sum(), recv[:, n_loc + 1].sum()

    r = b_loc.clone()
    x, p, s, z = (torch.zeros_like(r) for _ in range(4))
    b_full, _, _ = gather(r, r.new_zeros(()), r.new_zeros(()))
    w = local_apply(b_full)
    w_full, gamma, delta = gather(w, (r * r).sum(), (w * r).sum())
    bb = gamma
    gamma_old = alpha_old = None
    it = 0
    while it < max_iter:
        rel = torch.sqrt(gamma / bb) if bb > 0 else gamma * 0
        if bool(rel <= tol):
            break
        q = local_apply(w_full)
        if it == 0:
            beta = gamma * 0
            alpha = gamma / delta
        else:
            beta = gamma / gamma_old
            alpha = gamma / (delta - beta * gamma / alpha_old)
        z = q + beta * z
        s = w + beta * s
        p = r + beta * p
        x = x + alpha * p
        r = r - alpha * s
        w = w - alpha * z
        gamma_old, alpha_old = gamma, alpha
        w_full, gamma, delta = gather(w, (r * r).sum(), (w * r).sum())
        it += 1
    return x, it


# ------------------------------------------------------------------------------ reference (gloo tests)
def distributed_cg_reference(local_matvec, b_pad, part, rank, tol=1e-10, max_iter=1000, group=None):
    """The algorithm of csrc/cg.hip + operator.hip in plain torch, backend-agnostic: replicated
    vectors, row-partitioned operator (`local_matvec(u_full) -> rows [r0, r1)`), one all-gather per
    operator apply, Chronopoulos-Gear recurrence, per-column relative-residual stop.
    Used by the world_size-2 gloo CPU tests with the oracle as local operator."""
    import torch.distributed as dist

    def A(u):
        loc = local_matvec(u).contiguous()
        out = torch.empty_like(u)
        if part.world > 1:
            dist.all_gather_into_tensor(out, loc, group=group)
        else:
            out.copy_(loc)
        return out

    x = torch.zeros_like(b_pad)
    r = b_pad.clone()
    u = r.clone()
    w = A(u)
    p = torch.zeros_like(b_pad)
    s = torch.zeros_like(b_pad)
    bb = (r * r).sum(0)
    gamma_old = alpha_old = None
    it = 0
    for it in range(1, max_iter + 1):
        gamma = (r * u).sum(0)
        delta = (w * u).sum(0)
        rel = torch.sqrt((r * r).sum(0) / bb.clamp_min(1e-300))
        if bool((rel <= tol).all()):
            break
        if it == 1:
            beta = torch.zeros_like(gamma)
            alpha = gamma / delta
        else:
            beta = gamma / gamma_old
            alpha = gamma / (delta - beta * gamma / alpha_old)
        frozen = rel <= tol
        alpha = torch.where(frozen, torch.zeros_like(alpha), alpha)
        beta = torch.where(frozen, torch.zeros_like(beta), beta)
        p = u + beta * p
        s = w + beta * s
        x = x + alpha * p
        r = r - alpha * s
        u = r.clone()
        w = A(u)
        gamma_old, alpha_old = gamma, alpha
    return x, it - 1


# ------------------------------------------------------------------------------ bench (N > 1)
def bench_distributed(args, dev, rank, world, build_workload, spmm_bytes, hbm_peak, emit=print, cpu_baseline=None):
    """bench.py body for world > 1 (also reachable at world == 1 with MGP_FORCE_DIST=1).

    --scaling strong (default; BASELINE.json's metric is the N = 60k graph on 1/2/4/8 GPUs): the SAME graph on every
    world size, rows and vectors partitioned, partitioned pipelined CG (PcgPlan: one grouped RCCL all-gather per
    iteration).  --scaling weak: `world` x the per-GPU node count, round 1's replicated-vector plan (DistCgPlan).
    Setup (untimed): k-NN queries sharded by rows, lists all-gathered, graph + Laplacian built on every rank."""
    import torch.distributed as dist

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    strong = args.scaling == "strong"
    wl = build_workload(args, dev, rank, world, shard_knn=True, scale_nodes=not strong)
    g = wl["graph"]
    part = RowPartition(g.n, world)
    comm = init_comm(rank, world)
    from .graph import LaplacianData
    gp = pad_graph(g, part.n_pad)
    lap = wl["lap"]
    data = LaplacianData(gp, lap.data.eps, lap.data.self_loops)
    base = wl["desc"]
    pre = data.dsqrt if base.pre is not None else None
    desc = base.with_(data=data, pre=pre, post=pre)
    y = part.pad(wl["y"].view(-1, 1)).contiguous()
    if strong:
        # short well-conditioned solves (C3: 3 iterations): small chunks, no refinement; the ill-conditioned 1M-node
        # system: chunks of 32 iterations with residual replacement, refinement rounds on the true residual
        long_solve = args.workload != "c3"
        plan = PcgPlan(desc, part, rank, comm=comm, tol=args.tol, max_iter=4000, stop_mode=1,
                       check_every=16 if long_solve else 4, refine=3 if long_solve else 0,
                       recurrence="chronopoulos-gear" if long_solve else "pipelined")
        solve = lambda: plan.solve(y.view(-1))          # noqa: E731
        ghost = plan.pop.ghost_rows
    else:
        plan = DistCgPlan(desc, part, rank, comm, C=1, tol=args.tol, max_iter=2000, stop_mode=1)
        solve = lambda: plan.solve(y)                   # noqa: E731
        ghost = 0
    import gc
    gc.collect()
    gc.disable()          # a generation-2 collection (tens of ms) otherwise lands inside a long timed loop
    for _ in range(args.warmup):
        out = solve()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        out = solve()
        iters += plan.iters
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    tmax = torch.tensor([dt], device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    its = iters / args.steps
    B = spmm_bytes(g.n, g.M)                   # whole (all ranks) operator, one SpMV
    spmvs = (its + 1) * wl["nu"]               # pipelined: the start apply + one per iteration; classic: its + 1 as well
    value = B * spmvs * args.steps / dt / 1e9
    # true residual of the assembled solution (one more partitioned apply; every rank holds the gathered solution)
    if strong:
        xg = torch.zeros(part.n_pad, device=dev)
        r0, r1 = part.range(rank)
        xg[r0:r1] = out
        if world > 1:
            dist.all_reduce(xg)
        xg = xg.view(-1, 1)
    else:
        xg = out.contiguous()
    r = apply_partitioned(desc, part, rank, comm, xg) - y
    true_res = float(r.norm() / y.norm())
    # ---- the workload of this graph that CAN use several GPUs: the 100 one-hot right-hand sides of `_average_variance`
    # (precision_matern_operator.py:45-53), columns dealt to the ranks, every rank solving its share with the single-GPU
    # plan on the whole (replicated) graph, one all-gather of the solutions at the end -- no collective inside the solves
    sharded = None
    if args.workload == "c3" and not getattr(args, "no_extras", False):
        columns, tol_mr = 100, 1e-2
        desc_q = wl["desc"].with_(scale=1.0, form=0, noise=0.0)
        torch.manual_seed(1337)
        idxc = torch.randint(0, g.n - 1, (1, columns), device=dev)
        Bm = torch.zeros(g.n, columns, device=dev).scatter_(0, idxc, 1.0)
        for _ in range(2):
            Xs, its_s = solve_columns_sharded(desc_q, Bm, rank, world, tol=tol_mr, max_iter=1000, stop_mode=0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        reps = 3
        for _ in range(reps):
            Xs, its_s = solve_columns_sharded(desc_q, Bm, rank, world, tol=tol_mr, max_iter=1000, stop_mode=0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ts = torch.tensor([(time.perf_counter() - t0s) / reps], device=dev)
        if world > 1:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        rs = desc_q.apply(Xs) - Bm
        sharded = dict(columns=columns, columns_per_rank=-(-columns // world), solve_ms=round(float(ts.item()) * 1e3, 3),
                       iterations_rank0=its_s, stop="linear_cg rule, tol %g" % tol_mr,
                       true_mean_rel_residual=float((rs.norm(dim=0) / Bm.norm(dim=0)).mean()),
                       average_variance=float((Xs * Bm).sum() / columns),
                       how="columns [rank::world] solved with the single-GPU plan on the replicated graph, one all-gather of "
                           "the solutions; time = max over ranks, barriers on both sides")
    rccl = comm_info(comm, world)          # (collective: every rank takes part)
    # ---- the same solve with the single-GPU plan (cg.hip: one hipGraph per solve) on this rank's replica of the graph, in this
    # process: what the partitioned plan's chunked loop + collectives cost against it is then visible in the record itself
    from .solvers import CgPlan
    single = None
    if strong:
        refine1 = args.refine if getattr(args, "refine", -1) >= 0 else (0 if args.workload == "c3" else 3)
        sp = CgPlan(wl["desc"], 1, tol=args.tol, max_iter=5000, stop_mode=1, check_every=8, refine=refine1)
        y1 = wl["y"].view(-1, 1).contiguous()
        for _ in range(max(args.warmup, 2)):
            sp.solve(y1, copy=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sp.solve(y1, copy=False)
        torch.cuda.synchronize()
        single = dict(ms_per_step=round((time.perf_counter() - t1) / args.steps * 1e3, 4), iterations=sp.iters,
                      rel_residual=float(max(sp.resid)), refine=refine1,
                      note="rank 0, same process, same graph: solvers.CgPlan (what `bench.py --gpus 1` times)")
        sp.close()
    cb = None
    if rank == 0 and cpu_baseline is not None:
        cb, _ = cpu_baseline(wl, its)          # the N = 1 line's port (oracle/ref_torch.py), rank 0's host cores
    if rank == 0:
        how = ("rows AND vectors partitioned over %d ranks, %s, %d ghost rows on rank 0"
               % (world, "Chronopoulos-Gear recurrence: two RCCL collectives per iteration (gathered vector + gamma partials; "
                  "one delta per rank), refinement on the true residual" if long_solve else
                  "pipelined CG: one grouped RCCL all-gather per iteration (w slices + dot partials)", ghost)) if strong else \
              ("rows of L partitioned over %d ranks, vectors replicated, one grouped RCCL all-gather per SpMV" % world)
        line = dict(metric="CG-solve wall-time + SpMV HBM GB/s, N=60k RMNIST graph", value=round(value, 2),
                    unit="GB/s (algorithmic SpMV bytes inside the CG solve, all ranks)", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4),
                    higher_is_better=True, scaling="strong" if strong else "weak", vs_baseline=None, dtype="f32",
                    data="synthetic",
                    config=dict(workload=wl["name"], nodes=g.n, nodes_per_gpu=part.n_loc, edges=g.M,
                                rhs_columns=1, parallelism=how,
                                cg_tol=args.tol, cg_iters=its, cg_rel_residual=float(plan.resid if strong else max(plan.resid)),
                                cg_true_residual_fp32_apply=true_res, spmv_per_solve=spmvs, eps=wl["eps"],
                                expectation="N = 60k is latency-bound: one GPU runs an iteration in ~16 us, an RCCL "
                                            "all-gather over xGMI costs about as much by itself, so more GPUs are expected to be "
                                            "SLOWER on this graph (SURVEY.md section 7); the path that scales is the 1M-node "
                                            "workload (--workload s5) and the column-sharded multi-rhs solves"),
                    cg_solve_ms=round(dt / args.steps * 1e3, 4),
                    roofline=dict(bound="hbm", achieved=round(value / world, 1), peak=hbm_peak, unit="GB/s",
                                  frac=round(value / world / hbm_peak, 4), traffic=None,
                                  note="per-GPU share of the whole-job rate (includes collectives and vector "
                                       "kernels); the kernel-only figure is the N=1 line"))
        line["rccl_ranks"] = rccl
        if single is not None:
            line["ms_per_step_single_gpu_plan"] = single["ms_per_step"]
            line["single_gpu_plan"] = single
        if cb is not None:
            line["cpu_baseline"] = cb
        if sharded is not None:
            line["cg_multi_rhs_sharded"] = sharded
        line["unmeasured_note"] = ("the partitioned solver has run with more than one RCCL rank only in the driver's scale "
                                   "record; `--scaling weak` (round 1's replicated-vector plan) is the fallback form")
        emit(json.dumps(line))
    plan.close()
    if world > 1:
        dist.barrier()
