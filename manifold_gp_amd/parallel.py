"""Row-partitioned multi-GPU CG (one process per GPU, RCCL over xGMI).

The reference has no distributed code.  Design (csrc/operator.hip, cg.hip):
rank p owns the CSR rows [p*n_loc, (p+1)*n_loc) of the Laplacian, every vector is replicated at the
global (padded) length, each SpMM launch computes the local row slice and ONE grouped RCCL
all-gather assembles the output (plus, on the last launch of an operator chain, the dot-product
partials) on every rank.  The vector updates of the Chronopoulos-Gear recurrence are replicated, so
all ranks take bit-identical convergence decisions and issue identical collective sequences.

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
def bench_distributed(args, dev, rank, world, build_workload, spmm_bytes, hbm_peak, emit=print):
    """bench.py body for world > 1 (also reachable at world == 1 with MGP_FORCE_DIST=1): weak scaling,
    `world` x 60 000 points; k-NN queries sharded by rows, lists all-gathered, graph + Laplacian
    built redundantly per rank (setup, untimed), rows of the operator partitioned for the CG."""
    import torch.distributed as dist

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    wl = build_workload(args, dev, rank, world, shard_knn=True)
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
    plan = DistCgPlan(desc, part, rank, comm, C=1, tol=args.tol, max_iter=2000, stop_mode=1)
    y = part.pad(wl["y"].view(-1, 1)).contiguous()
    import gc
    gc.collect()
    gc.disable()          # a generation-2 collection (tens of ms) otherwise lands inside a long timed loop
    for _ in range(args.warmup):
        out = plan.solve(y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 0
    for _ in range(args.steps):
        out = plan.solve(y)
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
    spmvs = (its + 1) * wl["nu"]
    value = B * spmvs * args.steps / dt / 1e9
    # true residual with one partitioned apply
    r = apply_partitioned(desc, part, rank, comm, out.contiguous()) - y
    true_res = float(r.norm() / y.norm())
    if rank == 0:
        line = dict(metric="CG-solve wall-time + SpMV HBM GB/s, N=60k RMNIST graph", value=round(value, 2),
                    unit="GB/s (algorithmic SpMV bytes inside the CG solve, all ranks)", n_gpus=world,
                    steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4),
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=wl["name"], nodes=g.n, nodes_per_gpu=part.n_loc, edges=g.M,
                                rhs_columns=1, parallelism="rows of L partitioned over %d ranks, vectors replicated, "
                                "one grouped RCCL all-gather per SpMV" % world,
                                cg_tol=args.tol, cg_iters=its, cg_rel_residual=max(plan.resid),
                                cg_true_residual_fp32_apply=true_res, spmv_per_solve=spmvs, eps=wl["eps"]),
                    cg_solve_ms=round(dt / args.steps * 1e3, 4),
                    roofline=dict(bound="hbm", achieved=round(value / world, 1), peak=hbm_peak, unit="GB/s",
                                  frac=round(value / world / hbm_peak, 4), traffic=None,
                                  note="per-GPU share of the whole-job rate (includes collectives and vector "
                                       "kernels); the kernel-only figure is the N=1 line"))
        emit(json.dumps(line))
    plan.close()
    if world > 1:
        dist.barrier()
