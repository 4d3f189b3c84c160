"""Device-resident graph containers and their HIP builders (thin host glue over the C-ABI).

KnnGraph       symmetrised k-NN graph: the reference's COO view (idx[2,M], val[M]) plus the padded
               full-symmetric CSR the kernels sweep (manifold_gp/utils/nearest_neighbors.py:39-55).
LaplacianData  everything graph_laplacian_operator.py:52-106 caches, computed by three fused
               row passes for one (eps, self_loops).
"""
import ctypes

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


TILE_ROWS = 64             # rows per SpMV tile (workgroup of 256 lanes); 32 / 64 / 128 are supported


REORDER_BELOW_REUSE = 3.0  # entries per distinct tile column under which a locality order is tried


def bfs_order(n, rowptr, col):
    """Breadth-first locality order of the graph (mgp_graph_bfs_order): int32 [n] permutation."""
    order = torch.empty(n, dtype=torch.int32, device=col.device)
    wb = lib().mgp_graph_bfs_workspace_bytes(n)
    work = _lib.workspace(wb, "graph", col.device)
    check(lib().mgp_graph_bfs_order(n, ptr(rowptr), ptr(col), ptr(order), ptr(work), work.numel(), stream()),
          "mgp_graph_bfs_order")
    return order


def chain_order(n, rowptr, col, d2):
    """Nearest-neighbour chain order of the graph (mgp_graph_chain_order: a host walk over a copy of the CSR): int32 [n] permutation."""
    order = torch.empty(n, dtype=torch.int32, device=col.device)
    check(lib().mgp_graph_chain_order(n, ptr(rowptr), ptr(col), ptr(d2), ptr(order), stream()), "mgp_graph_chain_order")
    return order


def build_tiles(n, rowptr, col, nnz, tile_rows=None, order=None):
    """Row-tile column dictionaries for the C == 1 SpMV (mgp_graph_tiles).  Returns the dict that
    _lib.csr_struct takes, or None when the graph has no entries / a tile does not fit the LDS budget.
    order: optional int32 [n] row permutation (tiles over that order; adds tile_rowptr / emap / rowid)."""
    if nnz <= 0:
        return None
    dev = col.device
    for rows in ([int(tile_rows)] if tile_rows else [TILE_ROWS, 32]):
        ntiles = -(-n // rows)
        tile_ptr = torch.empty(ntiles + 1, dtype=torch.int32, device=dev)
        tile_cols = torch.empty(nnz, dtype=torch.int32, device=dev)
        lid = torch.empty(nnz, dtype=torch.int16, device=dev)
        tile_rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev) if order is not None else None
        emap = torch.empty(nnz, dtype=torch.int32, device=dev) if order is not None else None
        wb = lib().mgp_graph_tiles_workspace_bytes(n, nnz)
        work = _lib.workspace(wb, "graph", dev)
        total, mc, me = ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
        rc = lib().mgp_graph_tiles(n, ptr(rowptr), ptr(col), nnz, rows, ptr(order), ptr(tile_rowptr), ptr(emap),
                                   ptr(tile_ptr), ptr(tile_cols), ptr(lid), ctypes.byref(total), ctypes.byref(mc),
                                   ctypes.byref(me), ptr(work), work.numel(), stream())
        if rc == -3:                      # MGP_ERR_UNSUPPORTED: a tile references > 65536 columns
            continue
        check(rc, "mgp_graph_tiles")
        if (mc.value + me.value // 4) * 4 > 65536 - 64:
            continue
        t = dict(tile_ptr=tile_ptr, tile_cols=tile_cols[:total.value].clone(), lid=lid, rows=rows,
                 max_cols=mc.value, max_entries=me.value, total_cols=total.value, reuse=nnz / max(total.value, 1))
        if order is not None:
            t.update(tile_rowptr=tile_rowptr, emap=emap.long(), rowid=order)
        return t
    return None


WIDE_CHAIN = [True]         # wide products on the chain-relabelled matrix where it pays (KnnGraph.wide_relabelled)
WIDE_CHAIN_MIN_NODES = 4096
WIDE_CHAIN_MAX_NODES = 400000
WIDE_CHAIN_GAIN = 1.25
MT_MIN_NODES = 4096         # below this a wide SpMM is a few microseconds whatever the kernel
MT_MIN_FILL = 0.125         # non-zero share of the dense tiles below which the gather kernels are taken instead
MT_ENABLED = [True]


class MtPlan:
    """Dense 16-row tiles of one CSR (natural row order) for the matrix-core SpMM at 48 <= C <= 256 (csrc/spmm.hip
    spmm_mt_kernel): sptr [T + 1] steps before tile t, dcol the tiles' distinct columns padded to whole blocks of 16, img the
    values in MFMA operand order (mgp_spmm_mt_fill).  `build` returns None when the graph is small, the tiles would be less
    than MT_MIN_FILL full (no locality in the row order: 16 rows then name ~850 distinct columns) or the image would not fit a
    32-bit byte offset."""

    def __init__(self, sptr, dcol, img, tiles, steps, fill):
        self.sptr, self.dcol, self.img, self.tiles, self.steps, self.fill = sptr, dcol, img, tiles, steps, fill

    @staticmethod
    def structure(graph):
        """The part that depends on the sparsity pattern only, cached on the graph object (a new bandwidth refills the image,
        nothing else): 16-row tile dictionaries, step offsets, fill.  None when the image does not pay."""
        if not hasattr(graph, "_mt_structure"):
            graph._mt_structure = None
            n, nnz = graph.n, graph.nnz
            t = build_tiles(n, graph.rowptr, graph.col, nnz, tile_rows=16) if (MT_ENABLED[0] and n >= MT_MIN_NODES and nnz > 0) else None
            if t is not None:
                D = (t["tile_ptr"][1:] - t["tile_ptr"][:-1]).long()
                S = (D + 63) // 64 * 16                                    # steps per tile: whole bodies of four blocks of four
                sptr = torch.zeros(D.numel() + 1, dtype=torch.int64, device=graph.col.device)
                torch.cumsum(S, 0, out=sptr[1:])
                steps = int(sptr[-1])
                fill = 2 * graph.M / max(1, 64 * steps)                    # off-diagonal entries / cells of the dense tiles
                if steps > 0 and fill >= MT_MIN_FILL and (steps + 32) * 256 < 2 ** 31:
                    graph._mt_structure = dict(tiles=t, sptr=sptr.to(torch.int32), steps=steps, fill=fill, ntiles=int(D.numel()))
        return graph._mt_structure

    @classmethod
    def build(cls, graph, vals):
        st = cls.structure(graph)
        if st is None:
            return None
        t, steps = st["tiles"], st["steps"]
        dcol = torch.empty(4 * steps + 192, dtype=torch.int32, device=vals.device)
        img = torch.empty(64 * (steps + 32), dtype=torch.float32, device=vals.device)
        check(lib().mgp_spmm_mt_fill(graph.n, ptr(graph.rowptr), ptr(vals), ptr(t["lid"]), ptr(t["tile_ptr"]), ptr(t["tile_cols"]),
                                     ptr(st["sptr"]), steps, ptr(dcol), ptr(img), stream()), "mgp_spmm_mt_fill")
        return cls(st["sptr"], dcol, img, st["ntiles"], steps, st["fill"])


def morton_order(x):
    """Z-curve order of points with d <= 3 (mgp_morton_order): int32 [n] permutation."""
    n, d = x.shape
    order = torch.empty(n, dtype=torch.int32, device=x.device)
    wb = lib().mgp_morton_order_workspace_bytes(n)
    work = _lib.workspace(wb, "graph", x.device)
    check(lib().mgp_morton_order(ptr(_lib.f32c(x)), n, d, ptr(order), ptr(work), work.numel(), stream()),
          "mgp_morton_order")
    return order


def build_tiles_auto(n, rowptr, col, nnz, points=None):
    """Tiles in the given row order; when that order carries no locality (few entries per distinct tile
    column) a locality order is tried and kept if it helps: the Z-curve of the points when they have
    d <= 3 coordinates, else a breadth-first order of the graph."""
    t = build_tiles(n, rowptr, col, nnz)
    if t is None or t["reuse"] >= REORDER_BELOW_REUSE or n < 4 * TILE_ROWS:
        return t
    if points is not None and points.dim() == 2 and points.shape[1] <= 3 and points.shape[0] == n:
        order = morton_order(points)
    else:
        order = bfs_order(n, rowptr, col)
    t2 = build_tiles(n, rowptr, col, nnz, order=order)
    return t2 if (t2 is not None and t2["reuse"] > 1.3 * t["reuse"]) else t


class KnnGraph:
    _next_uid = [1]

    def __init__(self, n, tri_row, tri_col, tri_val, rowptr, col, d2, eid, tiles="auto", points=None):
        self.uid = KnnGraph._next_uid[0]          # never reused, unlike id(): what the solvers' plan cache keys on
        KnnGraph._next_uid[0] += 1
        self.n = int(n)
        self.tri_row, self.tri_col, self.tri_val = tri_row, tri_col, tri_val
        self.rowptr, self.col, self.d2, self.eid = rowptr, col, d2, eid
        self.M = int(tri_val.shape[0])
        self.nnz = int(col.shape[0])
        self._edge_index = None
        if isinstance(tiles, str):          # "auto": build on the device the CSR lives on (host tensors: none)
            tiles = build_tiles_auto(self.n, rowptr, col, self.nnz, points) if col.is_cuda else None
        self.tiles = tiles
        # sub-wave group width of the fallback (gather) C == 1 SpMV: 4 entries per lane per pass; 8 lanes
        # measured best for mean rows of ~60 entries (tools/tune_spmv.py), wider groups only for longer rows
        mean_row = self.nnz / max(self.n, 1)
        lanes = 8
        while lanes < 64 and lanes * 16 < mean_row:
            lanes *= 2
        self.spmv_lanes = lanes

    @property
    def device(self):
        return self.tri_val.device

    def csr_with(self, vals, diag, tile_vals=None, mt=None):
        """mgp_csr_t over this graph's structure with the given entry values / diagonal (tile_vals: the
        values in tile order when the tiles follow a row order, see tile_values)."""
        return _lib.csr_struct(self.n, self.rowptr, self.col, vals, diag, tiles=self.tiles, tile_vals=tile_vals, mt=mt)

    def tile_values(self, vals):
        """`vals` gathered into tile order (None when the tiles are in row order)."""
        if self.tiles is None or self.tiles.get("rowid") is None:
            return None
        return vals.index_select(0, self.tiles["emap"])

    def has_locality_order(self):
        t = self.tiles
        return t is not None and t.get("rowid") is not None and t.get("emap") is not None

    def relabelled(self):
        """RelabelledGraph of this graph (cached); only for graphs whose tiles follow a locality order."""
        if getattr(self, "_relabelled", None) is None:
            self._relabelled = RelabelledGraph(self)
        return self._relabelled

    def wide_relabelled(self):
        """RelabelledGraph over the nearest-neighbour CHAIN order, for the wide products only (the eigensolver's blocks, the
        100-column solves: csrc/spmm.hip spmm_mt_kernel, whose work is proportional to the distinct columns of a 16-row tile),
        or None when it does not pay.  Tried between WIDE_CHAIN_MIN_NODES and WIDE_CHAIN_MAX_NODES nodes (the walk is sequential,
        on the host); kept when the dense 16-row tiles shrink by WIDE_CHAIN_GAIN or more against what the wide products would
        run on otherwise (the given order, or the graph's own breadth-first / Z-curve locality order).  The C = 1 / small-C paths keep the caller's order (a
        relabelled solve permutes its right-hand side in and its solution out: two launches that a 55 us solve cannot afford,
        a 3 ms one can).  Cached."""
        if not hasattr(self, "_wide_relabelled"):
            self._wide_relabelled = None
            ordered = self.has_locality_order()
            if (WIDE_CHAIN[0] and self.col.is_cuda and self.nnz > 0 and WIDE_CHAIN_MIN_NODES <= self.n <= WIDE_CHAIN_MAX_NODES):
                # what the wide products would run on otherwise: the graph as given, or relabelled by its own locality order
                # (breadth-first / Z-curve: graph.build_tiles_auto)
                base = MtPlan.structure(self.relabelled() if ordered else self)
                order = chain_order(self.n, self.rowptr, self.col, self.d2)
                t64 = build_tiles(self.n, self.rowptr, self.col, self.nnz, tile_rows=TILE_ROWS, order=order)
                if t64 is not None:
                    rg = RelabelledGraph(self, tiles=t64)
                    rg.d2 = None
                    st = MtPlan.structure(rg)
                    if st is not None and (base is None or st["steps"] * WIDE_CHAIN_GAIN <= base["steps"]):
                        rg.emap = t64["emap"]
                        self._wide_relabelled = rg
        return self._wide_relabelled

    @property
    def edge_index(self):
        """idx[2, M] int64, row<col, sorted -- what NearestNeighbors.graph returns."""
        if self._edge_index is None:
            self._edge_index = torch.stack([self.tri_row, self.tri_col]).long()
        return self._edge_index

    @property
    def edge_value(self):
        return self.tri_val

    # ------------------------------------------------------------------ builders
    @classmethod
    def from_knn(cls, D, I, tiles="auto", points=None):
        """(D[n,k] f32, I[n,k] i32) on device -> KnnGraph via mgp_graph_build."""
        _lib.require_device(D, I)
        n, k = I.shape
        dev = D.device
        D = _lib.f32c(D)
        I = I.to(torch.int32).contiguous()
        cap_e = n * (k - 1)
        cap_z = 2 * cap_e + 4 * n
        tri_row = torch.empty(cap_e, dtype=torch.int32, device=dev)
        tri_col = torch.empty(cap_e, dtype=torch.int32, device=dev)
        tri_val = torch.empty(cap_e, dtype=torch.float32, device=dev)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(cap_z, dtype=torch.int32, device=dev)
        d2 = torch.empty(cap_z, dtype=torch.float32, device=dev)
        eid = torch.empty(cap_z, dtype=torch.int32, device=dev)
        wb = lib().mgp_graph_workspace_bytes(n, k)
        work = _lib.workspace(wb, "graph", dev)
        M, nnz = ctypes.c_int64(0), ctypes.c_int64(0)
        check(lib().mgp_graph_build(ptr(D), ptr(I), n, k, ptr(tri_row), ptr(tri_col), ptr(tri_val),
                                    ctypes.byref(M), ptr(rowptr), ptr(col), ptr(d2), ptr(eid),
                                    ctypes.byref(nnz), ptr(work), work.numel(), stream()), "mgp_graph_build")
        M, nnz = M.value, nnz.value
        return cls(n, tri_row[:M].clone(), tri_col[:M].clone(), tri_val[:M].clone(), rowptr,
                   col[:nnz].clone(), d2[:nnz].clone(), eid[:nnz].clone(), tiles=tiles, points=points)

    @classmethod
    def from_coo(cls, idx, val, n, tiles="auto"):
        """Reference-style edge list (idx[2,M] any int dtype, val[M]) -> KnnGraph."""
        _lib.require_device(idx, val)
        dev = val.device
        M = int(val.shape[0])
        tri_row = idx[0].to(torch.int32).contiguous()
        tri_col = idx[1].to(torch.int32).contiguous()
        tri_val = _lib.f32c(val.reshape(-1))
        cap_z = 2 * M + 4 * n
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(cap_z, dtype=torch.int32, device=dev)
        d2 = torch.empty(cap_z, dtype=torch.float32, device=dev)
        eid = torch.empty(cap_z, dtype=torch.int32, device=dev)
        wb = lib().mgp_graph_coo_workspace_bytes(n, M)
        work = _lib.workspace(wb, "graph", dev)
        nnz = ctypes.c_int64(0)
        check(lib().mgp_graph_from_coo(ptr(tri_row), ptr(tri_col), ptr(tri_val), M, n, ptr(rowptr),
                                       ptr(col), ptr(d2), ptr(eid), ctypes.byref(nnz), ptr(work), work.numel(),
                                       stream()), "mgp_graph_from_coo")
        nnz = nnz.value
        g = cls(n, tri_row, tri_col, tri_val, rowptr, col[:nnz].clone(), d2[:nnz].clone(), eid[:nnz].clone(),
                tiles=tiles)
        if idx.dtype == torch.int64:
            g._edge_index = idx
        return g


class RelabelledGraph:
    """The structure of P A P^T for a KnnGraph whose tiles follow a locality order (tiles["rowid"]): row p of this graph is
    node order[p] of the caller's.  The CSR in that order already exists as the tile view (tile_rowptr, values gathered
    through emap); what is new is only the column ids and the dictionaries' ids mapped to the new labels -- so that the
    VECTORS of an iteration can live in this order too and every kernel (tile SpMV, vector updates, dictionaries) streams
    instead of gathering / scattering through `rowid` at 4-byte granularity.  Built once per graph, on demand."""

    def __init__(self, g, tiles=None):
        t = g.tiles if tiles is None else tiles         # (tiles: an ordered 64-row tile view of g over another order, see wide_relabelled)
        dev = g.device
        self.n, self.nnz, self.M, self.spmv_lanes = g.n, g.nnz, g.M, g.spmv_lanes
        self.device = dev
        self.order = t["rowid"].long()                                  # new position -> caller's node id
        self.inv = torch.empty_like(self.order)
        self.inv[self.order] = torch.arange(g.n, device=dev)            # caller's node id -> new position
        self.order32, self.inv32 = self.order.to(torch.int32), self.inv.to(torch.int32)
        self.rowptr = t["tile_rowptr"]
        self.col = self.inv.index_select(0, g.col.long().index_select(0, t["emap"])).to(torch.int32)
        self.tiles = dict(tile_ptr=t["tile_ptr"], tile_cols=self.inv.index_select(0, t["tile_cols"].long()).to(torch.int32),
                          lid=t["lid"], rows=t["rows"], max_cols=t["max_cols"], max_entries=t["max_entries"],
                          reuse=t.get("reuse"))

    @staticmethod
    def _rows(v, idx32, idx64, out=None):
        # float32 [n, C] blocks on the device (right-hand sides, solutions): mgp_permute_rows, a float4 of a row per lane;
        # anything else (node vectors, float64 solutions, index tensors): torch
        if v.dtype == torch.float32 and v.dim() == 2 and v.is_cuda and v.is_contiguous() and v.shape[0] == idx32.shape[0] \
                and (out is None or (out.dtype == torch.float32 and out.is_contiguous() and out.shape == v.shape)):
            dst = torch.empty_like(v) if out is None else out
            check(lib().mgp_permute_rows(ptr(v), ptr(idx32), v.shape[0], v.shape[1], ptr(dst), stream()), "mgp_permute_rows")
            return dst
        return torch.index_select(v, 0, idx64, out=out)

    def permute(self, v):
        """caller's order -> this order, along dim 0"""
        return self._rows(v, self.order32, self.order)

    def unpermute(self, v, out=None):
        """this order -> caller's order, along dim 0"""
        return self._rows(v, self.inv32, self.inv, out)


class RelabelledData:
    """LaplacianData of the relabelled graph (same eps): values = the tile-order copy the data already holds, node vectors
    permuted once.  Quacks like LaplacianData for Descriptor / the solvers."""

    def __init__(self, data, rg=None):
        """rg: None = the graph's own locality order (its tile view); else a RelabelledGraph over another order of the same graph
        (KnnGraph.wide_relabelled: rg.emap maps its entries to the caller-order CSR's)."""
        own = rg is None
        rg = data.graph.relabelled() if own else rg
        self.graph = rg
        self.uid = LaplacianData._next_uid[0]
        LaplacianData._next_uid[0] += 1
        self.eps, self.self_loops = data.eps, data.self_loops
        self.vals = data.vals_t if own else data.vals.index_select(0, rg.emap)
        for name in ("degree_unnorm", "degree", "diag", "dsqrt", "dinvsqrt"):
            setattr(self, name, rg.permute(getattr(data, name)).contiguous())
        self.vals_t = None
        self._perm_cache = {}
        self._source = data                     # (keeps vals_t alive)

    def csr(self, wide=False):
        """wide: the caller is about to multiply 48 columns or more -- build the matrix-core tile image if there is none
        yet (once built it rides in every struct)."""
        g = self.graph
        return _lib.csr_struct(g.n, g.rowptr, g.col, self.vals, self.diag, tiles=g.tiles, mt=self.mt_plan(wide))

    def mt_plan(self, build=True):
        """MtPlan of this CSR (built at the first call with build=True; None when it does not pay)."""
        if not hasattr(self, "_mt"):
            if not build:
                return None
            g = self.graph
            self._mt = MtPlan.build(g, self.vals)
        return self._mt

    def permuted(self, v):
        """A node vector of the source data (pre / post of a descriptor) in this order; the data's own vectors map to the
        copies made above, anything else (a mask folded into pre / post) is gathered once and kept."""
        if v is None:
            return None
        src = self._source
        for name in ("dsqrt", "dinvsqrt", "degree", "degree_unnorm", "diag"):
            if v.data_ptr() == getattr(src, name).data_ptr():
                return getattr(self, name)
        key = (v.data_ptr(), v._version)
        hit = self._perm_cache.get(key)
        if hit is None:
            if len(self._perm_cache) > 8:
                self._perm_cache.clear()
            hit = (v, self.graph.permute(v).contiguous())      # (the source tensor is kept alive with its pointer key)
            self._perm_cache[key] = hit
        return hit[1]


_COO_CACHE = {}


def graph_for_coo(idx, val, n):
    """Operators built straight from (val, idx) as in the reference ctor share one CSR per edge list."""
    key = (idx.data_ptr(), val.data_ptr(), int(val.shape[0]), int(n), str(val.device))
    g = _COO_CACHE.get(key)
    if g is None:
        if len(_COO_CACHE) > 16:
            _COO_CACHE.clear()
        g = KnnGraph.from_coo(idx, val, n)
        g._keepalive = (idx, val)   # the cache key is a pointer: keep the tensors alive with it
        _COO_CACHE[key] = g
    return g


class LaplacianData:
    """degree_unnorm D~, degree D, diag, sqrt(D), 1/sqrt(D), CSR values S for one (eps, self_loops)."""
    _next_uid = [1]

    def __init__(self, graph, eps, self_loops):
        # never reused, unlike id(): what caches key on (solvers._cached_plan)
        self.uid = LaplacianData._next_uid[0]
        LaplacianData._next_uid[0] += 1
        dev = graph.device
        n = graph.n
        self.graph = graph
        self.eps = float(eps)
        self.self_loops = bool(self_loops)
        f = dict(dtype=torch.float32, device=dev)
        self.degree_unnorm = torch.empty(n, **f)
        self.degree = torch.empty(n, **f)
        self.diag = torch.empty(n, **f)
        self.dsqrt = torch.empty(n, **f)
        self.dinvsqrt = torch.empty(n, **f)
        self.vals = torch.empty(graph.nnz, **f)
        check(lib().mgp_laplacian_build(n, ptr(graph.rowptr), ptr(graph.col), ptr(graph.d2), self.eps,
                                        int(self.self_loops), ptr(self.degree_unnorm), ptr(self.degree),
                                        ptr(self.diag), ptr(self.dsqrt), ptr(self.dinvsqrt), ptr(self.vals),
                                        stream()), "mgp_laplacian_build")
        self._edge = {}
        self.vals_t = graph.tile_values(self.vals)       # tiles over a locality order stream their own copy

    def tangent(self):
        """d/d eps of every array of this object (mgp_laplacian_tangent), cached."""
        if getattr(self, "_tangent", None) is None:
            g = self.graph
            f = dict(dtype=torch.float32, device=g.device)

            class Tangent:
                pass
            t = Tangent()
            t.d_degree_unnorm, t.d_degree = torch.empty(g.n, **f), torch.empty(g.n, **f)
            t.d_diag, t.d_dsqrt, t.d_dinvsqrt = torch.empty(g.n, **f), torch.empty(g.n, **f), torch.empty(g.n, **f)
            t.d_vals = torch.empty(g.nnz, **f)
            check(lib().mgp_laplacian_tangent(g.n, ptr(g.rowptr), ptr(g.col), ptr(g.d2), self.eps, int(self.self_loops),
                                              ptr(self.degree_unnorm), ptr(self.degree), ptr(self.diag),
                                              ptr(t.d_degree_unnorm), ptr(t.d_degree), ptr(t.d_diag), ptr(t.d_dsqrt),
                                              ptr(t.d_dinvsqrt), ptr(t.d_vals), stream()), "mgp_laplacian_tangent")
            t.d_vals_t = g.tile_values(t.d_vals)
            self._tangent = t
        return self._tangent

    def csr(self, wide=False):
        """wide: the caller is about to multiply 48 columns or more -- build the matrix-core tile image if there is none
        yet (once built it rides in every struct)."""
        return self.graph.csr_with(self.vals, self.diag, self.vals_t, mt=self.mt_plan(wide))

    def mt_plan(self, build=True):
        """MtPlan of this CSR (built at the first call with build=True): only for graphs in natural row order -- a graph whose
        tiles follow a locality ORDER is multiplied through its relabelled copy (RelabelledData), which has its own."""
        if not hasattr(self, "_mt"):
            if not build:
                return None
            g = self.graph
            ordered = g.tiles is not None and g.tiles.get("rowid") is not None
            self._mt = None if ordered else MtPlan.build(g, self.vals)
        return self._mt

    def relabelled(self):
        """RelabelledData (cached) when the graph's tiles follow a locality order, else None."""
        if not self.graph.has_locality_order() or self.vals_t is None:
            return None
        if getattr(self, "_relabelled", None) is None:
            self._relabelled = RelabelledData(self)
        return self._relabelled

    def wide_relabelled(self):
        """RelabelledData over the graph's chain order for the wide products (KnnGraph.wide_relabelled), else the graph's own
        relabelled data (locality-ordered tiles), else None."""
        if getattr(self, "_wide_rel", False) is False:
            rg = self.graph.wide_relabelled() if hasattr(self.graph, "wide_relabelled") else None
            self._wide_rel = RelabelledData(self, rg) if rg is not None else None
        return self._wide_rel if self._wide_rel is not None else self.relabelled()

    def edge_values(self, which):
        """0: W (adjacency_unnorm_mat), 1: A (adjacency_mat), 2: S (laplacian_triu) in COO order."""
        out = self._edge.get(which)
        if out is None:
            g = self.graph
            out = torch.empty(g.M, dtype=torch.float32, device=g.device)
            check(lib().mgp_edge_values(ptr(g.tri_row), ptr(g.tri_col), ptr(g.tri_val), g.M,
                                        ptr(self.degree_unnorm), ptr(self.degree), self.eps, which, ptr(out),
                                        stream()), "mgp_edge_values")
            self._edge[which] = out
        return out
