"""NearestNeighbors on MI355X: the interface of manifold_gp/utils/nearest_neighbors.py:10-63
(train / search / graph, `min_ivf`, `nlist`, `nprobe`) over the exact HIP k-NN kernels.

faiss' Flat and IVFFlat(nlist=1) indices are both exhaustive searches, so `train` records the points and, from
32 features up, prepares what every search against them needs (the matrix-core operands of the candidate keys:
`mgp_knn_index_build`); `search` returns (D f32 squared-L2 ascending, I int64); `graph` returns the reference's
(idx[2,M] int64 with row<col sorted, val[M] mean squared distance) and keeps the padded CSR of the
same graph in `self.knn_graph` for the operators."""
import ctypes

import torch

from .. import _lib
from .._lib import check, lib, ptr, stream
from ..graph import KnnGraph


class NearestNeighbors():
    def __init__(self, x=None, nlist=1) -> None:
        self.min_ivf = 5000
        self.knn_graph = None
        self.last_stats = None
        self._index = None
        self._index_version = None
        if x is not None:
            self.train(x, nlist)

    def train(self, x, nlist=1):
        _lib.require_device(x)
        if x.dim() != 2:
            raise ValueError("x must be [n, d]")
        self.x = x
        self._xc = _lib.f32c(x)
        self.nlist = nlist
        self.is_trained = True
        self._build_index()
        return self

    def _build_index(self):
        """The part of a search that depends on the points alone (faiss: index.train / index.add,
        nearest_neighbors.py:20-33): built once per train() -- and again when x was modified in place since (torch's
        version counter), because the index is a snapshot.  None when d < 32 (nothing to prepare)."""
        N, d = self._xc.shape
        nb = int(lib().mgp_knn_index_bytes(N, d))
        self._index = None
        self._index_version = self.x._version
        if nb > 0:
            self._index = torch.empty(nb, dtype=torch.uint8, device=self._xc.device)
            check(lib().mgp_knn_index_build(ptr(self._xc), N, d, ptr(self._index), nb, stream()), "mgp_knn_index_build")

    def search(self, x, k, nprobe=1):
        _lib.require_device(x)
        assert self.is_trained
        q = _lib.f32c(x)
        N, d = self._xc.shape
        n = q.shape[0]
        if q.shape[1] != d:
            raise ValueError("query dimension %d != index dimension %d" % (q.shape[1], d))
        if not (0 < k <= min(N, 1024)):
            raise ValueError("k must be in [1, min(N, 1024)]")
        D = torch.empty(n, k, dtype=torch.float32, device=q.device)
        I = torch.empty(n, k, dtype=torch.int32, device=q.device)
        wb = lib().mgp_knn_workspace_bytes(N, n, d, k)
        work = _lib.workspace(wb, "knn", q.device)
        stats = (ctypes.c_int64 * 4)()
        if self._index is not None and self.x._version != self._index_version:
            self._xc = _lib.f32c(self.x)
            self._build_index()                            # x changed in place since train(): the snapshot is stale
        if self._index is not None:
            check(lib().mgp_knn_search_indexed(ptr(self._xc), N, d, ptr(self._index), self._index.numel(), ptr(q), n, k,
                                               ptr(D), ptr(I), ptr(work), work.numel(), stats, stream()),
                  "mgp_knn_search_indexed")
        else:
            check(lib().mgp_knn_search(ptr(self._xc), N, d, ptr(q), n, k, ptr(D), ptr(I), ptr(work), work.numel(),
                                       stats, stream()), "mgp_knn_search")
        self.last_stats = dict(rows_redone_wide=stats[0], rows_redone_exact=stats[1], chunks=stats[2],
                               candidates=stats[3], chunks_redone_direct=int(lib().mgp_knn_last_direct_chunks()),
                               filter_failover_rows=int(lib().mgp_knn_last_filter_failover()))
        return D, I.long()

    def graph(self, k, symmetric=True, self_loop=False, nprobe=1):
        val, idx = self.search(self.x, k, nprobe)
        n = self.x.shape[0]
        # the self-search's scratch is dropped once the lists exist when it is the whole-matrix key slab (14.8 GB at 60k x 784, up to
        # 32 GiB); the candidate filter's 4.9 GB stay cached: a later construction in the same process otherwise pays a fresh
        # device allocation of that size (~100 ms against a 12 ms search; tools/lab/first_eval.py)
        wb = _lib.workspace_bytes("knn", self.x.device)
        if wb > (8 << 30):
            _lib.release_workspace("knn", self.x.device)
        if symmetric and not self_loop:
            self.knn_graph = KnnGraph.from_knn(val, idx.to(torch.int32), points=self.x if self.x.shape[1] <= 3 else None)
            return self.knn_graph.edge_index, self.knn_graph.edge_value
        # non-default variants (nearest_neighbors.py:42-53): the edge list alone, no CSR (mgp_graph_edges)
        first = 0 if self_loop else 1
        total = n * (k - first)
        dev = self.x.device
        row = torch.empty(total, dtype=torch.int32, device=dev)
        col = torch.empty(total, dtype=torch.int32, device=dev)
        out = torch.empty(total, dtype=torch.float32, device=dev)
        M = ctypes.c_int64(0)
        work = None
        if symmetric:
            wb = lib().mgp_graph_workspace_bytes(n, k + 1)
            work = _lib.workspace(wb, "graph", dev)
        check(lib().mgp_graph_edges(ptr(_lib.f32c(val)), ptr(idx.to(torch.int32).contiguous()), n, int(k), int(bool(symmetric)),
                                    int(bool(self_loop)), ptr(row), ptr(col), ptr(out), ctypes.byref(M), ptr(work),
                                    work.numel() if work is not None else 0, stream()), "mgp_graph_edges")
        m = M.value
        return torch.stack([row[:m], col[:m]]).long(), out[:m].clone()

    @property
    def min_ivf(self):
        return self._min_ivf

    @min_ivf.setter
    def min_ivf(self, value):
        self._min_ivf = value
