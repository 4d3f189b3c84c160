"""CPU ORACLE (test infrastructure) -- exact k-NN search and symmetrised k-NN graph.

Follows manifold_gp/utils/nearest_neighbors.py:17-55.  faiss and torch_sparse are absent
(un-pinned third-party wheels, parity unpinned), so the oracle restates their published
behaviour: exact brute-force squared-L2 top-k ascending, and sort + mean coalescing.

Distance / tie rule (defined by this build, see oracle/knn_oracle.c): fp64 sum of squared
differences in ascending feature order without FMA contraction; ascending (d2, index).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(force=False):
    """Compile oracle/knn_oracle.c -> oracle/_build/liboracle.so (gcc).  Returns the path."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "_build/liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        try:
            lib = ctypes.CDLL(build_c_oracle())
            lib.oracle_knn_f64.restype = ctypes.c_int
            lib.oracle_knn_f64.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
            _LIB = lib
        except Exception:  # no compiler: numpy path below is the same arithmetic
            _LIB = False
    return _LIB


def knn_search_numpy(db, q, k):
    """Pure-numpy statement of the distance/tie rule (small cases; pins the C oracle)."""
    db = np.ascontiguousarray(db, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    n, d = q.shape
    N = db.shape[0]
    D = np.empty((n, k), np.float32)
    I = np.empty((n, k), np.int64)
    dbd = db.astype(np.float64)
    chunk = max(1, int(2e7 // max(N, 1)))
    for s in range(0, n, chunk):
        qd = q[s:s + chunk].astype(np.float64)
        acc = np.zeros((qd.shape[0], N), np.float64)
        for j in range(d):  # ascending j, each op individually rounded (no fma in numpy)
            df = qd[:, j:j + 1] - dbd[None, :, j]
            acc = acc + df * df
        # ascending (d2, index): stable argsort on d2 keeps the lower index first on ties
        order = np.argsort(acc, axis=1, kind="stable")[:, :k]
        I[s:s + chunk] = order
        D[s:s + chunk] = np.take_along_axis(acc, order, axis=1).astype(np.float32)
    return D, I


def knn_search(db, q, k):
    """nearest_neighbors.py:35-37 `search`: returns (D[n,k] f32 squared L2, I[n,k] i64)."""
    db = np.ascontiguousarray(db, dtype=np.float32)
    q = np.ascontiguousarray(q, dtype=np.float32)
    lib = _lib()
    if not lib:
        return knn_search_numpy(db, q, k)
    n, d = q.shape
    D = np.empty((n, k), np.float32)
    I = np.empty((n, k), np.int64)
    rc = lib.oracle_knn_f64(db.ctypes.data, db.shape[0], d, q.ctypes.data, n, k,
                            D.ctypes.data, I.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_knn_f64 failed: %d" % rc)
    return D, I


def coalesce_mean(rows, cols, vals, n):
    """torch_sparse.coalesce(op='mean') as used at nearest_neighbors.py:51: sort by
    (row, col), merge duplicates, fp32 mean (sum / count) of their values."""
    key = rows.astype(np.int64) * n + cols.astype(np.int64)
    order = np.argsort(key, kind="stable")
    key, v = key[order], vals[order].astype(np.float32)
    uniq, start, cnt = np.unique(key, return_index=True, return_counts=True)
    s = np.add.reduceat(v, start).astype(np.float32) if len(v) else v
    # reduceat sums in order within a segment; duplicates here are at most pairs
    out = (s / cnt.astype(np.float32)).astype(np.float32)
    idx = np.stack([uniq // n, uniq % n]).astype(np.int64)
    return idx, out


def knn_graph_from_search(D, I, n):
    """nearest_neighbors.py:39-55 `graph` (symmetric=True, self_loop=False) given (D, I):
    drop column 0, orient every edge as row<col, coalesce with mean."""
    val, idx = D[:, 1:], I[:, 1:]                                   # :42-43
    rows = np.repeat(np.arange(idx.shape[0], dtype=np.int64), idx.shape[1])   # :45
    cols = idx.reshape(-1).astype(np.int64)
    val = val.reshape(-1)
    split = cols > rows                                              # :49
    r = np.concatenate([rows[split], cols[~split]])                  # :50
    c = np.concatenate([cols[split], rows[~split]])
    v = np.concatenate([val[split], val[~split]])
    return coalesce_mean(r, c, v, n)                                 # :51


def knn_graph(x, k):
    """NearestNeighbors(x).graph(k): returns (idx[2,M] i64 with row<col sorted, val[M] f32)."""
    D, I = knn_search(x, x, k)
    return knn_graph_from_search(D, I, x.shape[0])
