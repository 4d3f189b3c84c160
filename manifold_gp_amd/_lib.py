"""ctypes binding of libmgp_hip.so (the C-ABI declared in include/mgp_hip.h).

The HIP library IS the product path: there is no CPU or torch fallback.  `lib()` raises when
the shared object is missing, and every wrapper raises when handed a tensor that does not live
on a HIP device.
"""
import ctypes
import os
import threading
from ctypes import (POINTER, Structure, byref, c_double, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint64, c_void_p)

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgp_hip.so")
_LIB = None
_LOCK = threading.Lock()

MGP_ERRORS = {
    -1: "MGP_ERR_ARG (null pointer / bad size / bad enum)",
    -2: "MGP_ERR_WORKSPACE (work buffer too small)",
    -3: "MGP_ERR_UNSUPPORTED",
    -4: "MGP_ERR_NOT_CONVERGED",
    -5: "MGP_ERR_KNN_AMBIGUOUS",
}


class MgpError(RuntimeError):
    def __init__(self, fn, code):
        self.code = code
        what = MGP_ERRORS.get(code, "hipError_t %d" % code if code > 0 else "error %d" % code)
        super().__init__("%s failed: %s" % (fn, what))


class CsrT(Structure):
    _fields_ = [("n", c_int64), ("rowptr", c_void_p), ("col", c_void_p), ("vals", c_void_p),
                ("diag", c_void_p),
                ("ncols", c_int64), ("tile_ptr", c_void_p), ("tile_cols", c_void_p), ("lid", c_void_p),
                ("tile_rows", c_int32), ("tile_max_cols", c_int32), ("tile_max_entries", c_int32),
                ("tile_reserved", c_int32), ("tile_rowptr", c_void_p), ("tile_vals", c_void_p),
                ("tile_rowid", c_void_p),
                ("mt_sptr", c_void_p), ("mt_dcol", c_void_p), ("mt_img", c_void_p), ("mt_tiles", c_int32), ("mt_steps", c_int32)]


class OperatorT(Structure):
    _fields_ = [("L", CsrT), ("pre", c_void_p), ("post", c_void_p), ("nu", c_int32),
                ("kappa", c_float), ("scale", c_float), ("form", c_int32), ("noise", c_float)]


class CgParamsT(Structure):
    _fields_ = [("tol", c_float), ("max_iter", c_int32), ("min_iter", c_int32),
                ("stop_mode", c_int32), ("check_every", c_int32), ("use_graph", c_int32),
                ("max_refine", c_int32)]


class LanczosParamsT(Structure):
    _fields_ = [("max_basis", c_int32), ("degree", c_int32), ("max_restarts", c_int32),
                ("tol", c_float), ("seed", c_uint64)]


_P = c_void_p
# name -> (restype, argtypes); mirrors include/mgp_hip.h one to one
SIGNATURES = {
    "mgp_version": (c_int, []),
    "mgp_device_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "mgp_knn_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int, c_int]),
    "mgp_knn_search": (c_int, [_P, c_int64, c_int, _P, c_int64, c_int, _P, _P, _P, c_size_t,
                               POINTER(c_int64), _P]),
    "mgp_knn_set_mfma": (c_int, [c_int]),
    "mgp_knn_set_symmetric": (c_int, [c_int]),
    "mgp_knn_set_filter": (c_int, [c_int]),
    "mgp_knn_last_filter_failover": (c_int64, []),
    "mgp_knn_index_bytes": (c_size_t, [c_int64, c_int]),
    "mgp_knn_index_build": (c_int, [_P, c_int64, c_int, _P, c_size_t, _P]),
    "mgp_knn_search_indexed": (c_int, [_P, c_int64, c_int, _P, c_size_t, _P, c_int64, c_int, _P, _P, _P, c_size_t, _P, _P]),
    "mgp_knn_last_direct_chunks": (c_int64, []),
    "mgp_graph_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mgp_graph_edges": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P, _P, _P, POINTER(c_int64), _P, c_size_t, _P]),
    "mgp_graph_tiles_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mgp_graph_tiles": (c_int, [c_int64, _P, _P, c_int64, c_int, _P, _P, _P, _P, _P, _P, POINTER(c_int64),
                                POINTER(c_int32), POINTER(c_int32), _P, c_size_t, _P]),
    "mgp_morton_order_workspace_bytes": (c_size_t, [c_int64]),
    "mgp_morton_order": (c_int, [_P, c_int64, c_int, _P, _P, c_size_t, _P]),
    "mgp_graph_chain_order": (c_int, [c_int64, _P, _P, _P, _P, _P]),
    "mgp_permute_rows": (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    "mgp_graph_bfs_workspace_bytes": (c_size_t, [c_int64]),
    "mgp_graph_bfs_order": (c_int, [c_int64, _P, _P, _P, _P, c_size_t, _P]),
    "mgp_spmm_dot_blocks_csr": (c_int, [POINTER(CsrT), c_int]),
    "mgp_spmm_set_tile_mode": (c_int, [c_int]),
    "mgp_spmm_set_tile_small_mode": (c_int, [c_int]),
    "mgp_spmm_set_tile_wide_mode": (c_int, [c_int]),
    "mgp_spmm_set_dict_mode": (c_int, [c_int]),
    "mgp_spmm_set_mt_mode": (c_int, [c_int]),
    "mgp_spmm_kernel_choice": (c_int, [POINTER(CsrT), c_int, c_int, c_int64]),
    "mgp_spmm_mt_fill": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    "mgp_spmm_timing_begin": (c_int, [c_int]),
    "mgp_spmm_timing_end": (c_int, [POINTER(c_float), POINTER(c_int)]),
    "mgp_spmm_set_v4_mode": (c_int, [c_int]),
    "mgp_cg_set_decide_in_update": (c_int, [c_int]),
    "mgp_cg_set_complex_shift": (c_int, [c_int]),
    "mgp_cg_plan_is_complex_shift": (c_int, [_P]),
    "mgp_cg_set_reduce_once": (c_int, [c_int]),
    "mgp_cg_set_update_quads": (c_int, [c_int]),
    "mgp_cg_set_poll_spin": (c_int, [c_int]),
    "mgp_cg_set_init_free": (c_int, [c_int]),
    "mgp_host_symeig": (c_int, [c_int, _P, _P, _P]),
    "mgp_lanczos_set_bound_mode": (c_int, [c_int]),
    "mgp_blz_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "mgp_blz_begin": (c_int, [_P, c_int64, c_int, c_int, _P, c_size_t, _P]),
    "mgp_blz_q": (c_void_p, [c_int64, c_int, c_int, c_int, _P, c_size_t]),
    "mgp_blz_step": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mgp_blz_end": (c_int, [c_int64, c_int, c_int, POINTER(c_float), POINTER(c_float), _P, c_size_t, _P]),
    "mgp_lanczos_tridiag_block_workspace_bytes": (c_size_t, [POINTER(OperatorT), c_int, c_int]),
    "mgp_lanczos_tridiag_block": (c_int, [POINTER(OperatorT), _P, c_int, c_int, POINTER(c_float), POINTER(c_float), _P,
                                         c_size_t, _P]),
    "mgp_graph_build": (c_int, [_P, _P, c_int64, c_int, _P, _P, _P, POINTER(c_int64), _P, _P, _P, _P,
                                POINTER(c_int64), _P, c_size_t, _P]),
    "mgp_graph_coo_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "mgp_graph_from_coo": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, POINTER(c_int64), _P,
                                   c_size_t, _P]),
    "mgp_laplacian_build": (c_int, [c_int64, _P, _P, _P, c_float, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "mgp_laplacian_tangent": (c_int, [c_int64, _P, _P, _P, c_float, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mgp_edge_values": (c_int, [_P, _P, _P, c_int64, _P, _P, c_float, c_int, _P, _P]),
    "mgp_spmm_backward_blocks": (c_int, [c_int64]),
    "mgp_spmm_backward_sums": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P]),
    "mgp_spmm_dot_blocks": (c_int, [c_int64, c_int]),
    "mgp_spmm_set_group_hint": (c_int, [c_int]),
    "mgp_spmm_set_rows_in_flight": (c_int, [c_int]),
    "mgp_spmm_fused": (c_int, [POINTER(CsrT), _P, c_int, _P, c_float, c_float, _P, _P, _P, c_float,
                               c_float, _P, _P, _P]),
    "mgp_spmm_fused_rows": (c_int, [POINTER(CsrT), c_int64, _P, c_int, _P, c_float, c_float, _P, _P, _P, c_float,
                                    c_float, _P, _P, _P]),
    "mgp_spmm_repeat": (c_int, [POINTER(CsrT), _P, c_int, _P, c_int, POINTER(c_float), _P]),
    "mgp_laplacian_matmul": (c_int, [POINTER(CsrT), _P, _P, c_int, _P, c_int, _P, _P, _P]),
    "mgp_operator_workspace_bytes": (c_size_t, [POINTER(OperatorT), c_int]),
    "mgp_operator_apply": (c_int, [POINTER(OperatorT), _P, c_int, _P, _P, c_size_t, _P]),
    "mgp_operator_apply_dot": (c_int, [POINTER(OperatorT), _P, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "mgp_operator_jacobi": (c_int, [POINTER(OperatorT), _P, _P]),
    "mgp_cg_workspace_bytes": (c_size_t, [POINTER(OperatorT), c_int]),
    "mgp_cg_solve": (c_int, [POINTER(OperatorT), _P, c_int, _P, _P, POINTER(CgParamsT), POINTER(c_int32),
                             POINTER(c_float), _P, c_size_t, _P]),
    "mgp_cg_plan_create": (c_int, [POINTER(OperatorT), c_int, _P, POINTER(CgParamsT), _P, c_size_t, _P,
                                   POINTER(c_void_p)]),
    "mgp_cg_plan_solve": (c_int, [_P, _P, _P, POINTER(c_int32), POINTER(c_float), POINTER(c_int32)]),
    "mgp_cg_plan_rebind": (c_int, [_P, POINTER(OperatorT), _P]),
    "mgp_cg_plan_x": (c_void_p, [_P]),
    "mgp_cg_plan_x64": (c_void_p, [_P]),
    "mgp_cg_plan_last_applies": (c_int, [_P]),
    "mgp_cg_plan_destroy": (c_int, [_P]),
    "mgp_cg_plan_poisoned": (c_int, [_P]),
    "mgp_cg_plan_poison": (c_int, [_P]),
    "mgp_dist_unique_id_bytes": (c_int, []),
    "mgp_dist_unique_id": (c_int, [_P]),
    "mgp_dist_init": (c_int, [c_int, c_int, _P, POINTER(c_void_p)]),
    "mgp_dist_destroy": (c_int, [_P]),
    "mgp_dist_comm_info": (c_int, [_P, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "mgp_dist_allgather": (c_int, [_P, c_int, c_int, _P, c_int64, _P]),
    "mgp_operator_apply_part": (c_int, [POINTER(OperatorT), _P, c_int, c_int, _P, c_int, _P, _P, c_size_t, _P]),
    "mgp_cg_dist_workspace_bytes": (c_size_t, [POINTER(OperatorT), c_int, c_int]),
    "mgp_cg_plan_create_dist": (c_int, [POINTER(OperatorT), c_int, _P, POINTER(CgParamsT), _P, c_int, c_int, _P,
                                        c_size_t, _P, POINTER(c_void_p)]),
    "mgp_pcg_shared_floats": (c_size_t, [c_int64, c_int64, c_int]),
    "mgp_pcg_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "mgp_pcg_plan_create": (c_int, [POINTER(OperatorT), POINTER(c_int64), c_int64, c_int64, c_int64, _P, c_int, c_int, _P, c_int,
                                    POINTER(CgParamsT), _P, c_size_t, _P, POINTER(c_void_p)]),
    "mgp_pcg_plan_solve": (c_int, [_P, _P, _P, POINTER(c_int32), POINTER(c_float), POINTER(c_int32)]),
    "mgp_pcg_plan_enqueue": (c_int, [_P, c_int, c_int, _P]),
    "mgp_pcg_plan_poll": (c_int, [_P, POINTER(c_int32), POINTER(c_float), POINTER(c_int32)]),
    "mgp_pcg_plan_x": (c_void_p, [_P]),
    "mgp_pcg_plan_destroy": (c_int, [_P]),
    "mgp_pcg_plan_poisoned": (c_int, [_P]),
    "mgp_lanczos_workspace_bytes": (c_size_t, [c_int64, c_int, POINTER(LanczosParamsT)]),
    "mgp_lanczos_smallest": (c_int, [POINTER(CsrT), c_int, POINTER(LanczosParamsT), POINTER(c_float), _P,
                                     POINTER(c_float), POINTER(c_int32), _P, c_size_t, _P]),
    "mgp_lanczos_block_size": (c_int, [c_int, POINTER(LanczosParamsT)]),
    "mgp_lanczos_smallest_ex": (c_int, [POINTER(CsrT), c_int, POINTER(LanczosParamsT), POINTER(c_float), _P,
                                        POINTER(c_float), POINTER(c_int32), POINTER(c_float), _P, POINTER(c_float), _P,
                                        c_size_t, _P]),
    "mgp_lanczos_smallest_warm": (c_int, [POINTER(CsrT), c_int, POINTER(LanczosParamsT), POINTER(c_float), _P,
                                          POINTER(c_float), POINTER(c_int32), POINTER(c_float), _P, POINTER(c_float), _P,
                                          POINTER(c_float), _P, c_size_t, _P]),
    "mgp_lanczos_tridiag_workspace_bytes": (c_size_t, [POINTER(OperatorT), c_int]),
    "mgp_lanczos_tridiag": (c_int, [POINTER(OperatorT), _P, c_int, POINTER(c_float), POINTER(c_float), _P, _P,
                                    c_size_t, _P]),
    "mgp_eigvec_postprocess_work_floats": (c_size_t, [c_int]),
    "mgp_eigvec_postprocess": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "mgp_features_insample": (c_int, [_P, _P, c_int64, c_int, c_int, c_float, _P, _P]),
    "mgp_features_oos": (c_int, [_P, _P, c_int64, c_int, c_int, c_float, c_float, c_int, _P, _P, _P, _P,
                                 c_int64, c_int, c_float, c_float, _P, _P]),
    "mgp_kernel_block": (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, _P, _P]),
    "mgp_kernel_block_set_pipe": (c_int, [c_int]),
    "mgp_kernel_diag": (c_int, [_P, _P, c_int64, c_int, c_float, _P, _P]),
    "mgp_lowrank_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mgp_lowrank_apply": (c_int, [_P, c_int64, c_int, _P, c_int, c_float, c_float, _P, _P, c_size_t, _P]),
    "mgp_gram_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "mgp_gram_f64": (c_int, [_P, c_int64, c_int, _P, _P, c_size_t, _P]),
    "mgp_gram_set_mfma": (c_int, [c_int]),
    "mgp_lowrank_residual": (c_int, [_P, c_int64, c_int, _P, _P, c_int, c_double, _P, _P]),
}


def lib():
    """Load libmgp_hip.so once.  Raises (no fallback) when it has not been built."""
    global _LIB
    if _LIB is None:
        with _LOCK:
            if _LIB is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "manifold_gp_amd: %s is missing -- the HIP extension is the product path and "
                        "there is no fallback.  Build it with `python -c \"import __graft_entry__ as g; "
                        "g.build()\"` or manifold_gp_amd/csrc/build.sh" % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)   # AttributeError = header/library mismatch
                    fn.restype = res
                    fn.argtypes = args
                _LIB = handle
    return _LIB


def check(code, fn):
    if code != 0:
        raise MgpError(fn, code)


def require_device(*tensors):
    """The product path runs on the MI355X only: refuse host tensors instead of computing on CPU."""
    for t in tensors:
        if t is None:
            continue
        if not torch.is_tensor(t) or t.device.type != "cuda":
            raise RuntimeError("manifold_gp_amd: expected a tensor on a HIP device (cuda:N), got %s -- "
                               "there is no CPU path" % (t.device if torch.is_tensor(t) else type(t)))


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def host_scalar(owner, name):
    """float(owner.<name>) for a hyper-parameter an operator holds, read from the device ONCE per (tensor object, version): an
    operator asks for its length scale / scale / noise every time it describes itself (tens of times per training epoch), and every
    `.item()` is a host wait that drains the queue.  The tensor is the operator's own attribute, so identity + `_version` say whether
    the cached value still holds."""
    t = getattr(owner, name)
    if not torch.is_tensor(t):
        return float(t)
    cache = owner.__dict__.setdefault("_host_scalars", {})
    hit = cache.get(name)
    if hit is not None and hit[0] is t and hit[1] == t._version:
        return hit[2]
    v = float(t.reshape(-1)[0].item())
    cache[name] = (t, t._version, v)
    return v


def f32c(t):
    """float32 + contiguous (the reference calls rhs.contiguous() at every _matmul)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_WORK = {}


def workspace(nbytes, tag, device):
    """Per-(tag, device) scratch tensor, grown geometrically; owned by the torch caching allocator."""
    key = (tag, str(device))
    buf = _WORK.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, device=device)
        _WORK[key] = buf
    return buf


def workspace_bytes(tag, device):
    """Size of the cached scratch of (tag, device); 0 when there is none."""
    buf = _WORK.get((tag, str(device)))
    return 0 if buf is None else int(buf.numel())


def release_workspace(tag, device=None):
    """Drop the cached scratch of `tag` (all devices when device is None): the 60k x 60k key slab of a graph build is
    ~18 GB that nothing needs once the lists exist (the allocator hands it to the next large request)."""
    for key in [k for k in _WORK if k[0] == tag and (device is None or k[1] == str(device))]:
        del _WORK[key]


_LEAKED = []


def leak(*objects):
    """Keep `objects` (workspace tensors of a poisoned plan: queued kernels still reference them) alive for the rest of the
    process."""
    _LEAKED.extend(objects)


def csr_struct(n, rowptr, col, vals, diag, ncols=0, tiles=None, tile_vals=None, mt=None):
    """tiles: None or the dict KnnGraph.tiles holds (tile_ptr, tile_cols, lid tensors + rows / max_cols /
    max_entries; for tiles over a row order also tile_rowptr / rowid / emap).  tile_vals: `vals` gathered
    through tiles["emap"] -- required with ordered tiles, which are otherwise left out of the struct.
    mt: None or a graph.MtPlan built over the same rowptr / col / vals."""
    s = CsrT(int(n), rowptr.data_ptr(), col.data_ptr(), vals.data_ptr(), diag.data_ptr(), int(ncols))
    if tiles is not None and (tiles.get("rowid") is None or tile_vals is not None):
        s.tile_ptr, s.tile_cols, s.lid = tiles["tile_ptr"].data_ptr(), tiles["tile_cols"].data_ptr(), tiles["lid"].data_ptr()
        s.tile_rows, s.tile_max_cols, s.tile_max_entries = tiles["rows"], tiles["max_cols"], tiles["max_entries"]
        if tiles.get("rowid") is not None:
            s.tile_rowptr, s.tile_rowid = tiles["tile_rowptr"].data_ptr(), tiles["rowid"].data_ptr()
            s.tile_vals = tile_vals.data_ptr()
    if mt is not None:        # graph.MtPlan: dense 16-row tiles for the matrix-core SpMM (48 <= C <= 256)
        s.mt_sptr, s.mt_dcol, s.mt_img = mt.sptr.data_ptr(), mt.dcol.data_ptr(), mt.img.data_ptr()
        s.mt_tiles, s.mt_steps = mt.tiles, mt.steps
    return s
