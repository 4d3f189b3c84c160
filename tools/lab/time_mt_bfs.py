"""Lab: the matrix-core tile SpMM on the C3 bench graph relabelled by a breadth-first order (P L P^T) against the given order."""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
from manifold_gp_amd.graph import build_tiles, bfs_order, MtPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
data = lap.data
assert data.relabelled() is None
class G: pass
def relabel(order):
    t = build_tiles(g.n, g.rowptr, g.col, g.nnz, order=order)
    o = t["rowid"].long(); inv = torch.empty_like(o); inv[o] = torch.arange(g.n, device=dev)
    rg = G(); rg.n, rg.nnz, rg.M = g.n, g.nnz, g.M
    rg.rowptr = t["tile_rowptr"]; rg.col = inv.index_select(0, g.col.long().index_select(0, t["emap"])).to(torch.int32)
    vals = data.vals.index_select(0, t["emap"]); diag = data.diag.index_select(0, o).contiguous()
    return rg, vals, diag
def timeit(rg, vals, diag, C):
    plan = MtPlan.build(rg, vals)
    csr = _lib.csr_struct(rg.n, rg.rowptr, rg.col, vals, diag, tiles=None, mt=plan)
    X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
    ms = ctypes.c_float(0.0)
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 5, None, _lib.stream()), "repeat")
    best = 1e9
    for _ in range(3):
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
        best = min(best, ms.value)
    return best / 30 * 1e3, plan.steps, lib.mgp_spmm_kernel_choice(ctypes.byref(csr), C, 0, 0)
rg0 = G(); rg0.n, rg0.nnz, rg0.M, rg0.rowptr, rg0.col = g.n, g.nnz, g.M, g.rowptr, g.col
rgb, vb, db = relabel(bfs_order(g.n, g.rowptr, g.col))
for C in (64, 128):
    t0, s0, k0 = timeit(rg0, data.vals, data.diag, C)
    t1, s1, k1 = timeit(rgb, vb, db, C)
    print("C %3d  given order: %.1f us (%d steps, kernel %d)   BFS order: %.1f us (%d steps, kernel %d)" % (C, t0, s0, k0, t1, s1, k1), flush=True)
