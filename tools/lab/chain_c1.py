"""Lab: the C = 1 / C = 12 tile kernels and the headline CG solve on the chain-ordered C3 graph: (a) natural order, (b) tiles over the
chain order with vectors in the caller's order (rowid indirection inside the kernels), (c) solves on the chain-relabelled matrix."""
import ctypes, os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib, solvers
from manifold_gp_amd.graph import build_tiles, chain_order, LaplacianData, RelabelledGraph, RelabelledData
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap, desc = wl["graph"], wl["lap"], wl["desc"]
lib = _lib.lib()
y = wl["y"].view(-1, 1).contiguous()

def spmm_us(csr, C, reps=200):
    X = torch.rand(g.n, C, device=dev); Y = torch.empty_like(X); ms = ctypes.c_float(0)
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 20, None, _lib.stream()), "r")
    best = 1e9
    for _ in range(3):
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), reps, ctypes.byref(ms), _lib.stream()), "r")
        best = min(best, ms.value)
    return best / reps * 1e3

def solve_us(d, relabel, n=2000):
    solvers.RELABEL_SOLVES[0] = relabel
    plan = CgPlan(d, 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8)
    for _ in range(300): plan.solve(y, copy=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): plan.solve(y, copy=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e6
    its = plan.iters; plan.close(); solvers.RELABEL_SOLVES[0] = True
    return dt, its

print("(a) natural order: C=1 %.2f us, C=12 %.2f us, tiles reuse %.2f; CG solve %.1f us (%d it)" % (
    spmm_us(lap.data.csr(), 1), spmm_us(lap.data.csr(), 12, 100), g.tiles["reuse"], *solve_us(desc, False)))
t0 = time.perf_counter()
order = chain_order(g.n, g.rowptr, g.col, g.d2)
torch.cuda.synchronize(); print("chain order: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t64 = build_tiles(g.n, g.rowptr, g.col, g.nnz, order=order)
keep = g.tiles
g.tiles = t64
data_b = LaplacianData(g, lap.data.eps, True)
desc_b = desc.with_(data=data_b, pre=data_b.dsqrt if desc.pre is not None else None, post=data_b.dsqrt if desc.post is not None else None)
print("(b) chain-ordered tiles, caller-order vectors: C=1 %.2f us, C=12 %.2f us, reuse %.2f; CG solve %.1f us (%d it)" % (
    spmm_us(data_b.csr(), 1), spmm_us(data_b.csr(), 12, 100), t64["reuse"], *solve_us(desc_b, False)))
rel = data_b.relabelled()
print("(c) relabelled: C=1 %.2f us, C=12 %.2f us; CG solve (permute in / out) %.1f us (%d it)" % (
    spmm_us(rel.csr(), 1), spmm_us(rel.csr(), 12, 100), *solve_us(desc_b, True)))
g.tiles = keep
