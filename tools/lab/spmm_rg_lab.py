#!/usr/bin/env python
"""Lab: the row-group SpMM prototype (spmm_rg_lab.hip) against the production SpMM on the C3 graph.
spmm_rg_lab.py [C] -- builds the merged stream on the CPU, checks Y, times both."""
import ctypes, os, sys, argparse, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
csr = lap.data.csr()
lab = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspmm_rg_lab.so"))
n = g.n
rowptr = g.rowptr.cpu().numpy().astype(np.int64); col = g.col.cpu().numpy(); vals = lap.data.vals.cpu().numpy()
diag = lap.data.diag
U = 8
def build(R):
    ng = (n + R - 1) // R
    ptr = np.zeros(ng + 1, np.int32); cols = []; vv = []
    for q in range(ng):
        rows = range(q * R, min(n, (q + 1) * R))
        cs = np.concatenate([col[rowptr[r]:rowptr[r + 1]] for r in rows])
        u = np.unique(cs[cs >= 0])
        m = np.zeros((len(u), R), np.float32)
        for j, r in enumerate(rows):
            c = col[rowptr[r]:rowptr[r + 1]]; v = vals[rowptr[r]:rowptr[r + 1]]
            ok = c >= 0
            np.add.at(m[:, j], np.searchsorted(u, c[ok]), v[ok])
        pad = (-len(u)) % U
        if pad:
            u = np.concatenate([u, np.full(pad, u[-1] if len(u) else 0, u.dtype)]); m = np.concatenate([m, np.zeros((pad, R), np.float32)])
        cols.append(u.astype(np.int32)); vv.append(m)
        ptr[q + 1] = ptr[q] + len(u) // U
    return (torch.from_numpy(ptr).to(dev), torch.from_numpy(np.concatenate(cols)).to(dev), torch.from_numpy(np.concatenate(vv)).to(dev).contiguous())
Cs = [int(sys.argv[1])] if len(sys.argv) > 1 else [128, 100, 64]
for R in ([int(sys.argv[2])] if len(sys.argv) > 2 else (4, 8)):
    ptr, cols, vv = build(R)
    print("R %d: %d batches, stream %.1f MB" % (R, int(ptr[-1]), (cols.numel() * 4 + vv.numel() * 4) / 1e6), flush=True)
    for C in Cs:
        X = torch.randn(n, C, device=dev); Y = torch.empty_like(X); Y2 = torch.zeros_like(X)
        ms = ctypes.c_float(0.0)
        lib.mgp_spmm_set_dict_mode(0)
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
        t_ref = ms.value / 30 * 1e3
        best = 1e9
        for rep in range(3):
            rc = lab.lab_spmm_rg(R, ctypes.c_void_p(ptr.data_ptr()), ctypes.c_void_p(cols.data_ptr()), ctypes.c_void_p(vv.data_ptr()), ctypes.c_void_p(diag.data_ptr()),
                                 ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(Y2.data_ptr()), n, C, 30, ctypes.byref(ms), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
            best = min(best, ms.value / 30 * 1e3)
        torch.cuda.synchronize()
        err = (Y - Y2).abs().max().item() / Y.abs().max().item()
        print("  C %3d: production %.1f us   row groups of %d: %.1f us   max rel diff %.2e  bitwise equal %s" % (C, t_ref, R, best, err, bool((Y == Y2).all())), flush=True)
