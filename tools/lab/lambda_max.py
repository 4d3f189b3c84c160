"""Lab: how loose is the Gershgorin bound the eigensolver's Chebyshev filter uses?  Power iteration / Lanczos estimate of lambda_max."""
import os, sys, argparse, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
dev = torch.device("cuda:0")
for wlname in sys.argv[1:] or ["c3"]:
    wl = bench.build_workload(argparse.Namespace(workload=wlname, nodes=0, s5_order="morton"), dev, 0, 1)
    g, lap = wl["graph"], wl["lap"]
    lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
    csr = lap.data.csr()
    n = g.n
    def L(x):
        y = torch.empty_like(x)
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(x), 1, _lib.ptr(y), 1, None, _lib.stream()), "spmm")
        return y
    # Gershgorin as the solver computes it: max_i (diag_i + sum_j |offdiag_ij|)
    rowptr = g.rowptr.long(); vals = lap.data.vals; diag = lap.data.diag
    rs = torch.zeros(n, device=dev).index_add_(0, torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1]), vals.abs())
    ger = (diag.abs() + rs).max().item()
    # Lanczos with full re-orthogonalisation, 40 steps, float64 on the host side of the recurrence
    k = 40
    q = torch.randn(n, 1, device=dev); q /= q.norm()
    Q = [q]; al = []; be = []
    for j in range(k):
        w = L(Q[-1].contiguous()).double()
        a = (w * Q[-1].double()).sum().item(); al.append(a)
        w = w - a * Q[-1].double() - (be[-1] * Q[-2].double() if j else 0)
        for qq in Q: w = w - (w * qq.double()).sum() * qq.double()
        b = w.norm().item(); be.append(b)
        Q.append((w / b).float())
    T = np.diag(al) + np.diag(be[:-1], 1) + np.diag(be[:-1], -1)
    th, S = np.linalg.eigh(T)
    bound = th[-1] + abs(be[-1] * S[-1, -1])
    print("%s: n %d  Gershgorin %.4f   Lanczos(40) theta_max %.4f   safe bound theta_max + |beta s_k| %.4f   ratio %.3f" % (wlname, n, ger, th[-1], bound, bound / ger), flush=True)
