#!/usr/bin/env python
"""Scale checks on the GPU box: eval() (eigensolve + features) at C3 size, optional S5 sizes."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import manifold_gp_amd as mgp  # noqa: E402
from manifold_gp_amd.solvers import lanczos_smallest, cg_solve  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--modes", type=int, default=100)
    ap.add_argument("--tol", type=float, default=1e-5)
    a = ap.parse_args()
    a.gpus = 1
    dev = torch.device("cuda:0")
    out = {}
    t0 = time.time()
    wl = bench.build_workload(a, dev, 0, 1)
    out["setup_s"] = round(time.time() - t0, 2)
    out["graph_s"] = round(wl["t_graph"], 3)
    g, lap = wl["graph"], wl["lap"]
    out.update(n=g.n, M=g.M, nnz=g.nnz)
    torch.cuda.synchronize()
    t0 = time.time()
    data = lap.data
    torch.cuda.synchronize()
    out["laplacian_build_ms"] = round((time.time() - t0) * 1e3, 3)
    t0 = time.time()
    evals, evecs, resid = lanczos_smallest(data, a.modes, tol=a.tol, max_restarts=60)
    torch.cuda.synchronize()
    out["eigensolve_s"] = round(time.time() - t0, 3)
    out["eig_info(outer,spmm,nconv,block)"] = lanczos_smallest.last_info
    ev = evals.cpu().numpy()
    out["evals_head"] = [float(v) for v in ev[:6]]
    out["evals_tail"] = [float(v) for v in ev[-3:]]
    out["max_resid"] = float(max(resid))
    # residual re-check with the operator itself
    sym = lap._symmetric_twin()
    R = sym.matmul(evecs) - evecs * evals.view(1, -1)
    out["recheck_resid_max"] = float(R.norm(dim=0).max())
    out["orth_err"] = float((evecs.t() @ evecs - torch.eye(a.modes, device=dev)).abs().max())
    # Q x = y solve (training-side conditioning) with and without Jacobi
    Q = mgp.operators.PrecisionMaternOperator(lap, wl["nu"], torch.tensor([[wl["hp"]["lengthscale"]]], device=dev))
    for jac in (False, True):
        torch.cuda.synchronize()
        t0 = time.time()
        x, its, res = cg_solve(Q._descriptor(), wl["y"], tol=1e-6, stop_mode=1, max_iter=5000, jacobi=jac)
        torch.cuda.synchronize()
        out["Qsolve_jacobi_%s" % jac] = dict(iters=its, ms=round((time.time() - t0) * 1e3, 2), resid=max(res))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
