"""Convergence of the pipelined recurrence (csrc/pcg.hip) against the Chronopoulos-Gear solver on an ill-conditioned
system: iterations and TRUE residual per tolerance (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd.graph import LaplacianData
from manifold_gp_amd.parallel import PcgPlan, RowPartition, pad_graph
from manifold_gp_amd.solvers import cg_solve
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for norm, form, nu in (("symmetric", 0, 2), ("symmetric", 2, 3), ("randomwalk", 0, 3)):
    idx, val = T(g["edge_index"].astype(np.int64)), T(g["edge_value"])
    lap = mgp.operators.GraphLaplacianOperator(val, idx, g["train_x"].shape[0], torch.tensor([[float(g["eps"])]], device=dev), norm, True)
    Q = mgp.operators.PrecisionMaternOperator(lap, nu, torch.tensor([[float(g["kappa"])]], device=dev))
    desc = Q._descriptor()
    desc = desc.with_(scale=0.7, form=2, noise=1e-2) if form == 2 else desc
    part = RowPartition(desc.n, 1)
    data = LaplacianData(pad_graph(lap.graph, part.n_pad), float(g["eps"]), True)
    sq = data.dsqrt if norm == "randomwalk" else None
    dd = desc.with_(data=data, pre=sq, post=sq)
    y = T(g["train_y"])
    yp = part.pad(y)
    for tol in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
        xs, its, _ = cg_solve(desc, y, tol=tol, stop_mode=1, max_iter=5000)
        plan = PcgPlan(dd, part, 0, tol=tol, max_iter=5000, stop_mode=1)
        x = plan.solve(yp).clone()[:desc.n]
        tr = lambda v: float((desc.apply(v) - y).norm() / y.norm())
        print("%s form %d nu %d tol %.0e | CG-CG its %4d true %.2e | pipelined its %4d status %d reported %.2e true %.2e"
              % (norm, form, nu, tol, its, tr(xs), plan.iters, plan.status, plan.resid, tr(x)))
        plan.close()
