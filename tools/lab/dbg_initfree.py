import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests/golden/dumbbell_k10_loop.npz")))
T = lambda a: torch.as_tensor(a, device=dev)
idx = T(g["edge_index"].astype(np.int64)); val = T(g["edge_value"])
n = int(g["train_x"].shape[0])
lib = _lib.lib()
for rep in range(12):
  for nu in (1, 2, 3):
    lap = mgp.operators.GraphLaplacianOperator(val, idx, n, torch.tensor([[float(g["eps"])]], device=dev), "randomwalk", bool(g["self_loops"]))
    Q = mgp.operators.PrecisionMaternOperator(lap, nu, torch.tensor([[float(g["kappa"])]], device=dev))
    desc = Q._descriptor().with_(scale=0.7, form=2, noise=1e-2)
    y = T(g["train_y"]).float().view(-1, 1).contiguous()
    y2 = torch.randn(n, 1, device=dev); z = torch.zeros(n, 1, device=dev)
    out = {}
    for mode, use_graph in ((0, True), (0, False), (1, True), (1, False)):
        lib.mgp_cg_set_init_free(mode)
        if mode == 0: 
            plan = CgPlan(desc, 1, tol=1e-6, max_iter=20000, stop_mode=1, check_every=8, use_graph=use_graph)
            for rhs in (y, y, y2, z, y, y.clone(), y): plan.solve(rhs).clone()
            plan.close(); continue
        plan = CgPlan(desc, 1, tol=1e-6, max_iter=20000, stop_mode=1, check_every=8, use_graph=use_graph)
        sols = []
        for rhs in (y, y, y2, z, y, y.clone(), y):
            x = plan.solve(rhs).clone(); sols.append((x, plan.iters, plan.status))
        out[use_graph] = sols; plan.close()
    for k in range(7):
        a, b = out[True][k], out[False][k]
        d = (a[0] != b[0]).sum().item()
        print("rep", rep, "nu", nu, "k", k, "iters", a[1], b[1], "status", a[2], b[2], "differ", d, "maxabs", float((a[0] - b[0]).abs().max()))
