"""Lab: per-iteration time of a C-column CG on the C3 graph (B = tau I + L_sym, Jacobi on/off), for the update-kernel work.
cg12.py [C] [reduce_once: 1|2] [chain_min_c] [jacobi 0|1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import solvers, _lib as _mlib
if os.environ.get("MGP_LAB_LIB"):
    _mlib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
from manifold_gp_amd._lib import lib
from manifold_gp_amd.operators._descriptor import Descriptor
C = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ro = int(sys.argv[2]) if len(sys.argv) > 2 else 1
solvers.CHAIN_SOLVE_MIN_C[0] = int(sys.argv[3]) if len(sys.argv) > 3 else 8
jac = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
lib().mgp_cg_set_reduce_once(ro)
if os.environ.get("MGP_UPD_QUADS"):
    lib().mgp_cg_set_update_quads(int(os.environ["MGP_UPD_QUADS"]))
if os.environ.get("MGP_UPD_BLOCK"):
    lib().mgp_cg_set_update_block(int(os.environ["MGP_UPD_BLOCK"]))
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
data = wl["desc"].data
mask = (torch.rand(data.graph.n, device=dev) > 0.1).float()
desc = Descriptor(data=data, nu=2, kappa=3.0, pre=data.dsqrt * mask, post=data.dsqrt * mask)
torch.manual_seed(0)
B = torch.randn(data.graph.n, C, device=dev) * mask.view(-1, 1)
plan = solvers.CgPlan(desc, C, tol=1e-4, max_iter=int(os.environ.get("MGP_MAX_ITER", 400)), stop_mode=1, jacobi=jac)
for _ in range(3):
    X = plan.solve(B)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    X = plan.solve(B)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / reps
R = B - desc.apply(X)
print(dict(C=C, reduce_once=ro, chain=plan._rg is not None, jacobi=jac, iters=plan.iters, solve_ms=round(ms, 3), us_per_iter=round(ms * 1e3 / plan.iters, 2),
           true_rel=float((R.norm(dim=0) / B.norm(dim=0)).max())))
