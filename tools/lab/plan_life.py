"""Lab: host cost of a CG plan's life on the C3 graph: create, solves 1..5 (graphs are captured at the second), close.
plan_life.py [C] [jacobi 0|1] [masked 0|1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import solvers
from manifold_gp_amd.operators._descriptor import Descriptor
C = int(sys.argv[1]) if len(sys.argv) > 1 else 12
jac = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
masked = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
data = wl["desc"].data
mask = (torch.rand(data.graph.n, device=dev) > 0.1).float()
torch.manual_seed(0)
B = torch.randn(data.graph.n, C, device=dev)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    if masked:
        desc = Descriptor(data=data, nu=2, kappa=3.0 + 0.01 * rep, pre=data.dsqrt * mask, post=data.dsqrt * mask)
    else:
        desc = Descriptor(data=data, nu=1, kappa=3.0 + 0.01 * rep)
    t = [T()]
    plan = solvers.CgPlan(desc, C, tol=1e-2, max_iter=400, stop_mode=0, jacobi=jac)
    t.append(T())
    for _ in range(6):
        plan.solve(B); t.append(T())
    its = plan.iters
    plan.close(); t.append(T())
    d = [round((b - a) * 1e3, 3) for a, b in zip(t[:-1], t[1:])]
    print(dict(C=C, jacobi=jac, masked=masked, iters=its, create_ms=d[0], solves_ms=d[1:7], close_ms=d[7]))
