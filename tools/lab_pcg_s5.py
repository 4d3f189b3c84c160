"""Partitioned pipelined CG on the 1M-node swiss roll (S5): one rank with / without refinement against cg.hip, and 8
virtual ranks (ghost rows per rank, agreement with the one-rank solve).  GPU box."""
import os, sys, time, argparse
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from manifold_gp_amd.graph import LaplacianData
from manifold_gp_amd.parallel import PcgPlan, RowPartition, pad_graph, virtual_pcg_solve
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
a = argparse.Namespace(workload="s5", nodes=n, s5_order="morton")
wl = bench.build_workload(a, dev, 0, 1)
g, desc, y = wl["graph"], wl["desc"], wl["y"]
tr = lambda v: float((desc.apply(v) - y).norm() / y.norm())
plan = CgPlan(desc, 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=3)
torch.cuda.synchronize(); t0 = time.perf_counter()
xs = plan.solve(y.view(-1, 1)).clone()[:, 0]
torch.cuda.synchronize(); print("cg.hip refine=3: its %d resid %.2e true(fp32 apply) %.2e  %.1f ms" % (plan.iters, max(plan.resid), tr(xs), (time.perf_counter() - t0) * 1e3))
for world in (1,):
    part = RowPartition(g.n, world)
    data = LaplacianData(pad_graph(g, part.n_pad), wl["lap"].data.eps, True)
    dd = desc.with_(data=data)
    for refine, chunk in ((0, 8), (0, 32), (4, 32), (4, 64)):
        p = PcgPlan(dd, part, 0, tol=1e-6, max_iter=3000, stop_mode=1, check_every=chunk, refine=refine)
        yp = part.pad(y)
        x = p.solve(yp).clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x = p.solve(yp).clone()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("pcg world 1 refine %d chunk %d: its %d status %d resid %.2e true(fp32 apply) %.2e diff vs cg %.2e  %.1f ms"
              % (refine, chunk, p.iters, p.status, p.resid, tr(x[:g.n]), float((x[:g.n] - xs).abs().max() / xs.abs().max()), dt * 1e3))
        p.close()
part1 = RowPartition(g.n, 1)
data1 = LaplacianData(pad_graph(g, part1.n_pad), wl["lap"].data.eps, True)
for refine in (0, 3):
    p = PcgPlan(desc.with_(data=data1), part1, 0, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=refine, recurrence="chronopoulos-gear")
    yp = part1.pad(y)
    x = p.solve(yp).clone()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = p.solve(yp).clone()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("partitioned Chronopoulos-Gear world 1 refine %d: its %d status %d resid %.2e true(fp32 apply) %.2e diff vs cg %.2e  %.1f ms"
          % (refine, p.iters, p.status, p.resid, tr(x[:g.n]), float((x[:g.n] - xs).abs().max() / xs.abs().max()), dt * 1e3))
    p.close()
part = RowPartition(g.n, 8)
data = LaplacianData(pad_graph(g, part.n_pad), wl["lap"].data.eps, True)
dd = desc.with_(data=data)
t0 = time.perf_counter()
x8, its, status, ghosts = virtual_pcg_solve(dd, part, part.pad(y), tol=1e-4, max_iter=800, stop_mode=1, recurrence="chronopoulos-gear")
print("8 virtual ranks (Chronopoulos-Gear): its %d status %d ghosts/rank %s (n_loc %d) true %.2e  (%.1f s incl. setup)" % (its, status, ghosts, part.n_loc, tr(x8[:g.n]), time.perf_counter() - t0))
