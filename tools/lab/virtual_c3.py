"""Lab: the bench's strong-scaling configuration (C3, N = 60k) on ONE GPU with `world` virtual ranks (each its own row block,
ghost layers, tile view and plan; the all-gather is the identity): solution against the single-GPU solve, iterations, ghosts."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd.graph import LaplacianData
from manifold_gp_amd.parallel import RowPartition, pad_graph, virtual_pcg_solve
from manifold_gp_amd.solvers import cg_solve
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, base = wl["graph"], wl["desc"]
xs, its1, _ = cg_solve(base, wl["y"], tol=1e-6, stop_mode=1)
for world in (2, 4, 8):
    part = RowPartition(g.n, world)
    gp = pad_graph(g, part.n_pad)
    data = LaplacianData(gp, wl["eps"], True)
    pre = data.dsqrt if wl["norm"] == "randomwalk" else None
    desc = base.with_(data=data, pre=pre, post=pre)
    y = part.pad(wl["y"])
    for rec in ("pipelined", "chronopoulos-gear"):
        t0 = time.perf_counter()
        x, its, status, ghosts = virtual_pcg_solve(desc, part, y, tol=1e-6, max_iter=4000, stop_mode=1, recurrence=rec)
        torch.cuda.synchronize()
        r = base.apply(x[:g.n]) - wl["y"]
        print("world %d %-17s: status %d iterations %d (one GPU: %d), ghosts per rank %s, max |x - x1| / max |x1| = %.2e, true residual %.2e, padding rows zero %s"
              % (world, rec, status, its, its1, ghosts, float((x[:g.n] - xs.view(-1)).abs().max() / xs.abs().max()), float(r.norm() / wl["y"].norm()),
                 bool(float(x[g.n:].abs().max()) == 0.0) if x.numel() > g.n else True), flush=True)
