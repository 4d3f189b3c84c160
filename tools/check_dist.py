#!/usr/bin/env python
"""Multi-GPU check of the partitioned solvers against the single-GPU solve (one process per GPU over RCCL):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/check_dist.py

Every rank builds the same graph (the golden dumbbell, k = 50), solves A x = y with
  * PcgPlan (rows AND vectors partitioned; pipelined and Chronopoulos-Gear recurrences),
  * DistCgPlan (rows partitioned, vectors replicated; round 1),
  * solve_columns_sharded (right-hand sides sharded, no data-path collective),
and compares with cg_solve on its own GPU.  Exit code 0 and "OK" on rank 0 when everything agrees.
Run by tests/test_gpu_multi.py when the box has two GPUs or more; has NOT run on hardware yet (the build box has one GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import manifold_gp_amd as mgp
    from manifold_gp_amd.graph import LaplacianData
    from manifold_gp_amd.parallel import (DistCgPlan, PcgPlan, RowPartition, init_comm, pad_graph, solve_columns_sharded)
    from manifold_gp_amd.solvers import cg_solve
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k50_noloop.npz")))
    T = lambda a: torch.as_tensor(a, device=dev)                                  # noqa: E731
    idx, val = T(g["edge_index"].astype(np.int64)), T(g["edge_value"])
    n = int(g["train_x"].shape[0])
    eps = torch.tensor([[float(g["eps"])]], device=dev)
    fails = []
    for norm in ("randomwalk", "symmetric"):
        lap = mgp.operators.GraphLaplacianOperator(val, idx, n, eps, norm, bool(g["self_loops"]))
        Q = mgp.operators.PrecisionMaternOperator(lap, 2, torch.tensor([[float(g["kappa"])]], device=dev))
        desc = Q._descriptor().with_(scale=0.7, form=2, noise=1e-2)
        part = RowPartition(desc.n, world)
        data = LaplacianData(pad_graph(lap.graph, part.n_pad), float(g["eps"]), bool(g["self_loops"]))
        sq = data.dsqrt if norm == "randomwalk" else None
        dd = desc.with_(data=data, pre=sq, post=sq)
        yv = T(g["train_y"]).float()
        y = part.pad(yv)
        xs, its, _ = cg_solve(desc, yv, tol=1e-6, stop_mode=1, max_iter=20000)
        scale = float(xs.abs().max())
        comm = init_comm(rank, world)
        r0, r1 = part.range(rank)
        for rec in ("pipelined", "chronopoulos-gear"):
            plan = PcgPlan(dd, part, rank, comm=comm, tol=1e-6, max_iter=20000, stop_mode=1, recurrence=rec)
            for _ in range(3):                       # the second solve captures the iteration graph
                x_loc = plan.solve(y).clone()
            xg = torch.zeros(part.n_pad, device=dev)
            xg[r0:r1] = x_loc
            dist.all_reduce(xg)
            err = float((xg[:n] - xs.view(-1)).abs().max()) / scale
            res = float((desc.apply(xg[:n].view(-1, 1)).view(-1) - yv).norm() / yv.norm())
            if not (plan.status == 1 and err < 5e-4 and res < 2e-5):
                fails.append(("PcgPlan", norm, rec, plan.status, plan.iters, its, err, res))
            plan.close()
        plan = DistCgPlan(dd, part, rank, comm, C=1, tol=1e-6, max_iter=20000, stop_mode=1)
        for _ in range(2):
            xr = plan.solve(y.view(-1, 1).contiguous()).clone()
        err = float((xr.view(-1)[:n] - xs.view(-1)).abs().max()) / scale
        if not err < 5e-4:
            fails.append(("DistCgPlan", norm, err))
        plan.close()
        B = T(g["probes"]).float()
        with mgp.settings.cg_tolerance(1e-6), mgp.settings.cg_stop_mode(1):
            X, _ = solve_columns_sharded(desc, B, rank, world)
            Xr, _, _ = cg_solve(desc, B)
        # (each rank solves its columns as a narrower block: other kernels, the same systems)
        cerr = float((X - Xr).abs().max()) / float(Xr.abs().max())
        if not cerr < 5e-4:
            fails.append(("solve_columns_sharded", norm, cerr))
    bad = torch.tensor([len(fails)], device=dev)
    dist.all_reduce(bad)
    if fails:
        print("rank %d FAILED: %s" % (rank, fails), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if int(bad.item()) != 0:
        sys.exit(1)
    if rank == 0:
        print("OK: %d ranks" % world, flush=True)


if __name__ == "__main__":
    main()
