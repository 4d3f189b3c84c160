#!/usr/bin/env python
"""bench.py -- headline benchmark of the hot path on MI355X.

Metric (BASELINE.json): "CG-solve wall-time + SpMV HBM GB/s, N=60k RMNIST graph, 1/2/4/8 GPUs".
Workload at N=1 (config C3 of SURVEY.md section 8): synthetic RMNIST-like 60 000 x 784 points
(tools/synth.py), exact k-NN graph k=50 built by the HIP kernels, random-walk Laplacian, Matern
nu=2 precision with the reference's trained hyper-parameters (models/srmnist_manifold_
semisupervised.pth -> tests/golden/hyperparameters.json, bandwidth raised to the notebooks' eps_min
rule when weights would underflow), posterior-mean system (K + s I) in precision form
A = I + s Q2, right-hand side = standardised rotation angle.

One "step" = one complete CG solve of A x = y to a relative residual of 1e-6 on device-resident
inputs.  `value` = algorithmic SpMV bytes streamed by the solves / wall time (whole job, GB/s);
`ms_per_step` = CG-solve wall time.  `roofline` times the dominant kernel (the fused C=1 SpMV)
back to back with HIP events on the launch stream; `cpu_baseline` times the reference-style torch
CPU operator (oracle/ref_torch.py, kind "port") on the host cores in the same run.

N>1 (launched by torch.distributed.run, one rank per GPU): STRONG scaling by default -- the same 60 000-node graph
(the metric's configuration), rows and vectors partitioned across ranks, partitioned pipelined CG with one grouped
RCCL all-gather per iteration (manifold_gp_amd/parallel.py, csrc/pcg.hip); `--scaling weak` keeps N x 60 000 points
with round 1's replicated-vector plan.  `--workload s5` is the 1M-node graph that is expected to scale.
"""
import argparse
import json
import os
import re
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def spmm_bytes(n, M, C=1):
    """full symmetric CSR, fp32 values, int32 col, int32 rowptr, fp32 diag,
    x read once, y written once."""
    return 8 * (2 * M) + 4 * (n + 1) + 4 * n + 8 * n * C


def source_hash():
    """sha256 over the kernel sources and headers of this tree (first 16 hex digits): what a committed profile summary must
    carry for the bench line to quote it (tools/summarize_profile.py writes it)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "manifold_gp_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "manifold_gp_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(args, dev, rank, world, shard_knn=False, scale_nodes=True):
    """scale_nodes: weak scaling (world x the per-GPU node count); False: the same graph on every world size."""
    import manifold_gp_amd as mgp
    from tools import synth
    t0 = time.time()
    if not scale_nodes:
        nodes_mult = 1
    else:
        nodes_mult = world
    if args.workload == "c3":
        bases = args.nodes // 100 if args.nodes else 600
        x_t, y_t = synth.rmnist_like(bases * nodes_mult, 100, seed=1337, device=dev)
        k, nu, norm = 50, 2, "randomwalk"
        with open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")) as fh:
            hp = json.load(fh)["srmnist_manifold_semisupervised"]
        name = "C3 RMNIST-like N=%d d=784 k=50 nu=2 randomwalk" % x_t.shape[0]
    elif args.workload == "s5":
        n = args.nodes or 1000000
        x_np, y_np = synth.swiss_roll(n * nodes_mult, order=args.s5_order)
        x_t, y_t = torch.from_numpy(x_np).to(dev), torch.from_numpy(y_np).to(dev)
        k, nu, norm = 64, 2, "symmetric"
        hp = dict(graphbandwidth=0.0, lengthscale=1.0, outputscale=1.0, noise=0.01, eps_scale=3.0)
        name = "S5 swiss-roll N=%d d=3 k=64 nu=2 symmetric (%s order)" % (x_np.shape[0], args.s5_order)
    else:
        raise SystemExit("unknown workload " + args.workload)
    t_data = time.time() - t0
    x, y = x_t, y_t
    if shard_knn and args.workload == "s5" and args.s5_order == "random":
        # the row partition of the multi-GPU solvers cuts the node order into contiguous blocks: on an order without locality
        # every block's ghost layer would be the whole graph.  The single-GPU solvers relabel internally (solvers.CgPlan); the
        # partitioned job relabels the PROBLEM: points and targets permuted into the library's locality order (the Z-curve,
        # mgp_morton_order) before the graph is built -- the solution is then reported in that order
        from manifold_gp_amd.graph import morton_order
        order = morton_order(x).long()
        x, y = x.index_select(0, order).contiguous(), y.index_select(0, order).contiguous()
        name += " -> relabelled by the library's Z-curve for the row partition"
    torch.cuda.synchronize()
    t0 = time.time()
    knn = mgp.utils.NearestNeighbors(x)
    if shard_knn and world > 1:
        # k-NN queries sharded by rows (X replicated, no collective in the search); the lists are
        # all-gathered once and every rank symmetrises the full graph (setup, untimed)
        import torch.distributed as dist
        from manifold_gp_amd.graph import KnnGraph
        n = x.shape[0]
        per = -(-n // world)
        r0, r1 = rank * per, min(n, (rank + 1) * per)
        Dl, Il = knn.search(x[r0:r1], k)
        Dp = torch.zeros(per, k, device=dev)
        Ip = torch.zeros(per, k, dtype=torch.int32, device=dev)
        Dp[: r1 - r0] = Dl
        Ip[: r1 - r0] = Il.to(torch.int32)
        Dg = torch.empty(world * per, k, device=dev)
        Ig = torch.empty(world * per, k, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(Dg, Dp)
        dist.all_gather_into_tensor(Ig, Ip)
        knn.knn_graph = KnnGraph.from_knn(Dg[:n].contiguous(), Ig[:n].contiguous())
        idx, val = knn.knn_graph.edge_index, knn.knn_graph.edge_value
    else:
        idx, val = knn.graph(k)
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    graph = knn.knn_graph
    # eps: trained value, floored by the notebooks' eps_min rule on the 1-NN distances
    D1, _ = knn.search(x[: min(20000, x.shape[0])], 2)
    eps, eps_min = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
    eps = max(eps, hp.get("eps_scale", 1.0) * eps_min)
    log("[bench] data %.1fs, k-NN+graph %.2fs (knn stats %s), N=%d M=%d nnz_padded=%d eps=%.4f (eps_min %.4f)"
        % (t_data, t_graph, knn.last_stats, graph.n, graph.M, graph.nnz, eps, eps_min))
    lap = mgp.operators.GraphLaplacianOperator(val, idx, graph.n, torch.tensor([[eps]], device=dev), norm, graph=graph)
    Q = mgp.operators.PrecisionMaternOperator(lap, nu, torch.tensor([[hp["lengthscale"]]], device=dev))
    desc = Q._descriptor().with_(scale=hp["outputscale"], form=2, noise=hp["noise"])
    return dict(mgp=mgp, graph=graph, lap=lap, desc=desc, y=y, nu=nu, name=name, hp=hp, eps=eps, norm=norm,
                t_graph=t_graph, knn=knn, x=x, k=k)


def time_spmv_kernel(wl, reps=200, as_given=False):
    """Average duration of the dominant kernel (fused C=1 SpMV, L_sym) launched back to back from C
    (mgp_spmm_repeat), HIP events on the launch stream (= torch's current stream).  The matrix is the one the iterative
    solvers run on: for a graph handed over without locality that is P L P^T in the library's locality order
    (solvers.CgPlan); as_given=True times the product in the caller's order instead (what one `lap.matmul(v)` launches)."""
    import ctypes
    from manifold_gp_amd import _lib
    lap = wl["lap"]
    g = wl["graph"]
    v = torch.rand(lap.shape[0], 1, device=wl["y"].device)
    out = torch.empty_like(v)
    lib = _lib.lib()
    lib.mgp_spmm_set_group_hint(g.spmv_lanes)
    rel = None if as_given else lap.data.relabelled()
    csr = (rel or lap.data).csr()
    st = _lib.stream()
    ms = ctypes.c_float(0.0)
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), 20, None, st), "mgp_spmm_repeat")
    best = 1e30
    for _ in range(3):
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(v), 1, _lib.ptr(out), reps, ctypes.byref(ms), st),
                   "mgp_spmm_repeat")
        best = min(best, ms.value)
    return best / reps * 1e-3   # seconds per launch


def time_spmv_in_solve(wl, args, refine, y, solves=40):
    """Average duration of the same kernel INSIDE the CG solves, measured live: an eager twin of the timed plan
    (use_graph = 0: the same kernels, launched one by one) runs `solves` solves while every launch of the tile kernel
    carries its own start / stop event pair (mgp_spmm_timing_begin / _end: hipExtLaunchKernelGGL, i.e. the dispatch's
    begin / end timestamps -- what rocprofv3 --kernel-trace reports).  This is the figure `roofline.frac` uses: the
    back-to-back replay above leaves out what a solve does to the kernel (a different kernel ran just before it, the
    first SpMV of a solve scales its input)."""
    import ctypes
    from manifold_gp_amd import _lib
    from manifold_gp_amd.solvers import CgPlan
    lib = _lib.lib()
    plan = CgPlan(wl["desc"], 1, tol=args.tol, max_iter=5000, stop_mode=1, check_every=8, use_graph=False, refine=refine)
    for _ in range(5):
        plan.solve(y, copy=False)
    torch.cuda.synchronize()
    _lib.check(lib.mgp_spmm_timing_begin(16384), "mgp_spmm_timing_begin")
    for _ in range(solves):
        plan.solve(y, copy=False)
    torch.cuda.synchronize()
    ms, cnt = ctypes.c_float(0.0), ctypes.c_int(0)
    _lib.check(lib.mgp_spmm_timing_end(ctypes.byref(ms), ctypes.byref(cnt)), "mgp_spmm_timing_end")
    plan.close()
    if cnt.value == 0:
        return None, 0
    return ms.value / cnt.value * 1e-3, cnt.value      # seconds per launch, launches timed


def hbm_streaming_roofline(dev, order, reps=100, knn_stage=False, nodes=0):
    """Secondary roofline on a working set that does NOT fit the 256 MiB Infinity Cache: config C5's graph
    (1M-point swiss roll, k = 64, ~0.55 GB of CSR per SpMV).  One graph build + `reps` back-to-back launches of the
    same fused C = 1 SpMV kernel, HIP events on the launch stream."""
    a = argparse.Namespace(workload="s5", nodes=nodes, s5_order=order)
    wl = build_workload(a, dev, 0, 1)
    g = wl["graph"]
    t_k = time_spmv_kernel(wl, reps=reps)
    B = spmm_bytes(g.n, g.M)
    ordered = g.tiles is not None and g.tiles.get("rowid") is not None
    out = dict(workload=wl["name"], nodes=g.n, edges=g.M, bytes_per_launch=B, avg_launch_us=round(t_k * 1e6, 2),
               achieved=round(B / t_k / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(B / t_k / 1e9 / HBM_PEAK_GBS, 4),
               measured="live_back_to_back (this run: %d graph-replayed launches, HIP events)" % reps,
               tile_order=("locality order (Z-curve); the solvers iterate on the relabelled matrix, vectors permuted in / out once "
                           "per solve" if ordered else "input order"),
               entries_per_dictionary_column=round(g.tiles["reuse"], 2) if g.tiles is not None else None,
               knn_graph_build_s=round(wl["t_graph"], 3))
    if ordered:
        t_g = time_spmv_kernel(wl, reps=reps, as_given=True)
        out["single_product_in_caller_order_us"] = round(t_g * 1e6, 2)
    if knn_stage:
        from tools import bench_stages
        out["knn"] = bench_stages.knn_stage(wl["x"], wl["k"], knn=wl["knn"])
    # the C5 posterior-mean solve (I + noise s Q) x = y, tol 1e-6, fp64-residual refinement: caller-order y in, caller-order
    # x out -- what a user of the path times; the two input orders must cost the same
    from manifold_gp_amd.solvers import CgPlan
    plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=3)
    y = wl["y"].view(-1, 1).contiguous()
    plan.solve(y, copy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.solve(y, copy=False)
    torch.cuda.synchronize()
    out["cg_solve"] = dict(ms=round((time.perf_counter() - t0) * 1e3, 2), iterations=plan.iters, status=plan.status,
                           true_rel_residual=float(max(plan.resid)))
    plan.close()
    if knn_stage:        # (the Morton-order call only) none / jacobi / Chebyshev-polynomial preconditioners on the same system
        from tools import bench_stages
        out["cg_solve"]["preconditioner"] = bench_stages.preconditioner_block(wl["desc"], wl["y"], 1e-6, 3)
    del wl
    torch.cuda.empty_cache()
    return out


def multi_rhs_solve(wl, columns=100, tol=1e-2):
    """The largest CG workload of training (precision_matern_operator.py:45-53, `_average_variance`): `columns`
    random one-hot right-hand sides on Q itself, linear_cg's stopping rule at the notebooks' cg_tolerance -- through
    solvers.cg_solve, i.e. what `Q.solve` runs: since round 2 the chain Q = (tau I + L)^nu x D is solved factor by factor
    (nu CG solves with tau I + L, one SpMM per iteration, true residual checked); the CG on the whole chain is timed next
    to it."""
    from manifold_gp_amd.solvers import cg_solve
    g = wl["graph"]
    dev = wl["y"].device
    desc = wl["desc"].with_(scale=1.0, form=0, noise=0.0)
    torch.manual_seed(1337)
    idx = torch.randint(0, g.n - 1, (1, columns), device=dev)
    B = torch.zeros(g.n, columns, device=dev).scatter_(0, idx, 1.0)
    Bm = spmm_bytes(g.n, g.M, columns)

    def run(**kw):
        for _ in range(2):
            X, its, res = cg_solve(desc, B, tol=tol, max_iter=1000, stop_mode=0, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            X, its, res = cg_solve(desc, B, tol=tol, max_iter=1000, stop_mode=0, **kw)
        torch.cuda.synchronize()
        r = desc.apply(X) - B
        return X, its, (time.perf_counter() - t0) / reps, float((r.norm(dim=0) / B.norm(dim=0)).mean())

    X, its, dt, true_res = run()
    _, its_w, dt_w, true_w = run(factorise=False)
    nu = wl["nu"]
    return dict(columns=columns, operator="Q = D^1/2 (2 nu / kappa^2 I + L_sym)^nu D^1/2", stop="linear_cg rule, tol %g" % tol,
                method="nu sequential CG solves with 2 nu / kappa^2 I + L_sym (one SpMM per iteration) + true-residual check",
                iterations=its, solve_ms=round(dt * 1e3, 3), spmm_launches=its + 2 * nu * 1, spmm_bytes_per_launch=Bm,
                spmm_gbs=round(Bm * (its + 2 * nu) / dt / 1e9, 1), true_mean_rel_residual=true_res,
                average_variance=float((X * B).sum() / columns),
                whole_chain_cg=dict(iterations=its_w, solve_ms=round(dt_w * 1e3, 3), spmm_launches=(its_w + 1) * nu,
                                    true_mean_rel_residual=true_w))


def cpu_baseline(wl, gpu_iters):
    """Reference-style CPU path (oracle/ref_torch.py) on the host cores: SpMV rate and the same CG
    solve.  Bounded: 20 SpMVs + one CG solve (a few seconds at N=60k)."""
    from oracle.ref_torch import TorchCooLaplacian, TorchPrecision, torch_cg
    g, lap = wl["graph"], wl["lap"]
    # the GPU box exposes every host core (256) but one GPU's share is 16; over-subscribing the torch
    # CPU ops (index_add_) with 256 threads is ~100x slower than 16
    cores = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(cores)
    idx = g.edge_index.cpu()
    triu = lap.laplacian_triu.cpu()
    diag = lap.laplacian_diag.cpu()
    deg = lap.degree_mat.cpu()
    ref = TorchCooLaplacian(idx, triu, diag, deg, wl["norm"])
    torch.manual_seed(1337)
    v = torch.rand(g.n)                               # bench_sparse_laplacian.py:63-64
    t0 = time.perf_counter()
    ref.matmul(v)
    first = time.perf_counter() - t0                  # the reference's single-shot time.time() style
    for _ in range(3):
        ref.matmul(v)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        ref.matmul(v)
        ts.append(time.perf_counter() - t0)
    t_mv = statistics.median(ts)
    prec = TorchPrecision(ref, wl["nu"], wl["hp"]["lengthscale"], wl["hp"]["outputscale"], wl["hp"]["noise"])
    y = wl["y"].cpu()
    t0 = time.perf_counter()
    xs, its = torch_cg(prec.posterior_system, y, 1e-6, 1000)
    t_cg = time.perf_counter() - t0
    B = spmm_bytes(g.n, g.M)
    spmvs = its * wl["nu"]                            # torch_cg: one operator apply per iteration, x0 = 0
    return dict(value=round(B * spmvs / t_cg / 1e9, 3), unit="GB/s", cores=cores, kind="port",
                sample="full C3 workload: 1 CG solve (%d iterations, %d SpMV) + 20 SpMV repeats on %d threads"
                       % (its, spmvs, torch.get_num_threads()),
                spmv_ms=round(t_mv * 1e3, 3), spmv_first_call_ms=round(first * 1e3, 3),
                spmv_gbs=round(B / t_mv / 1e9, 3), cg_solve_ms=round(t_cg * 1e3, 2), cg_iters=its), xs


class _QuietStdout:
    """Exactly ONE JSON line may reach stdout: libraries (RCCL prints a version banner at communicator
    creation) write to fd 1, so fd 1 is pointed at stderr until the result is ready."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def emit(self, text):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        print(text, flush=True)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    with _QuietStdout() as quiet:
        _main(quiet)


def _main(quiet):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "s5"])
    ap.add_argument("--nodes", type=int, default=0, help="override nodes per GPU (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the cg_multi_rhs / roofline_hbm / stages blocks")
    ap.add_argument("--force-extras", action="store_true",
                    help="with --nodes: run the extra blocks all the same, at reduced sizes (tests of the line's contract)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the SAME graph on every world size (the metric's N = 60k), weak = N x nodes")
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--s5-order", default="morton", choices=["random", "morton"])
    ap.add_argument("--refine", type=int, default=-1, help="CG refinement rounds (-1: 0 for c3, 3 for s5)")
    ap.add_argument("--classic-cg", action="store_true",
                    help="CG on A even where the complex-shift factorisation applies (the in-solve SpMV profile of --workload s5)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        # torch-side collectives (setup all-gathers, barriers) give up after 10 minutes instead of hanging on a dead peer
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=600))

    if world > 1 or os.environ.get("MGP_FORCE_DIST") == "1":
        from manifold_gp_amd import parallel
        try:
            return parallel.bench_distributed(args, dev, rank, world, build_workload, spmm_bytes, HBM_PEAK_GBS, quiet.emit,
                                              cpu_baseline=None if args.no_cpu_baseline else cpu_baseline)
        except BaseException as e:      # a failed or timed-out collective (MGP_ERR_TIMEOUT) must end THIS process with a
            # non-zero code at once -- no communicator teardown (it can hang on a dead peer), never a re-exec
            import traceback
            traceback.print_exc()
            log("[bench] rank %d: distributed run failed (%r): exiting 3" % (rank, e))
            sys.stderr.flush()
            os._exit(3)

    wl = build_workload(args, dev, rank, world)
    from manifold_gp_amd.solvers import CgPlan
    if args.classic_cg:
        from manifold_gp_amd import _lib as _l
        _l.lib().mgp_cg_set_complex_shift(0)
    g = wl["graph"]
    refine = args.refine if args.refine >= 0 else (0 if args.workload == "c3" else 3)
    plan = CgPlan(wl["desc"], 1, tol=args.tol, max_iter=5000, stop_mode=1, check_every=8, refine=refine)
    y = wl["y"].view(-1, 1).contiguous()
    # a generation-2 garbage collection of the interpreter (tens of ms with the workload's objects alive) lands
    # inside the timed loop once it is longer than ~60 solves: collect now, keep the collector out of the loop
    import gc
    gc.collect()
    gc.disable()
    # steady state: the timed region is 20 solves of ~60 us by default -- about a millisecond, right behind seconds of
    # host work (graph build, garbage collection) during which the GPU clocks have dropped; a fraction of a second of
    # untimed solves first (tools/lab/ab_warm.py: 59-60 us per solve straight after an idle gap, 57 us after 3 000
    # solves on the same box), then the W warm-up steps of the contract, then exactly K timed steps
    preheat = 3000 if args.workload == "c3" else 0
    for _ in range(preheat):
        plan.solve(y, copy=False)
    for _ in range(args.warmup):
        out = plan.solve(y, copy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = applies = 0
    for _ in range(args.steps):
        out = plan.solve(y, copy=False)      # solution stays in the plan's device buffer
        iters += plan.iters
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    applies = plan.applies * args.steps       # the same right-hand side every step: every solve runs the same launches
    its = iters / args.steps
    B = spmm_bytes(g.n, g.M)
    # SpMVs that actually ran: nu per operator apply; the plan reports its applies (a solve of k iterations
    # runs k of them once its graph ends in the decision-only launch, k + 1 before).  With refinement rounds the
    # fp32 applies of every round are iterations + 1.
    spmvs_per_solve = (applies / args.steps if refine == 0 else its + 1) * wl["nu"]
    if plan.complex_shift:
        # the plan solved through the complex factorisation (--workload s5): `its` products with B = tau I + L_sym in four
        # columns (re, im, re, im) instead of (its + 1) * nu single-column products with L
        spmvs_per_solve = its
        B = spmm_bytes(g.n, g.M, 4)
    value = B * spmvs_per_solve * args.steps / dt / 1e9
    B = spmm_bytes(g.n, g.M)
    resid = max(plan.resid)
    # residual re-check with one explicit operator apply
    r = wl["desc"].apply(out) - y
    true_res = float(r.norm() / y.norm())

    # the same K steps with the right-hand side alternating between two buffers: every solve re-points the graph's root
    # node (hipGraphExecKernelNodeSetParams), which the loop above -- one buffer -- never pays
    y2 = y.clone()
    for _ in range(max(args.warmup, 2)):
        plan.solve(y2, copy=False)
        plan.solve(y, copy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        plan.solve(y2 if i & 1 else y, copy=False)
    torch.cuda.synchronize()
    dt_alt = time.perf_counter() - t0
    gc.enable()

    # Three figures for the same kernel.  Live, this run: (1) back to back = 200 graph-replayed launches with nothing in
    # between (the kernel alone; optimistic for a solve); (2) eager in-solve = per-launch begin / end timestamps inside an
    # eager twin of the timed plan (pessimistic: eager launches carry system-scope fences and idle gaps a captured
    # solve does not).  From the committed rocprofv3 --kernel-trace of this same command: (3) the mean over the solves'
    # own graph launches, which no HIP event can bracket (events of captured launches are not usable: tools/lab/
    # timing_in_graph.py).  (1) <= (3) <= (2) on every box so far; `frac` quotes (3) when the profile is there -- the
    # line then reproduces from profiles/ -- and (2) otherwise.
    t_b2b = time_spmv_kernel(wl)
    t_in, n_in = time_spmv_in_solve(wl, args, refine, y)
    # (3) is quoted only from a profile taken on THIS source tree: tools/summarize_profile.py stores a hash of csrc/ + include/
    # in the summary; a summary whose hash differs from the tree's, or whose mean is far from this run's back-to-back
    # figure (band below), is stale -- `frac` then quotes the live in-solve figure and the line says so
    t_prof, prof_src, prof_hash, prof_file = None, None, None, None
    suffix = "_pmc_traffic.json" if args.workload == "c3" else "_s5_pmc_traffic.json"
    cands = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+" + suffix, f)), reverse=True) \
        if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    for cand in cands:
        if args.nodes:
            break
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            ns = pj.get("spmv_kernel_trace_mean_ns") or pj.get("spmv_kernel (no pre-scaling)", {}).get("kernel_trace_median_ns")
            if ns:
                t_prof, prof_file, prof_hash = ns * 1e-9, cand, pj.get("source_hash")
                prof_src = "profiles/%s (%s of %s launches, rocprofv3 --kernel-trace of this command)" % (
                    cand, "mean" if pj.get("spmv_kernel_trace_mean_ns") else "median",
                    pj.get("spmv_kernel_trace_launches") or pj.get("spmv_kernel (no pre-scaling)", {}).get("kernel_trace_launches"))
                break
        except Exception:
            pass
    tree_hash = source_hash()
    # (band: the in-graph mean sits ABOVE this run's back-to-back figure -- (1) <= (3) -- by 3-8 % on one box, and the boxes of the
    # pool differ by up to 14 % in this kernel (5.3-6.2 us in-graph): a profile of this tree -- the hash is the staleness test --
    # within -15 % (a profile from a fast box read on a slow one) / +25 % of the live figure is current)
    profile_age_ok = bool(t_prof) and prof_hash == tree_hash and -0.15 * t_b2b <= t_prof - t_b2b <= 0.25 * t_b2b
    t_k = (t_prof if profile_age_ok else None) or t_in or t_b2b
    def fig(t, note):
        return dict(avg_launch_us=round(t * 1e6, 2), achieved=round(B / t / 1e9, 1), frac=round(B / t / 1e9 / HBM_PEAK_GBS, 4), note=note)
    roof = dict(bound="hbm", achieved=round(B / t_k / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(B / t_k / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                measured=("in_graph_profile: " + prof_src) if profile_age_ok else ("live_eager_in_solve" if t_in else "live_back_to_back"),
                profile_age_ok=profile_age_ok, source_hash=tree_hash, profile_source_hash=prof_hash,
                live_back_to_back=fig(t_b2b, "this run: 200 graph-replayed launches of the kernel with nothing in between, HIP events"),
                live_eager_in_solve=fig(t_in, "this run: begin / end timestamps of %d launches inside 40 eager CG solves "
                                              "(hipExtLaunchKernelGGL event pairs)" % n_in) if t_in else None,
                in_graph_profile=fig(t_prof, prof_src) if t_prof else None,
                kernel=("spmv_tile_kernel<false,%d> (fused CSR SpMV, C=1, %d-row tiles, x dictionary in LDS)"
                        % (4 * g.tiles["rows"], g.tiles["rows"])) if g.tiles is not None else
                       ("spmv_kernel<%d,1,false,false,256> (fused CSR SpMV, C=1, %d lanes per row)" % (g.spmv_lanes, g.spmv_lanes)),
                bytes_per_launch=B, avg_launch_us=round(t_k * 1e6, 2),
                note=("working set %.0f MB is Infinity-Cache resident: the 8 TB/s HBM peak is NOMINAL for this launch (counter "
                      "traffic is served by the cache; the kernel is latency-bound, docs/measurement.md) -- the HBM figure is "
                      "roofline_hbm.frac (1M-node graph, streams from HBM)"
                      if B < 2.5e8 else "working set %.0f MB streams from HBM") % (B / 1e6))
    # HBM bytes per launch from the PMC passes of the same command (tools/profile.sh -> tools/summarize_profile.py):
    # FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE.  NOT a run-time counter: read from the committed PMC summary, with
    # the same staleness flag as above
    if prof_file:
        try:
            roof["traffic"] = json.load(open(os.path.join(ROOT, "profiles", prof_file))).get("spmv_hbm_bytes_per_launch")
            roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, " \
                                     "2 x FETCH_SIZE + WRITE_SIZE; not measured in this run; profile of this source tree: %s)" \
                                     % (prof_file, prof_hash == tree_hash)
        except Exception:
            pass
    line = dict(metric="CG-solve wall-time + SpMV HBM GB/s, N=60k RMNIST graph", value=round(value, 2),
                unit="GB/s (algorithmic SpMV bytes inside the CG solve)", n_gpus=1, steps=args.steps,
                warmup=args.warmup, preheat_solves=preheat, ms_per_step=round(dt / args.steps * 1e3, 4),
                ms_per_step_alternating_rhs=round(dt_alt / args.steps * 1e3, 4), higher_is_better=True,
                scaling=args.scaling, vs_baseline=None, dtype="f32", data="synthetic",
                config=dict(workload=wl["name"], nodes=g.n, edges=g.M, nnz_padded=g.nnz, rhs_columns=1,
                            system="A = I + noise*outputscale*Q, Q=(2nu/kappa^2 I + L)^nu x D",
                            solver=("COCG on the complex factor I + i sigma B (A = I + c B^2): one 4-column SpMM per iteration"
                                    if plan.complex_shift else "CG on A (Chronopoulos-Gear, one hipGraph per solve)"),
                            cg_tol=args.tol, cg_iters=its, cg_rel_residual=resid, cg_true_residual_fp32_apply=true_res,
                            spmv_per_solve=spmvs_per_solve, eps=wl["eps"], knn_graph_build_s=round(wl["t_graph"], 3)),
                cg_solve_ms=round(dt / args.steps * 1e3, 4), roofline=roof)
    if args.workload == "c3" and not args.no_extras and (not args.nodes or args.force_extras):
        small = 100000 if args.nodes else 0          # (--force-extras: the blocks at sizes a test can afford)
        line["cg_multi_rhs"] = multi_rhs_solve(wl)
        # the C3 `Q` solve (precision_matern_operator.py:53 on Q itself, one column, tol 1e-6, CG on the whole chain)
        from tools import bench_stages as _bs
        line["cg_multi_rhs"]["q_solve_preconditioner"] = _bs.preconditioner_block(
            wl["desc"].with_(scale=1.0, form=0, noise=0.0), wl["y"], 1e-6, 0, max_iter=2000)
        hb = hbm_streaming_roofline(dev, "morton", knn_stage=True, nodes=small)
        hb["random_order_input"] = {k: v for k, v in hbm_streaming_roofline(dev, "random", nodes=small).items()
                                    if k in ("avg_launch_us", "achieved", "frac", "tile_order", "entries_per_dictionary_column",
                                             "single_product_in_caller_order_us", "cg_solve")}
        hb["traffic"], hb["traffic_source"] = None, None
        for cand in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_s5_pmc_traffic.json", f)),
                           reverse=True):
            pj = json.load(open(os.path.join(ROOT, "profiles", cand)))
            hb["traffic"] = pj.get("spmv_hbm_bytes_per_launch")
            hb["traffic_source"] = "profiles/%s (bench.py --workload s5 under rocprofv3 --pmc; profile of this source tree: %s)" \
                                   % (cand, pj.get("source_hash") == tree_hash)
            # the same kernel in the committed kernel trace of `bench.py --workload s5` (another box of the pool, launches inside
            # the solves' graphs): both figures in the line, as `roofline` does for the C3 graph
            ns = pj.get("spmv_kernel_trace_mean_ns")
            if ns:
                Bh = hb["bytes_per_launch"]
                hb["live_back_to_back"] = dict(avg_launch_us=hb["avg_launch_us"], achieved=hb["achieved"], frac=hb["frac"],
                                               note="this run, this box: back-to-back graph replays, HIP events")
                hb["in_graph_profile"] = dict(avg_launch_us=round(ns * 1e-3, 2), achieved=round(Bh / ns, 1),
                                              frac=round(Bh / ns / HBM_PEAK_GBS, 4),
                                              note="profiles/%s: mean of %s launches inside the S5 solves, rocprofv3 --kernel-trace, "
                                                   "the box the profile was taken on; profile of this source tree: %s"
                                                   % (cand, pj.get("spmv_kernel_trace_launches"), pj.get("source_hash") == tree_hash))
            break
        line["roofline_hbm"] = hb
        from tools import bench_stages
        log("[bench] stages ...")
        st = dict(workload=wl["name"])
        st["knn"] = bench_stages.knn_stage(wl["x"], wl["k"], knn=wl["knn"])
        st.update(bench_stages.graph_laplacian_stage(wl["knn"], wl["x"], wl["k"], wl["eps"]))
        kern = wl["mgp"].kernels.RiemannMaternKernel(nu=wl["nu"], x=wl["x"], nearest_neighbors=wl["k"], laplacian_normalization=wl["norm"],
                                                     num_modes=100, bump_scale=3.0, bump_decay=0.01).to(dev)
        st.update(bench_stages.spectral_stage(kern, wl["x"], wl["eps"], wl["hp"]["lengthscale"], spmm_bytes))
        del kern
        st["train_epoch_supervised"] = bench_stages.training_stage(wl["x"], wl["y"], wl["hp"], dev, semisup=False, epochs=4)
        st["train_epoch_semisupervised"] = bench_stages.training_stage(wl["x"], wl["y"], wl["hp"], dev, semisup=True, epochs=3)
        for key, rx in (("train_epoch_supervised", r"r\d+_training_supervised\.json"),
                        ("train_epoch_semisupervised", r"r\d+_training_semisupervised\.json")):
            ps = bench_stages.profile_share(rx)
            if ps is not None:
                ps["profile_of_this_source_tree"] = ps.get("source_hash") == tree_hash
                st[key]["kernel_time_profile"] = ps
                # kernel-time share of the epoch: the profile's kernel time (rocprofv3 does not stretch kernel durations) over THIS
                # run's epoch time; only quoted while the profile is of this source tree
                if ps["profile_of_this_source_tree"] and ps.get("kernel_ms_per_epoch") and st[key].get("epoch_ms"):
                    st[key]["kernel_share"] = round(min(1.0, ps["kernel_ms_per_epoch"] / st[key]["epoch_ms"]), 3)
        line["stages"] = st
        log("[bench] reference shape ...")
        line["reference_bench"] = bench_stages.reference_bench(dev)
        log("[bench] manifold_784 ...")
        line["manifold_784"] = bench_stages.manifold784_block(dev, spmm_bytes, n_all=(args.nodes + 600) if args.nodes else 60600)
        torch.cuda.empty_cache()
    if not args.no_cpu_baseline:
        cb, xs = cpu_baseline(wl, its)
        rb = line.get("reference_bench")
        if rb:       # the reference benchmark's three legs on the host cores (benchmark/bench_sparse_laplacian.py:15-34 at its shape)
            cb["reference_shape"] = dict(shape=rb["shape"], mv_ms=rb.get("cpu_matvec_ms"), mv_first_call_ms=rb.get("cpu_matvec_first_call_ms"),
                                         grad_ms=rb.get("cpu_grad_backward_ms"), eigen_ms=rb.get("cpu_dense_symeig_ms"),
                                         threads=rb.get("cpu_threads"), kind="port")
        line["cpu_baseline"] = cb
        err = float((out.view(-1).cpu() - xs).abs().max() / xs.abs().max())
        line["config"]["max_rel_diff_vs_cpu_solution"] = err
    plan.close()
    quiet.emit(json.dumps(line))


if __name__ == "__main__":
    main()
