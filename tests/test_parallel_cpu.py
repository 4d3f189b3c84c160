"""CPU tests of the multi-GPU host logic: the row partition, graph padding / slicing, and the
distributed CG algorithm run under torch.distributed `gloo` with world_size 2 (the oracle is the
local operator), compared with the single-process fp64 solve."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_partition_arithmetic():
    from manifold_gp_amd.parallel import RowPartition
    p = RowPartition(1546, 2)
    assert p.n_loc % 64 == 0 and p.n_pad == 2 * p.n_loc and p.n_pad >= 1546
    assert p.range(0) == (0, p.n_loc) and p.range(1) == (p.n_loc, 2 * p.n_loc)
    assert p.owned(1) == (p.n_loc, 1546)
    v = torch.arange(1546.0)
    vp = p.pad(v)
    assert vp.shape[0] == p.n_pad and torch.equal(p.unpad(vp), v) and float(vp[1546:].abs().sum()) == 0
    q = RowPartition(60000 * 8, 8)
    assert q.n_loc == 60032 - 32 or q.n_loc % 64 == 0
    assert sum(q.owned(r)[1] - q.owned(r)[0] for r in range(8)) == 480000
    # more ranks than 64-row blocks: trailing ranks own only padding
    t = RowPartition(100, 4)
    assert t.n_loc == 64 and t.owned(2) == (128, 128) and t.owned(3) == (192, 192)
    with pytest.raises(ValueError):
        RowPartition(0, 2)


def test_pad_and_slice_graph_on_cpu_tensors(golden):
    """pad_graph / local_csr are pure index bookkeeping: run them on CPU tensors and check that the
    row slices tile the padded CSR exactly."""
    from manifold_gp_amd.graph import KnnGraph
    from manifold_gp_amd.parallel import RowPartition, local_csr, pad_graph
    from oracle import knn as oknn
    g = golden("dumbbell_k10_loop")
    n = g["train_x"].shape[0]
    idx, val = g["edge_index"].astype(np.int64), g["edge_value"]
    # padded symmetric CSR built on the host (same layout as mgp_graph_from_coo)
    rows = np.r_[idx[0], idx[1]]
    cols = np.r_[idx[1], idx[0]]
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    cnt = np.bincount(rows, minlength=n)
    pc = (cnt + 3) // 4 * 4
    rowptr = np.r_[0, np.cumsum(pc)].astype(np.int32)
    col = np.repeat(np.arange(n), pc).astype(np.int32)
    start = np.r_[0, np.cumsum(cnt)]
    for r in range(n):
        col[rowptr[r]:rowptr[r] + cnt[r]] = cols[start[r]:start[r + 1]]
    graph = KnnGraph(n, torch.from_numpy(idx[0].astype(np.int32)), torch.from_numpy(idx[1].astype(np.int32)),
                     torch.from_numpy(val), torch.from_numpy(rowptr), torch.from_numpy(col),
                     torch.zeros(len(col)), torch.zeros(len(col), dtype=torch.int32))
    part = RowPartition(n, 3)
    gp = pad_graph(graph, part.n_pad)
    assert gp.n == part.n_pad and gp.rowptr.shape[0] == part.n_pad + 1
    assert int(gp.rowptr[-1]) == int(graph.rowptr[-1]) and torch.equal(gp.rowptr[: n + 1], graph.rowptr)

    class FakeData:
        pass
    d = FakeData()
    d.graph = gp
    d.vals = torch.arange(len(col), dtype=torch.float32)
    d.diag = torch.arange(part.n_pad, dtype=torch.float32)
    total = 0
    for r in range(3):
        loc = local_csr(d, part, r)
        r0, r1 = part.range(r)
        assert loc["rowptr"][0] == 0 and loc["rowptr"].shape[0] == part.n_loc + 1
        nn = int(loc["rowptr"][-1])
        total += nn
        if nn:
            assert torch.equal(loc["vals"][:nn], d.vals[loc["e0"]:loc["e1"]])
            assert loc["e0"] % 4 == 0                     # 16-byte aligned slice start
        assert torch.equal(loc["diag"], d.diag[r0:r1])
    assert total == int(graph.rowptr[-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from manifold_gp_amd.parallel import RowPartition, distributed_cg_reference
        from oracle.laplacian import LaplacianOracle
        from oracle.precision import PrecisionMaternOracle
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
        n = g["train_x"].shape[0]
        lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
        Q = PrecisionMaternOracle(lap, 2, float(g["kappa"]))
        s_out, noise = 0.7, 1e-2
        part = RowPartition(n, world)
        r0, r1 = part.range(rank)

        def full_apply(u):           # A = I + noise * s * Q on the padded space (padding rows: identity)
            un = u[:n].numpy()
            out = u.clone()
            out[:n] += torch.from_numpy(noise * s_out * Q.matmul(un))
            return out

        def local_matvec(u):         # this rank's rows only -- what the HIP SpMM slice computes
            return full_apply(u)[r0:r1]

        B = torch.zeros(part.n_pad, 2, dtype=torch.float64)
        B[:n, 0] = torch.from_numpy(g["train_y"].astype(np.float64))
        B[:n, 1] = torch.from_numpy(g["probes"][:, 0].astype(np.float64))
        x, its = distributed_cg_reference(local_matvec, B, part, rank, tol=1e-11, max_iter=2000)
        # every rank holds the same replicated solution
        xs = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(xs, x)
        same = all(torch.equal(xs[0], t) for t in xs)
        res = float((full_apply(x) - B).norm() / B.norm())
        if rank == 0:
            q.put(dict(ok=True, same=same, res=res, its=its, x=x[:n].numpy(), pad=float(x[n:].abs().max()) if part.n_pad > n else 0.0))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(dict(ok=False, err=repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_distributed_cg_world2_gloo_matches_single_process(golden):
    from oracle.laplacian import LaplacianOracle
    from oracle.precision import PrecisionMaternOracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert out["ok"], out.get("err")
    assert out["same"] and out["res"] < 1e-10 and out["pad"] == 0.0
    g = golden("dumbbell_k10_loop")
    n = g["train_x"].shape[0]
    lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
    A = np.eye(n) + 1e-2 * 0.7 * PrecisionMaternOracle(lap, 2, float(g["kappa"])).dense()
    ref = np.linalg.solve(A, np.stack([g["train_y"], g["probes"][:, 0]], 1).astype(np.float64))
    assert np.abs(out["x"] - ref).max() < 1e-9 * np.abs(ref).max()


def _worker_pcg(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from manifold_gp_amd.parallel import RowPartition, distributed_pcg_reference
        from oracle.laplacian import LaplacianOracle
        from oracle.sparse import SparsePrecision
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
        n = g["train_x"].shape[0]
        lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
        sq = SparsePrecision(lap, 2, float(g["kappa"]), 0.7)
        noise = 1e-2
        part = RowPartition(n, world)
        r0, r1 = part.range(rank)

        def local_apply(v_full):         # this rank's rows of A v on the padded space (padding rows: identity)
            out = v_full.clone()
            out[:n] = torch.from_numpy(sq.posterior_system(v_full[:n].numpy(), noise))
            return out[r0:r1]

        b = torch.zeros(part.n_pad, dtype=torch.float64)
        b[:n] = torch.from_numpy(g["train_y"].astype(np.float64))
        x_loc, its = distributed_pcg_reference(local_apply, b[r0:r1].clone(), part, rank, tol=1e-11, max_iter=3000)
        xs = [torch.empty_like(x_loc) for _ in range(world)]
        dist.all_gather(xs, x_loc)
        if rank == 0:
            q.put(dict(ok=True, its=its, x=torch.cat(xs)[:n].numpy(), pad=float(torch.cat(xs)[n:].abs().max())))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(dict(ok=False, err=repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_partitioned_pipelined_cg_world2_gloo(golden):
    """The algorithm of csrc/pcg.hip (partitioned vectors, pipelined recurrence, ONE all-gather per iteration) under
    gloo with two processes, the float64 oracle as local operator, against the dense solve."""
    from oracle.laplacian import LaplacianOracle
    from oracle.precision import PrecisionMaternOracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pcg, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert out["ok"], out.get("err")
    g = golden("dumbbell_k10_loop")
    n = g["train_x"].shape[0]
    lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
    A = np.eye(n) + 1e-2 * 0.7 * PrecisionMaternOracle(lap, 2, float(g["kappa"])).dense()
    ref = np.linalg.solve(A, g["train_y"].astype(np.float64))
    assert out["pad"] == 0.0
    assert np.abs(out["x"] - ref).max() < 1e-8 * np.abs(ref).max(), np.abs(out["x"] - ref).max() / np.abs(ref).max()


def _worker_sharded(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from manifold_gp_amd.parallel import solve_columns_sharded
        from oracle.laplacian import LaplacianOracle
        from oracle.sparse import SparsePrecision
        g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
        n = g["train_x"].shape[0]
        lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
        sq = SparsePrecision(lap, 2, float(g["kappa"]), 0.7)
        calls = []

        def oracle_solver(desc, Bc, **kw):             # stands for solvers.cg_solve: the float64 oracle CG, column by column
            calls.append(Bc.shape[1])
            X = sq.solve(Bc.numpy(), matvec=lambda z: sq.posterior_system(z, 1e-2), tol=1e-12)
            return torch.from_numpy(X), 7 + rank, [0.0] * Bc.shape[1]

        rng = np.random.default_rng(3)
        B = torch.from_numpy(rng.normal(size=(n, 5)))                  # 5 columns over 2 ranks: 3 + 2
        X, its = solve_columns_sharded(None, B, rank, world, solver=oracle_solver)
        xs = [torch.empty_like(X) for _ in range(world)]
        dist.all_gather(xs, X)
        if rank == 0:
            q.put(dict(ok=True, same=all(torch.equal(xs[0], t) for t in xs), X=X.numpy(), B=B.numpy(), calls=calls, its=its))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(dict(ok=False, err=repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_sharded_columns_world2_gloo_with_a_column_count_not_divisible_by_the_world(golden):
    """parallel.solve_columns_sharded (the multi-right-hand-side workloads' way of using several GPUs: columns dealt
    round-robin, one all-gather of the solutions) under gloo with two processes and FIVE columns: rank 0 solves columns
    0, 2, 4, rank 1 columns 1, 3 and sends a zero pad column; every rank ends with all five solutions in place."""
    from oracle.laplacian import LaplacianOracle
    from oracle.sparse import SparsePrecision
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert out["ok"], out.get("err")
    assert out["same"] and out["calls"] == [3] and out["its"] == 7
    g = golden("dumbbell_k10_loop")
    n = g["train_x"].shape[0]
    lap = LaplacianOracle(g["edge_value"], g["edge_index"], n, float(g["eps"]), "randomwalk", True, dtype=np.float64)
    sq = SparsePrecision(lap, 2, float(g["kappa"]), 0.7)
    R = sq.posterior_system(out["X"], 1e-2) - out["B"]
    assert np.abs(R).max() < 1e-9 * np.abs(out["B"]).max()


def test_ghost_layers_and_rank_order_on_cpu_tensors():
    """parallel.ghost_layers is index bookkeeping: on a path graph with two extra chords the layers of a row block
    are its successive neighbour shells."""
    from manifold_gp_amd.parallel import ghost_layers

    class G:
        pass
    n = 12
    nbrs = {i: {(i - 1) % n, (i + 1) % n} for i in range(n)}
    nbrs[2].add(9), nbrs[9].add(2)
    rowptr, col = [0], []
    for i in range(n):
        col += sorted(nbrs[i])
        rowptr.append(len(col))
    g = G()
    g.n, g.rowptr, g.col = n, torch.tensor(rowptr, dtype=torch.int32), torch.tensor(col, dtype=torch.int32)
    l1, l2 = ghost_layers(g, 0, 4, 2)
    assert l1.tolist() == [4, 9, 11]                       # neighbours of {0,1,2,3}: 11, 4 and the chord 2-9
    assert l2.tolist() == [5, 8, 10]


def _small_problem():
    """100 nodes on a ring with chords (float64 oracle operator A = I + noise s Q): with 4 ranks and 64-row blocks the
    partition is [0, 64), [64, 100) + padding, and TWO ranks that own nothing but padding rows."""
    from oracle.laplacian import LaplacianOracle
    from oracle.sparse import SparsePrecision
    n = 100
    rng = np.random.default_rng(2)
    pairs = {(i, (i + 1) % n) for i in range(n)} | {(int(a), int(b)) for a, b in rng.integers(0, n, (60, 2)) if a != b}
    pairs = np.array(sorted({(min(a, b), max(a, b)) for a, b in pairs}), dtype=np.int64).T
    val = (rng.random(pairs.shape[1]) * 0.02).astype(np.float64)
    lap = LaplacianOracle(val, pairs, n, 0.1, "randomwalk", True, dtype=np.float64)
    return n, SparsePrecision(lap, 2, 0.8, 0.7), 1e-2, rng.normal(size=n)


def _worker_small(rank, world, port, q, algo):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from manifold_gp_amd.parallel import RowPartition, distributed_cg_reference, distributed_pcg_reference
        n, sq, noise, b_np = _small_problem()
        part = RowPartition(n, world)
        r0, r1 = part.range(rank)

        def rows_of_A(v_full):             # padding rows: identity
            out = v_full.clone()
            out[:n] = torch.from_numpy(sq.posterior_system(v_full[:n].numpy(), noise)).to(out.dtype).reshape(out[:n].shape)
            return out[r0:r1]

        b = torch.zeros(part.n_pad, dtype=torch.float64)
        b[:n] = torch.from_numpy(b_np)
        if algo == "pcg":
            x_loc, its = distributed_pcg_reference(rows_of_A, b[r0:r1].clone(), part, rank, tol=1e-12, max_iter=500)
            xs = [torch.empty_like(x_loc) for _ in range(world)]
            dist.all_gather(xs, x_loc)
            x = torch.cat(xs)
        else:
            def local_matvec(u):           # [n_pad, C] replicated vectors, this rank's rows
                out = u.clone()
                out[:n] = torch.from_numpy(sq.posterior_system(u[:n].numpy(), noise))
                return out[r0:r1]
            xr, its = distributed_cg_reference(local_matvec, b.view(-1, 1), part, rank, tol=1e-12, max_iter=500)
            x = xr[:, 0]
        if rank == 0:
            q.put(dict(ok=True, its=its, x=x[:n].numpy(), pad=float(x[n:].abs().max()), owned=[part.owned(r) for r in range(world)]))
    except Exception as e:  # pragma: no cover
        if rank == 0:
            q.put(dict(ok=False, err=repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["pcg", "cg"])
def test_world4_gloo_with_ranks_that_own_only_padding(algo):
    """Four processes, 100 nodes: ranks 2 and 3 own no real row (RowPartition pads to whole 64-row blocks).  Both
    distributed recurrences (the partitioned pipelined CG of csrc/pcg.hip and round 1's replicated-vector CG) must still
    take identical decisions on every rank -- the padding ranks contribute zeros to every gathered dot product and must
    issue the same collectives -- and reproduce the dense float64 solve."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_small, args=(r, 4, port, q, algo)) for r in range(4)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["ok"], out.get("err")
    assert out["owned"][2] == (128, 128) and out["owned"][3] == (192, 192)
    n, sq, noise, b = _small_problem()
    A = np.stack([sq.posterior_system(e, noise) for e in np.eye(n)], 1)
    ref = np.linalg.solve(A, b)
    assert out["pad"] == 0.0 and out["its"] < 200
    assert np.abs(out["x"] - ref).max() < 1e-9 * np.abs(ref).max(), np.abs(out["x"] - ref).max() / np.abs(ref).max()
