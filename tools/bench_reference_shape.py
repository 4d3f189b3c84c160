#!/usr/bin/env python
"""The reference's own micro-benchmark (benchmark/bench_sparse_laplacian.py: 5000 RMNIST points, k = 50,
symmetric Laplacian; matvec, gradient wrt the bandwidth, eigendecomposition) on the GPU path and, beside
it, the reference-style torch path on this box's host cores (oracle/ref_torch.py).  Prints one JSON
object.  GPU box only; the CPU leg is bench / test infrastructure, never the product path."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp  # noqa: E402
from oracle.ref_torch import TorchCooLaplacian, torch_laplacian_from_edges  # noqa: E402
from tools import synth  # noqa: E402


def sync_time(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


def run(dev=None, cpu=True):
    """Returns the dict main() prints.  cpu=False skips the host legs (dense eigh of the 5000 x 5000 matrix: ~1 s)."""
    dev = dev or torch.device("cuda:0")
    threads = min(len(os.sched_getaffinity(0)), 16)
    torch.set_num_threads(threads)
    x_np, _ = synth.rmnist_like(50, 100, seed=1337)
    n = x_np.shape[0]
    x = torch.from_numpy(x_np).to(dev)
    res = dict(shape="5000 x 784 RMNIST-like, k = 50, symmetric (benchmark/bench_sparse_laplacian.py:40-58)",
               cpu_threads=threads)
    knn = mgp.utils.NearestNeighbors(x)
    (idx, val), t = sync_time(lambda: knn.graph(50))
    res["gpu_knn_graph_ms"] = round(t, 2)
    d1, _ = knn.search(x, 2)
    eps, _ = synth.bandwidth_rule(d1[:, 1].cpu().numpy(), 0.05)
    v = torch.rand(n, generator=torch.Generator().manual_seed(1337))
    vd = v.to(dev)

    # ---- matvec (bench_sparse_mv)
    eps_t = torch.tensor([[eps]], device=dev)
    lap = mgp.operators.GraphLaplacianOperator(val, idx, n, eps_t, "symmetric", graph=knn.knn_graph)
    lap.matmul(vd.view(-1, 1))
    _, t = sync_time(lambda: lap.matmul(vd.view(-1, 1)), reps=50)
    res["gpu_matvec_ms"] = round(t, 4)
    eps_c = torch.tensor(float(eps), requires_grad=True)
    with torch.no_grad():
        diag, triu, deg = torch_laplacian_from_edges(val.cpu(), idx.cpu(), n, eps_c)
    ref = TorchCooLaplacian(idx.cpu(), triu, diag, deg)
    t0 = time.perf_counter(); out_c = ref.matmul(v.view(-1, 1)); res["cpu_matvec_first_call_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    t0 = time.perf_counter()
    for _ in range(20):
        ref.matmul(v.view(-1, 1))
    res["cpu_matvec_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
    out_g = lap.matmul(vd.view(-1, 1)).cpu()
    res["matvec_max_rel_diff"] = float((out_g - out_c).abs().max() / out_c.abs().max())

    # ---- gradient wrt the bandwidth (bench_sparse_grad)
    def gpu_grad():
        e = torch.tensor([[eps]], device=dev, requires_grad=True)
        op = mgp.operators.GraphLaplacianOperator(val, idx, n, e, "symmetric", graph=knn.knn_graph)
        loss = op.matmul(vd.view(-1, 1)).sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        return float(e.grad), (time.perf_counter() - t0) * 1e3
    gpu_grad()
    g_gpu, t = gpu_grad()
    res["gpu_grad_backward_ms"] = round(t, 3)

    def cpu_grad():
        e = torch.tensor(float(eps), requires_grad=True)
        dg, tr, de = torch_laplacian_from_edges(val.cpu(), idx.cpu(), n, e)
        loss = TorchCooLaplacian(idx.cpu(), tr, dg, de).matmul(v.view(-1, 1)).sum()
        t0 = time.perf_counter()
        loss.backward()
        return float(e.grad), (time.perf_counter() - t0) * 1e3
    cpu_grad()
    g_cpu, t = cpu_grad()
    res["cpu_grad_backward_ms"] = round(t, 3)
    res["grad_rel_diff"] = abs(g_gpu - g_cpu) / max(abs(g_cpu), 1e-30)

    # ---- eigendecomposition (bench_sparse_eigen: dense symeig of the 5000 x 5000 matrix)
    (ev_d, _), t = sync_time(lambda: lap.diagonalization(method="symeig"))
    res["gpu_dense_symeig_ms"] = round(t, 1)
    (ev_b, _), t = sync_time(lambda: lap.diagonalization(method="lanczos", num_modes=100))
    res["gpu_block_eigensolver_100_modes_ms"] = round(t, 1)
    res["eig_100_max_abs_diff"] = float((ev_b[1:100].cpu() - ev_d[1:100].cpu()).abs().max())
    if cpu:
        dense = torch.zeros(n, n)
        ic = idx.cpu()
        dense[ic[0], ic[1]] = -triu
        dense[ic[1], ic[0]] = -triu
        dense[torch.arange(n), torch.arange(n)] = diag
        t0 = time.perf_counter()
        torch.linalg.eigh(dense)
        res["cpu_dense_symeig_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    return res


def main():
    print(json.dumps(run(), indent=1))


if __name__ == "__main__":
    main()
