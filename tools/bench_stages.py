"""Stage-by-stage figures of the hot path for bench.py's N = 1 line (`stages`, `reference_bench`, `manifold_784`).

The headline of bench.py is the C = 1 SpMV inside the CG solve.  Every other stage of the path has its own roofline in
SURVEY.md section 8(d) -- k-NN (fp32 FMA / bf16 MFMA), graph symmetrise + Laplacian build (HBM), eigensolve (wide SpMM), spectral
features, the kernel block (fp32 MFMA), the training epochs of manifold_gp/utils/train_model.py:63-90 -- and the reference's own
micro-benchmark times three legs (benchmark/bench_sparse_laplacian.py:15-34: mv, grad, eigen).  This module times each of them
on the bench's workload in the bench's process so that the driver's line carries them; wall-clock per call (perf_counter around
a stream synchronise: what a caller of the Python surface sees, launch overhead included) unless a kernel-only figure is named.

Bench infrastructure: nothing here is on the product path."""
import ctypes
import json
import os
import statistics
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA dense peak
BF16_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def timed(fn, reps=3, warm=1):
    """(last result, [ms per call]) with a device synchronise on both sides of every call."""
    out = None
    for _ in range(warm):
        out = fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return out, ts


def _ms(ts):
    return dict(ms=round(min(ts), 4), ms_median=round(statistics.median(ts), 4), reps=len(ts))


def spmm_repeat_us(csr, n, C, reps=50):
    """Back-to-back launches of Y = L X (mgp_spmm_repeat: one hipGraph of `reps` launches, HIP events on the launch stream)."""
    from manifold_gp_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    X = torch.rand(n, C, device=dev)
    Y = torch.empty_like(X)
    ms = ctypes.c_float(0.0)
    st = _lib.stream()
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 5, None, st), "mgp_spmm_repeat")
    best = 1e30
    for _ in range(3):
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), reps, ctypes.byref(ms), st), "mgp_spmm_repeat")
        best = min(best, ms.value)
    return best / reps * 1e3


def knn_stage(x, k, knn=None):
    """NearestNeighbors.search of the points against themselves (nearest_neighbors.py:35-37 as the graph build calls it)."""
    import manifold_gp_amd as mgp
    N, d = x.shape
    knn = knn or mgp.utils.NearestNeighbors(x)
    (_, _), ts = timed(lambda: knn.search(x, k), reps=3)
    t = min(ts) * 1e-3
    out = dict(shape="%d x %d, k = %d (self-search of the graph build)" % (N, d, k), **_ms(ts),
               pair_distances_per_s=round(N * N / t, 1), stats=knn.last_stats)
    if d >= 32:
        dpad = -(-d // 32) * 32
        tiles = -(-N // 128)
        pairs = tiles * (tiles + 1) // 2 * 128 * 128              # tile pairs on and above the diagonal (mgp_knn_set_symmetric)
        filtered = knn.last_stats.get("filter_failover_rows", -1) >= 0
        if filtered:
            # + the keys of every row to the sampled points (every 16th point up to k = 64: mgp_knn_set_filter)
            want = k + max(k // 2, 24)
            stride = 16
            while stride > 4 and stride * want > 1536:
                stride //= 2
            pairs += tiles * -(-(-(-N // stride)) // 128) * 128 * 128
        from manifold_gp_amd import _lib as _l
        out["pipeline"] = ("candidate filter: sampled bounds -> filtered key pass (log) -> regroup -> select from lists; no key slab"
                           if filtered else "key slab: keys -> N x n slab -> select")
        out["workspace_GB"] = round(_l.lib().mgp_knn_workspace_bytes(N, N, d, k) / 1e9, 2)
        out.update(effective_fp32_tflops=round(2.0 * N * N * d / t / 1e12, 1), fp32_peak_tflops=FP32_PEAK_TFLOPS,
                   frac_of_fp32_peak=round(2.0 * N * N * d / t / 1e12 / FP32_PEAK_TFLOPS, 3),
                   bf16_mfma_tflops_whole_search=round(3 * 2.0 * pairs * dpad / t / 1e12, 1), bf16_peak_tflops=BF16_PEAK_TFLOPS,
                   frac_of_bf16_peak_whole_search=round(3 * 2.0 * pairs * dpad / t / 1e12 / BF16_PEAK_TFLOPS, 3),
                   note="effective = 2 N^2 d flops of the all-pairs GEMM form over the WHOLE search (keys + select + fp64 "
                        "re-rank); the key kernel executes 3 bf16 MFMA products per pair on the upper-triangle tile pairs only: "
                        "bf16_mfma_tflops_whole_search divides those flops by the whole search time (a lower bound of the key "
                        "kernel's own rate, which profiles/*_knn_kernel_stats.csv gives)")
    else:
        out["note"] = "low-dimensional path (Morton order + chunk-box pruning): brute-force-equivalent pair distances per second"
    return out


def graph_laplacian_stage(knn, x, k, eps):
    """graph symmetrise (nearest_neighbors.py:39-55) + Laplacian build (graph_laplacian_operator.py:52-106)."""
    from manifold_gp_amd.graph import KnnGraph, LaplacianData
    D, I = knn.search(x, k)
    I32 = I.to(torch.int32)
    g, ts_g = timed(lambda: KnnGraph.from_knn(D, I32, points=x if x.shape[1] <= 3 else None), reps=3)
    _, ts_l = timed(lambda: LaplacianData(g, eps, True), reps=5)
    lap_bytes = 24 * 2 * g.M + 16 * g.n
    t = min(ts_l) * 1e-3
    return dict(graph_symmetrise_csr_tiles=dict(**_ms(ts_g), edges=g.M, nnz_padded=g.nnz,
                                                note="sort + merge of N (k - 1) directed pairs, padded CSR, 64-row tile dictionaries; "
                                                     "synchronises (edge count to the host)"),
                laplacian_build=dict(**_ms(ts_l), bytes=lap_bytes, gbs=round(lap_bytes / t / 1e9, 1), peak=HBM_PEAK_GBS,
                                     frac=round(lap_bytes / t / 1e9 / HBM_PEAK_GBS, 4),
                                     note="three fused row passes + six output allocations, wall clock of the Python call "
                                          "(launch-bound at 60k: the kernels are < 0.1 ms); bytes = 24 * 2M + 16 N (SURVEY 8d)"))


def spectral_stage(kern, x, eps, kappa, spmm_bytes, oos_points=600):
    """eval() (riemann_kernel.py:117-130), features in / out of sample (:132-149), kernel block (:92-100)."""
    from manifold_gp_amd.solvers import kernel_block, lanczos_smallest
    kern.initialize(graphbandwidth=eps, lengthscale=kappa)
    import warnings
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        _, ts = timed(lambda: kern.eval(), reps=3)
    info = list(getattr(lanczos_smallest, "last_info", [0, 0, 0, 0]))
    out["eigensolve"] = dict(**_ms(ts), modes=int(kern.num_modes), tol=kern.eigen_tol, rounds=info[0], spmm_applies=info[1],
                             pairs_under_tol=info[2], max_residual=float(max(kern.eigen_residuals)),
                             note="RiemannKernel.eval(): Laplacian build + block eigensolver + post-processing; cold every time on this "
                                  "graph (the solve ends at the fp32 residual floor: no block is kept for a warm start; manifold_784 "
                                  "carries the warm figure)")
    data = kern.laplacian_operator.data
    g = data.graph
    # the matrix the eigensolver / the wide solves multiply with: chain-relabelled where the graph has such an order (round 5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rel = data.wide_relabelled()
    torch.cuda.synchronize()
    out["wide_relabelling"] = dict(first_call_ms=round((time.perf_counter() - t0) * 1e3, 2),
                                   order=("nearest-neighbour chain (host walk) + relabelled CSR + tile image" if rel is not None and
                                          getattr(rel.graph, "emap", None) is not None else
                                          ("graph's own locality order" if rel is not None else "caller's order")),
                                   note="cached per graph / bandwidth: 0 when eval() above has built it")
    csr = (rel or data).csr(wide=True)
    mt = (rel or data).mt_plan(False)
    if mt is not None:
        out["wide_relabelling"].update(tile_image_MB=round(mt.img.numel() * 4 / 1e6, 1), steps=mt.steps, fill=round(mt.fill, 3))
    from manifold_gp_amd import _lib
    names = {0: "spmm_kernel (per-entry X-row gather)", 3: "spmm_mt_kernel (fp32 matrix-core 16-row tiles)", 5: "spmm_dict_kernel",
             6: "chunked dictionary"}
    for C in (128, 100):
        us = spmm_repeat_us(csr, g.n, C)
        choice = int(_lib.lib().mgp_spmm_kernel_choice(ctypes.byref(csr), C, 0, 0))
        B = spmm_bytes(g.n, g.M, C)
        out["spmm_wide_C%d" % C] = dict(us=round(us, 2), bytes=B, gbs=round(B / us / 1e3, 1), peak=HBM_PEAK_GBS,
                                        frac=round(B / us / 1e3 / HBM_PEAK_GBS, 4),
                                        kernel=names.get(choice, "choice %d" % choice))
    Z, ts = timed(lambda: kern.features(x), reps=5)
    out["features_insample"] = dict(**_ms(ts), shape=list(Z.shape))
    torch.manual_seed(7)
    xt = (x[:oos_points] + 1e-3 * torch.randn_like(x[:oos_points])).contiguous()
    Zt, ts = timed(lambda: kern.features(xt), reps=5)
    out["features_oos"] = dict(**_ms(ts), shape=list(Zt.shape), note="k-NN search of %d jittered training points + fused Nystrom / "
                                                                      "bump kernel" % oos_points)
    K, ts = timed(lambda: kernel_block(Zt, Z, 1.0), reps=10, warm=2)
    fl = 2.0 * Zt.shape[0] * Z.shape[0] * Z.shape[1]
    t = min(ts) * 1e-3
    out["kernel_block"] = dict(**_ms(ts), shape="%d x %d x %d" % (Zt.shape[0], Z.shape[0], Z.shape[1]),
                               tflops=round(fl / t / 1e12, 1), peak=FP32_PEAK_TFLOPS, frac=round(fl / t / 1e12 / FP32_PEAK_TFLOPS, 3),
                               note="wall clock of solvers.kernel_block (output allocation + launch included)")
    return out


def training_stage(x, y, hp, dev, semisup=False, epochs=4):
    """One epoch of manifold_informed_train (train_model.py:63-90: loss + gradients wrt the 4 hyper-parameters + Adam step),
    supervised or with 10 % labelled points (Schur complement inside the loss).  Epoch times from a scheduler hook."""
    import warnings
    import manifold_gp_amd as mgp
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from manifold_gp_amd.utils import manifold_informed_train
    from tools import synth
    kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
    D1, _ = kern.knn.search(x[: min(20000, x.shape[0])], 2)
    eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
    kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
    if semisup:
        torch.manual_seed(1337)
        labeled = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
        labeled[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
        model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev),
                          labeled=labeled).to(dev)
    else:
        model = RiemannGP(x, y, GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev)).to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    times, losses = [], []

    class Rec:
        def step(self, loss):
            torch.cuda.synchronize()
            times.append(time.perf_counter())
            losses.append(float(loss.detach()))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        manifold_informed_train(model, opt, max_iter=epochs - 1, tolerance=0.0, num_rand_vec=32 if semisup else 100, max_cholesky=800,
                                cg_tolerance=1e-2, cg_max_iter=1000, scheduler=Rec())
        torch.cuda.synchronize()
    ep = [(b - a) * 1e3 for a, b in zip([t0] + times[:-1], times)]
    steady = ep[1:] if len(ep) > 1 else ep
    return dict(epoch_ms=round(min(steady), 2), epoch_ms_all=[round(e, 2) for e in ep], epochs=len(ep), first_loss=losses[0],
                last_loss=losses[-1], labelled=int(model.train_targets.shape[0]), nodes=int(x.shape[0]),
                note="epoch 1 includes the untimed-elsewhere `_average_variance` normalisation and first-call plan captures; "
                     "epoch_ms = fastest later epoch")


def manifold784_block(dev, spmm_bytes, n_all=60600):
    """Secondary 60k workload with a conditioned spectrum (tools/synth.py::manifold_784; the workload of
    test_c3_manifold784_60k_posterior_vs_independent_float64): the whole pipeline + posterior at 600 held-out points, timed."""
    import warnings
    import manifold_gp_amd as mgp
    from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
    from manifold_gp_amd.solvers import CgPlan
    from tools import synth
    k, m, nu = 50, 100, 2
    eps, kappa, s, noise = 0.3, 3.0, 1.0, 1e-2
    if n_all != 60600:              # reduced sizes (contract tests): the k = 50 neighbourhood radius grows as 1 / sqrt(density)
        eps = eps * (60600.0 / n_all) ** 0.5
    x_np, y_np, _ = synth.manifold_784(n_all)
    rng = np.random.default_rng(11)
    perm = rng.permutation(n_all)
    te, tr = np.sort(perm[:600]), np.sort(perm[600:])
    x, y = torch.from_numpy(x_np[tr]).to(dev), torch.from_numpy(y_np[tr]).to(dev)
    xt, yt = torch.from_numpy(x_np[te]).to(dev), torch.from_numpy(y_np[te]).to(dev)
    out = dict(workload="swiss roll in R^784, N = %d + 600 held out, k = 50, randomwalk, nu = 2, 100 modes, eps %.3g, kappa 3, noise 1e-2"
                        % (n_all - 600, eps))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = mgp.kernels.RiemannMaternKernel(nu=nu, x=x, nearest_neighbors=k, laplacian_normalization="randomwalk", num_modes=m,
                                           bump_scale=3.0, bump_decay=0.01).to(dev)
    torch.cuda.synchronize()
    out["kernel_ctor_knn_graph_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    kern.initialize(graphbandwidth=eps, lengthscale=kappa)
    model = RiemannGP(x, y, GaussianLikelihood(noise).to(dev), ScaleKernel(kern, s).to(dev)).to(dev)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        kern.warm_start = False
        _, ts = timed(lambda: model.eval(), reps=2)
        out["eval_eigensolve_ms"] = round(min(ts), 2)
        out["eval_cold"] = dict(ms=round(min(ts), 2), rounds=kern.eigen_info[0], block_products=kern.eigen_info[1])
        # eval() again after a 1 % bandwidth change, started from the previous block (mgp_lanczos_smallest_warm), then back
        kern.warm_start = True
        model.eval()
        kern.initialize(graphbandwidth=1.01 * eps)
        _, tw = timed(lambda: model.eval(), reps=1, warm=0)
        out["eval_warm_after_1pct_bandwidth_change"] = dict(ms=round(tw[0], 2), rounds=kern.eigen_info[0], block_products=kern.eigen_info[1],
                                                            pairs_under_tol=kern.eigen_info[2], max_residual=float(max(kern.eigen_residuals)))
        kern.initialize(graphbandwidth=eps)
        model.eval()
    out["eigen_max_residual"] = float(max(kern.eigen_residuals))

    def post():
        model._cache = None
        model.posterior(xt)
        return model.posterior_mean
    mean, ts = timed(post, reps=3)
    out["posterior_600_points_ms"] = round(min(ts), 3)
    out["posterior_test_rmse"] = float((mean - yt).square().mean().sqrt())
    # the same posterior mean at the nodes in precision form: CG on (I + noise s Q) x = y
    desc = kern.precision()._descriptor().with_(scale=s, form=2, noise=noise)
    plan = CgPlan(desc, 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8)
    yy = y.view(-1, 1).contiguous()
    for _ in range(3):
        plan.solve(yy, copy=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        plan.solve(yy, copy=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    g = kern.knn.knn_graph
    spmvs = plan.applies * nu
    out["precision_cg"] = dict(ms=round(dt * 1e3, 4), iterations=plan.iters, spmv_per_solve=spmvs, rel_residual=float(max(plan.resid)),
                               gbs=round(spmm_bytes(g.n, g.M) * spmvs / dt / 1e9, 1))
    plan.close()
    return out


def reference_bench(dev):
    """tools/bench_reference_shape.py: benchmark/bench_sparse_laplacian.py's mv / grad / eigen at its 5 000-point shape, GPU
    path and reference-style torch on the host cores."""
    from tools import bench_reference_shape
    return bench_reference_shape.run(dev)


def profile_share(name_regex):
    """kernel-time share of a training epoch from the newest committed profile summary (profiles/rNN_training.json, written
    by tools/summarize_training_profile.py from a rocprofv3 --kernel-trace of tools/run_training*.py); None when absent."""
    import re
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for f in sorted((f for f in os.listdir(pdir) if re.fullmatch(name_regex, f)), reverse=True):
        try:
            return dict(json.load(open(os.path.join(pdir, f))), source="profiles/" + f)
        except Exception:
            continue
    return None


def _pcg_eager(desc, b, tol, max_iter, precond=None):
    """Textbook preconditioned CG with the operator applies on the HIP path and the vector algebra in torch ops (bench
    infrastructure: the like-for-like loop the three preconditioners are compared in).  Stops on ||r|| <= tol ||b||.
    Returns (x, iterations, operator applies)."""
    x = torch.zeros_like(b)
    r = b.clone()
    applies = [0]

    def A(v):
        applies[0] += 1
        return desc.apply(v)
    z = precond(r, A) if precond is not None else r
    p = z.clone()
    rz = torch.dot(r.view(-1), z.view(-1))
    bn = float(b.norm())
    it = 0
    for it in range(1, max_iter + 1):
        q = A(p)
        alpha = rz / torch.dot(p.view(-1), q.view(-1))
        x += alpha * p
        r -= alpha * q
        if float(r.norm()) <= tol * bn:
            break
        z = precond(r, A) if precond is not None else r
        rz_new = torch.dot(r.view(-1), z.view(-1))
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it, applies[0]


def chebyshev_preconditioner(lmin, lmax, degree):
    """z = p(A) r, p the degree-`degree` Chebyshev approximation of 1 / t on [lmin, lmax]: `degree` steps of the Chebyshev
    iteration for A z = r from z = 0 (a fixed polynomial in A: symmetric positive definite, valid inside CG).  Costs `degree`
    operator applies per call."""
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta

    def apply(r, A):
        rho = 1.0 / sigma
        d = r / theta
        z = d.clone()
        for _ in range(degree):
            rho_new = 1.0 / (2.0 * sigma - rho)
            res = r - A(z)
            d = (rho_new * rho) * d + (2.0 * rho_new / delta) * res
            z = z + d
            rho = rho_new
        return z
    return apply


def preconditioner_block(desc, y, tol, refine, max_iter=5000, degrees=(2, 4)):
    """North star: "the preconditioned-CG solve of (K + sigma^2 I) x = y"; the reference's linear_cg call is unpreconditioned
    (precision_matern_operator.py:53).  The same system solved (i) by the plan without a preconditioner (the default), (ii) by
    the plan with Jacobi (mgp_operator_jacobi), (iii) with a Chebyshev polynomial in A as preconditioner -- in an eager
    loop, next to the same loop without one, so that (iii) is compared like for like: iterations, operator applies
    (nu SpMVs each) and wall time."""
    from manifold_gp_amd import _lib
    from manifold_gp_amd.solvers import CgPlan
    yy = y.view(-1, 1).contiguous()
    out = {}
    lib = _lib.lib()
    for name, jac, cx in (("none", False, 0), ("jacobi", True, 0), ("complex_shift", False, 1)):
        prev = lib.mgp_cg_set_complex_shift(cx)
        try:
            plan = CgPlan(desc, 1, tol=tol, max_iter=max_iter, stop_mode=1, check_every=8, refine=refine, jacobi=jac)
        finally:
            lib.mgp_cg_set_complex_shift(prev)
        if cx and not plan.complex_shift:      # the system does not factorise (random-walk pre / post vectors, nu != 2, form 0)
            plan.close()
            out[name] = dict(applicable=False, why="A = I + c B^2 with B = tau I + L_sym only (form 2, nu = 2, symmetric normalisation)")
            continue
        plan.solve(yy, copy=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = plan.solve(yy, copy=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = desc.apply(x) - yy
        out[name] = dict(iterations=plan.iters, ms=round(dt * 1e3, 3), status=plan.status, resid_reported=float(max(plan.resid)),
                         true_rel_residual_fp32_apply=float(r.norm() / yy.norm()),
                         solver=("mgp_cg_plan, COCG on I + i sigma B: one 4-column product with B per iteration" if cx else
                                 "mgp_cg_plan, CG on A (hipGraph chunks): nu SpMVs per iteration"))
        plan.close()
    # spectrum bounds of A = (form 2) I + s c (tau + L)^nu or (form 0) c (tau + L)^nu, L_sym in [0, 2 max diag] (Gershgorin)
    d = desc.data
    tau = 2.0 * desc.nu / desc.kappa ** 2
    lam_hi = 2.0 * float(d.diag.max())
    pre_lo = float(d.dsqrt.min()) ** 2 if desc.pre is not None else 1.0
    pre_hi = float(d.dsqrt.max()) ** 2 if desc.pre is not None else 1.0
    q_lo, q_hi = desc.scale * pre_lo * tau ** desc.nu, desc.scale * pre_hi * (tau + lam_hi) ** desc.nu
    lmin, lmax = (1.0 + desc.noise * q_lo, 1.0 + desc.noise * q_hi) if desc.form == 2 else (q_lo, q_hi)
    out["spectrum_bounds_used"] = [lmin, lmax]
    tol_e = max(tol, 1e-5)      # the eager fp32 recurrences are compared at a tolerance all of them reach without refinement
    for name, pc in [("eager_loop_none", None)] + [("chebyshev_%d" % k, chebyshev_preconditioner(lmin, lmax, k)) for k in degrees]:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x, its, applies = _pcg_eager(desc, yy, tol_e, max_iter, pc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r = desc.apply(x) - yy
        out[name] = dict(iterations=its, operator_applies=applies, ms=round(dt * 1e3, 3), tol=tol_e,
                         true_rel_residual_fp32_apply=float(r.norm() / yy.norm()),
                         solver="eager torch loop over mgp_operator_apply (no graph): compare with eager_loop_none")
    base = out["eager_loop_none"]
    best = min((out[k] for k in out if k.startswith("chebyshev_")), key=lambda e: e["operator_applies"])
    cs = out.get("complex_shift", {})
    out["verdict"] = ("chebyshev polynomial preconditioning trades iterations (dots / updates, and collectives at N > 1) for operator "
                      "applies: best degree needs %d applies against %d without (CG is optimal over the same Krylov space, so the "
                      "apply count cannot drop) -- not taken; Jacobi: see its row (it bites only where diag(A) varies: the random-walk "
                      "form D^1/2 (.) D^1/2, which the factorised solves already divide out); " % (best["operator_applies"], base["operator_applies"])
                      + ("the complex-shift factorisation is the default for this system: %d iterations of one product against %d of %d"
                         % (cs["iterations"], out["none"]["iterations"], int(desc.nu)) if cs.get("iterations") else
                         "the complex-shift factorisation does not apply to this system"))
    return out
