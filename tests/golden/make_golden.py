#!/usr/bin/env python
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Imports, by file path and with PYTHONDONTWRITEBYTECODE=1, the pieces of the reference that run
without gpytorch/faiss/torch_sparse:
  * /root/reference/test/_dense_operators.py      (graph_laplacian, matern_* dense twins)
  * /root/reference/manifold_gp/utils/torch_utils.py   (bump_function)
  * /root/reference/manifold_gp/utils/load_dataset.py  (get_data, groundtruth_from_samples)
and evaluates them on the configuration of the reference's live test
(test/test_laplacian.py:18-50: dumbbell, seed-1337 split with 10 held-out points, k=50, nu=1,
20 modes, eps=kappa=0.5, bump 3.0/1.0) plus a k=10 / self-loop variant that matches the kernel
defaults (riemann_kernel.py:30-35,115).

The k-NN graph fed to the reference code comes from oracle/knn.py (faiss is absent: parity
unpinned for the k-NN row); everything downstream of (idx, val) is reference arithmetic.
The eigen/feature/out-of-sample expectations restate riemann_kernel.py:121-136 and
test/_test_functions.py:134-150 with the same torch calls on top of the imported dense
Laplacian (those two files import gpytorch and cannot be imported here).

Only DATA is written (inputs + expected outputs, npz); no reference source text is stored.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True

from oracle import knn as oknn  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def reference_posterior(dense, tu, train_x, train_y, post_x, k, eps, kappa, nu, modes, bump, norm, outputscale, noise):
    """The reference's whole prediction pipeline in float64 on the imported reference functions, dense and without any
    low-rank shortcut: dense Laplacian (test/_dense_operators.py:7-24, self loops as riemann_kernel.py:115 builds it)
    -> torch.linalg.eigh, first `modes` pairs, eigval[0] = 0, D^-1/2 scaling + column normalisation
    (riemann_kernel.py:121-128) -> in-sample features (:134-136) / out-of-sample features with the support mask and
    the bump (:138-147, dense extension matrix of test/_test_functions.py:134-150) -> ExactGP prediction with a
    zero ConstantMean, ScaleKernel(outputscale) and Gaussian noise (riemann_gp.py:16-30,45-46):
        mean = K*t (K + noise I)^-1 y,   cov = K** - K*t (K + noise I)^-1 K*t^T,   K = outputscale Z Z^T.
    Returns numpy float64 (mean [T], cov [T,T], alpha = (K + noise I)^-1 y [n], Z* Z^T of the first 64 nodes)."""
    n = train_x.shape[0]
    D, I = oknn.knn_search(train_x.numpy(), train_x.numpy(), k)
    idx_np, val_np = oknn.knn_graph_from_search(D, I, n)
    Dt, It = oknn.knn_search(train_x.numpy(), post_x.numpy(), k)
    idx, val = torch.from_numpy(idx_np), torch.from_numpy(val_np).double()
    gb = torch.tensor([[eps]], dtype=torch.float64)
    ls = torch.tensor([[kappa]], dtype=torch.float64)
    # eval() always diagonalises the SYMMETRIC matrix (riemann_kernel.py:121-124 assembles diag / -triu, which are
    # the symmetric-normalised pieces for both settings) and always rescales by D^-1/2
    Lsym, _, deg_un, _, deg = dense.graph_laplacian(idx, val, gb, n, normalization="symmetric", self_loops=True)
    ev, U = torch.linalg.eigh(Lsym)
    gap = (ev[modes] - ev[modes - 1]).item()
    ev, U = ev[:modes].clone(), U[:, :modes].clone()
    ev[0] = 0.0
    U = U * deg.pow(-0.5).view(-1, 1)
    U = torch.nn.functional.normalize(U, p=2, dim=0)
    sd = (2 * nu / ls.square() + ev).pow(-nu)                  # riemann_matern_kernel.py:21-22
    sd = sd / sd.sum()
    Z = (sd * n).sqrt() * U                                     # riemann_kernel.py:134-136
    dv, di = torch.from_numpy(Dt).double(), torch.from_numpy(It)
    T = di.shape[0]
    within = dv[:, 0].sqrt() < bump[0] * eps                    # :140
    Zt = torch.zeros(T, modes, dtype=torch.float64)
    if within.sum() != 0:                                       # :142 -- only the rows inside the support are extended
        wv, wi = dv[within], di[within]
        Tw = wi.shape[0]
        rows = torch.arange(Tw).repeat_interleave(wi.shape[1])
        aeu = torch.sparse_coo_tensor(torch.stack([rows, wi.reshape(-1)]), wv.reshape(-1).div(-4 * gb.square()).exp().squeeze(),
                                      (Tw, n)).to_dense()
        deu = aeu.sum(dim=1)
        ae = torch.mm(deu.pow(-1).diag(), torch.mm(aeu, deg_un.pow(-1).diag()))
        de = ae.sum(dim=1)
        if norm == "symmetric":
            ext = torch.mm(de.pow(-0.5).diag(), torch.mm(ae, deg.pow(-0.5).diag()))
        else:
            ext = torch.mm(de.pow(-1.0).diag(), ae)
        sd2 = (2 * nu / ls.square() + ev).pow(-nu) / (1 - ev * gb.square()).square()    # :144
        sd2 = sd2 / sd2.sum() * n
        b = tu.bump_function(wv[:, 0].sqrt(), torch.tensor(bump[0] * eps, dtype=torch.float64), bump[1])
        Zt[within] = sd2.sqrt() * torch.mm(ext, U) * b.unsqueeze(-1)                    # :146-147
    y = train_y.double()
    K = outputscale * (Z @ Z.T) + noise * torch.eye(n, dtype=torch.float64)
    Ks = outputscale * (Zt @ Z.T)
    alpha = torch.linalg.solve(K, y)
    mean = Ks @ alpha
    cov = outputscale * (Zt @ Zt.T) - Ks @ torch.linalg.solve(K, Ks.T)
    return dict(mean=mean.numpy(), cov=cov.numpy(), alpha=alpha.numpy(), cross64=(Zt @ Z[:64].T).numpy(),
                within=within.numpy(), gap=np.float64(gap), evals=ev.numpy(), knn_D=Dt, knn_I=It.astype(np.int32))


def posterior_goldens(dense, tu, train_x, train_y, test_x):
    """tests/golden/dumbbell_posterior.npz: the end-to-end posterior of the reference pipeline (float64) for the two
    dumbbell configurations, both normalisations, nu in {1, 2}; test inputs = the 10 held-out nodes, 40 jittered
    training points and 5 points far outside the bump support."""
    rng = np.random.default_rng(3)
    n = train_x.shape[0]
    jit = train_x.numpy()[rng.choice(n, 40, replace=False)] + rng.normal(scale=0.02, size=(40, train_x.shape[1])).astype(np.float32)
    far = train_x.numpy()[:5] + 5.0
    post_x = torch.from_numpy(np.concatenate([test_x.numpy(), jit, far]).astype(np.float32))
    out = dict(post_x=post_x.numpy(), outputscale=np.float64(0.7), noise=np.float64(1e-2))
    for tag, k, eps, kappa, modes, bump in [("k10", 10, 0.05, 0.5, 50, (1.0, 0.01)), ("k50", 50, 0.5, 0.5, 20, (3.0, 1.0))]:
        out[tag + "_cfg"] = np.array([k, eps, kappa, modes, bump[0], bump[1]], np.float64)
        for norm in ("symmetric", "randomwalk"):
            for nu in (1, 2):
                torch.set_default_dtype(torch.float64)
                r = reference_posterior(dense, tu, train_x, train_y, post_x, k, eps, kappa, nu, modes, bump, norm, 0.7, 1e-2)
                torch.set_default_dtype(torch.float32)
                for key, v in r.items():
                    if key in ("knn_D", "knn_I", "within", "gap", "evals") and not (norm == "symmetric" and nu == 1):
                        continue
                    name = f"{tag}_{key}" if key in ("knn_D", "knn_I", "within", "gap", "evals") else f"{tag}_{norm}_nu{nu}_{key}"
                    out[name] = v
                print(tag, norm, nu, "gap behind the kept block %.3e" % r["gap"], "|mean| %.3f" % np.abs(r["mean"]).max(),
                      "within", int(r["within"].sum()))
    np.savez_compressed(os.path.join(HERE, "dumbbell_posterior.npz"), **out)


def main():
    dense = _load("ref_dense_operators", f"{REF}/test/_dense_operators.py")
    tu = _load("ref_torch_utils", f"{REF}/manifold_gp/utils/torch_utils.py")
    ld = _load("ref_load_dataset", f"{REF}/manifold_gp/utils/load_dataset.py")

    # ---- dataset (load_dataset.py:10-18 without the importlib.resources lookup) ----
    data = ld.get_data(f"{REF}/manifold_gp/data/dumbbell.msh", "Nodes", "Elements")
    vertices = data["Nodes"][:, 1:-1]
    edges = data["Elements"][:, -2:].astype(int) - 1
    truth, _ = ld.groundtruth_from_samples(vertices, edges)
    sampled_x = torch.from_numpy(vertices).float()
    sampled_y = torch.from_numpy(truth).float()
    if "--only-posterior" not in sys.argv:
        np.savez_compressed(os.path.join(HERE, "dumbbell.npz"),
                            x=sampled_x.numpy(), y=sampled_y.numpy(), segments=edges.astype(np.int32))

    # ---- split (test_laplacian.py:20-23) ----
    torch.manual_seed(1337)
    num_test = 10
    test_idx = torch.zeros(sampled_x.shape[0]).scatter_(0, torch.randperm(sampled_x.shape[0])[:num_test], 1).bool()
    train_x, test_x = sampled_x[~test_idx].contiguous(), sampled_x[test_idx].contiguous()
    train_y, test_y = sampled_y[~test_idx].contiguous(), sampled_y[test_idx].contiguous()
    n = train_x.shape[0]
    torch.manual_seed(7)
    probes = torch.randn(n, 4)

    posterior_goldens(dense, tu, train_x, train_y, test_x)
    if "--only-posterior" in sys.argv:
        return

    for tag, k, self_loops, eps, kappa, nu_list, modes, bump in [
        ("k50_noloop", 50, False, 0.5, 0.5, (1, 2, 3), 20, (3.0, 1.0)),   # test_laplacian.py:31-50
        ("k10_loop", 10, True, 0.05, 0.5, (1, 2), 50, (1.0, 0.01)),       # kernel defaults, data-scaled eps
    ]:
        D, I = oknn.knn_search(train_x.numpy(), train_x.numpy(), k)
        idx_np, val_np = oknn.knn_graph_from_search(D, I, n)
        Dt, It = oknn.knn_search(train_x.numpy(), test_x.numpy(), k)
        idx, val = torch.from_numpy(idx_np), torch.from_numpy(val_np)
        gb = torch.tensor([[eps]], dtype=torch.float32)
        ls = torch.tensor([[kappa]], dtype=torch.float32)
        out = dict(train_x=train_x.numpy(), train_y=train_y.numpy(), test_x=test_x.numpy(), test_y=test_y.numpy(),
                   test_mask=test_idx.numpy(), knn_D=D, knn_I=I.astype(np.int32), knn_test_D=Dt,
                   knn_test_I=It.astype(np.int32), edge_index=idx_np.astype(np.int32), edge_value=val_np,
                   probes=probes.numpy(), eps=np.float32(eps), kappa=np.float32(kappa), k=np.int32(k),
                   self_loops=np.bool_(self_loops), modes=np.int32(modes), bump=np.float32(bump))
        for norm in ("symmetric", "randomwalk"):
            L, adj_un, deg_un, adj, deg = dense.graph_laplacian(idx, val, gb, n, normalization=norm,
                                                                self_loops=self_loops)
            p = f"{norm}_"
            out[p + "degree_unnorm"] = deg_un.numpy()
            out[p + "degree"] = deg.numpy()
            out[p + "diag"] = L.diag().numpy()
            out[p + "mv"] = torch.mv(L, train_y).numpy()                    # _test_functions.py:13
            out[p + "mvT"] = torch.mv(L.T, train_y).numpy()                 # :25
            out[p + "mm"] = torch.mm(L, probes).numpy()
            out[p + "mmT"] = torch.mm(L.T, probes).numpy()
            r, c = idx[0, :64], idx[1, :64]
            out[p + "offdiag64"] = L[r, c].numpy()
            # per-edge views of the dense matrices the reference's twin returns (test/_dense_operators.py:7-24):
            # W (graph_laplacian_operator.py:54-56), A (:73-75) and every off-diagonal entry of L, both triangles
            ra, ca = idx[0].long(), idx[1].long()
            out[p + "adjacency_unnorm_edges"] = adj_un[ra, ca].numpy()
            out[p + "adjacency_edges"] = adj[ra, ca].numpy()
            out[p + "offdiag"] = L[ra, ca].numpy()
            out[p + "offdiagT"] = L[ca, ra].numpy()
            for nu in nu_list:
                Q = dense.matern_precision(L, nu, ls, deg if norm == "randomwalk" else None)
                out[p + f"Q{nu}_mv"] = torch.mv(Q, train_y).numpy()
                out[p + f"Q{nu}_mm"] = torch.mm(Q, probes).numpy()
                if nu == nu_list[0]:
                    noise = torch.tensor(1e-2)
                    outputscale = torch.tensor(0.7)
                    out[p + "Qscaled_mv"] = torch.mv(dense.matern_scaled_precision(Q, outputscale), train_y).numpy()
                    out[p + "Qnoisy_mv"] = torch.mv(dense.matern_noisy_precision(Q, noise), train_y).numpy()
                    torch.manual_seed(11)
                    mask = torch.zeros(n).scatter_(0, torch.randperm(n)[: n // 10], 1).bool()
                    S = dense.matern_labeled_precision(Q.double(), mask)
                    out[p + "schur_mask"] = mask.numpy()
                    out[p + "schur_mv"] = (S @ train_y[mask].double()).float().numpy()
                    out[p + "solve"] = torch.linalg.solve(Q.double(), train_y.double()).float().numpy()
            # ---- gradients (test/_test_functions.py:59-104 `test_grad`, `test_ml`): torch autograd through
            # the reference's dense operators, evaluated in float64
            torch.set_default_dtype(torch.float64)
            e_ = torch.tensor([[eps]], dtype=torch.float64, requires_grad=True)
            Lg = dense.graph_laplacian(idx, val.double(), e_, n, normalization=norm, self_loops=self_loops)[0]
            (torch.mm(Lg.T, train_y.double().view(-1, 1)).sum()).backward()                 # test_grad's loss
            out[p + "grad_eps_sum_LTv"] = np.float64(e_.grad.item())
            e_ = torch.tensor([[eps]], dtype=torch.float64, requires_grad=True)
            Lg = dense.graph_laplacian(idx, val.double(), e_, n, normalization=norm, self_loops=self_loops)[0]
            ((probes.double() * torch.mm(Lg, probes.double())).sum()).backward()
            out[p + "grad_eps_quadform"] = np.float64(e_.grad.item())
            # hyper-parameters chosen so that noise * |Q2| < 1 (the Neumann noise model of
            # noise_wrapper_operator.py:22 is meant for that regime)
            nu_ml = 2 if tag == "k50_noloop" else 1
            e_ = torch.tensor([[eps]], dtype=torch.float64, requires_grad=True)
            k_ = torch.tensor([[kappa]], dtype=torch.float64, requires_grad=True)
            s_ = torch.tensor(0.7, dtype=torch.float64, requires_grad=True)
            z_ = torch.tensor(1e-3, dtype=torch.float64, requires_grad=True)
            Lg, _, _, _, deg_g = dense.graph_laplacian(idx, val.double(), e_, n, normalization=norm, self_loops=self_loops)
            Qg = dense.matern_precision(Lg, nu_ml, k_, deg_g if norm == "randomwalk" else None)
            Q3 = dense.matern_noisy_precision(Qg * s_, z_)          # riemann_gp.py:32-39: scale multiplies
            yv = train_y.double()
            loss = 0.5 * (torch.dot(yv, torch.mv(Q3, yv)) - torch.logdet(Q3) + n * np.log(2 * np.pi))   # test_ml
            loss.backward()
            out[p + "ml_nu"] = np.int32(nu_ml)
            out[p + "ml_loss"] = np.float64(loss.item())
            out[p + "ml_quad"] = np.float64(torch.dot(yv, torch.mv(Q3, yv)).item())
            out[p + "ml_logdet"] = np.float64(torch.logdet(Q3).item())
            out[p + "ml_grads"] = np.array([e_.grad.item(), k_.grad.item(), s_.grad.item(), z_.grad.item()])
            torch.set_default_dtype(torch.float32)
            # ---- spectrum / features (riemann_kernel.py:121-136 on the reference's dense L_sym) ----
            if norm == "symmetric":
                Lsym, deg_sym, deg_un_sym = L, deg, deg_un
            evals, evecs = torch.linalg.eigh(Lsym)
            evals, evecs = evals[:modes].clone(), evecs[:, :modes].clone()
            out[p + "evals_raw"] = evals.numpy().copy()
            evals[0] = 0.0
            evecs = evecs * deg_sym.pow(-0.5).view(-1, 1)
            evecs = torch.nn.functional.normalize(evecs, p=2, dim=0)
            nu = nu_list[0]
            sd = (2 * nu / ls.square() + evals).pow(-nu)
            sd = sd / sd.sum()
            Z = (sd * n).sqrt() * evecs
            out[p + "evals"] = evals.numpy()
            out[p + "features_gram_64"] = (Z[:64] @ Z[:64].T).numpy()
            out[p + "features_diag"] = (Z * Z).sum(-1).numpy()
            # ---- out of sample (test/_test_functions.py:134-150, dense) ----
            ev, ei = torch.from_numpy(Dt), torch.from_numpy(It)
            T = ei.shape[0]
            rows = torch.arange(T).repeat_interleave(ei.shape[1])
            cols = ei.reshape(-1)
            adj_ext_un = torch.sparse_coo_tensor(torch.stack([rows, cols]), ev.reshape(-1).div(-4 * gb.square()).exp().squeeze(),
                                                 (T, n)).to_dense()
            deg_ext_un = adj_ext_un.sum(dim=1)
            adj_ext = torch.mm(deg_ext_un.pow(-1).diag(), torch.mm(adj_ext_un, deg_un.pow(-1).diag()))
            deg_ext = adj_ext.sum(dim=1)
            if norm == "symmetric":
                ext = torch.mm(deg_ext.pow(-0.5).diag(), torch.mm(adj_ext, deg.pow(-0.5).diag()))
            else:
                ext = torch.mm(deg_ext.pow(-1.0).diag(), adj_ext)
            sd2 = (2 * nu / ls.square() + evals).pow(-nu)
            sd2 = sd2 / (1 - evals * gb.square()).square()
            sd2 = sd2 / sd2.sum()
            sd2 = sd2 * n
            Zext = sd2.sqrt() * torch.mm(ext, evecs)
            d1 = ev[:, 0].sqrt()
            b = tu.bump_function(d1, torch.tensor(bump[0] * eps), bump[1])
            out[p + "oos_gram"] = (Zext @ Z[:64].T).numpy()           # rotation-invariant
            out[p + "oos_bump"] = b.numpy()
            out[p + "oos_features_abs_col1"] = Zext[:, 1].abs().numpy()
            # ---- the same imported reference functions on .double() inputs: goldens whose own round-off is
            # negligible, so the HIP path is held to ITS fp32 round-off and not to that of an fp32 eigh
            torch.set_default_dtype(torch.float64)
            gb64, ls64 = gb.double(), ls.double()
            L64, adj_un64, deg_un64, adj64, deg64 = dense.graph_laplacian(idx, val.double(), gb64, n, normalization=norm,
                                                                          self_loops=self_loops)
            out[p + "degree_unnorm_f64"] = deg_un64.numpy()
            out[p + "degree_f64"] = deg64.numpy()
            out[p + "diag_f64"] = L64.diag().numpy()
            out[p + "adjacency_unnorm_edges_f64"] = adj_un64[ra, ca].numpy()
            out[p + "adjacency_edges_f64"] = adj64[ra, ca].numpy()
            out[p + "offdiag_f64"] = L64[ra, ca].numpy()
            y64, P64 = train_y.double(), probes.double()
            out[p + "mv_f64"] = torch.mv(L64, y64).numpy()
            out[p + "mvT_f64"] = torch.mv(L64.T, y64).numpy()
            for nu_ in nu_list:
                Q64 = dense.matern_precision(L64, nu_, ls64, deg64 if norm == "randomwalk" else None)
                out[p + f"Q{nu_}_mv_f64"] = torch.mv(Q64, y64).numpy()
                out[p + f"Q{nu_}_mm_f64"] = torch.mm(Q64, P64).numpy()
                if nu_ == nu_list[0]:
                    out[p + "Qscaled_mv_f64"] = torch.mv(dense.matern_scaled_precision(Q64, torch.tensor(0.7)), y64).numpy()
                    out[p + "Qnoisy_mv_f64"] = torch.mv(dense.matern_noisy_precision(Q64, torch.tensor(1e-2)), y64).numpy()
                    out[p + "schur_mv_f64"] = (dense.matern_labeled_precision(Q64, mask) @ y64[mask]).numpy()
                    out[p + "solve_f64"] = torch.linalg.solve(Q64, y64).numpy()
            if norm == "symmetric":
                Lsym64, deg_sym64, deg_un_sym64 = L64, deg64, deg_un64
            ev64, U64 = torch.linalg.eigh(Lsym64)
            ev64, U64 = ev64[:modes + 8].clone(), U64[:, :modes + 8].clone()
            out[p + "evals_raw_f64"] = ev64.numpy().copy()            # modes + 8: the gap behind the kept block
            ev64, U64 = ev64[:modes].clone(), U64[:, :modes].clone()
            ev64[0] = 0.0
            U64 = U64 * deg_sym64.pow(-0.5).view(-1, 1)
            U64 = torch.nn.functional.normalize(U64, p=2, dim=0)
            sd64 = (2 * nu / ls64.square() + ev64).pow(-nu)
            sd64 = sd64 / sd64.sum()
            Z64 = (sd64 * n).sqrt() * U64
            out[p + "features_gram_64_f64"] = (Z64[:64] @ Z64[:64].T).numpy()
            out[p + "features_diag_f64"] = (Z64 * Z64).sum(-1).numpy()
            aeu = torch.sparse_coo_tensor(torch.stack([rows, cols]), ev.double().reshape(-1).div(-4 * gb64.square()).exp().squeeze(),
                                          (T, n)).to_dense()
            deu = aeu.sum(dim=1)
            ae = torch.mm(deu.pow(-1).diag(), torch.mm(aeu, deg_un_sym64.pow(-1).diag()))
            de = ae.sum(dim=1)
            if norm == "symmetric":
                ext64 = torch.mm(de.pow(-0.5).diag(), torch.mm(ae, deg_sym64.pow(-0.5).diag()))
            else:
                ext64 = torch.mm(de.pow(-1.0).diag(), ae)
            sd2 = (2 * nu / ls64.square() + ev64).pow(-nu)
            sd2 = sd2 / (1 - ev64 * gb64.square()).square()
            sd2 = sd2 / sd2.sum() * n
            Zext64 = sd2.sqrt() * torch.mm(ext64, U64)
            out[p + "oos_gram_f64"] = (Zext64 @ Z64[:64].T).numpy()
            out[p + "oos_bump_f64"] = tu.bump_function(ev.double()[:, 0].sqrt(), torch.tensor(bump[0] * eps), bump[1]).numpy()
            torch.set_default_dtype(torch.float32)
        np.savez_compressed(os.path.join(HERE, f"dumbbell_{tag}.npz"), **out)

    # ---- bump function known answers (torch_utils.py:38-41) ----
    xs = torch.linspace(0, 1.2, 25)
    bumps = {f"a{a}_b{b}": tu.bump_function(xs, torch.tensor(a), b).numpy() for a in (0.5, 1.0) for b in (0.01, 1.0)}
    np.savez_compressed(os.path.join(HERE, "bump.npz"), x=xs.numpy(), **bumps)

    # ---- trained hyper-parameters (models/*.pth -> constrained values via softplus) ----
    hp = {}
    for f in ("srmnist_manifold_semisupervised", "srmnist_manifold_supervised", "1D_manifold_semisupervised"):
        sd = torch.load(f"{REF}/models/{f}.pth", weights_only=True, map_location="cpu")
        sp = torch.nn.functional.softplus
        lb = sd["covar_module.base_kernel.raw_graphbandwidth_constraint.lower_bound"]
        hp[f] = dict(
            graphbandwidth=float(sp(sd["covar_module.base_kernel.raw_graphbandwidth"]).item() + lb.item()),
            lengthscale=float(sp(sd["covar_module.base_kernel.raw_lengthscale"]).item()),
            outputscale=float(sp(sd["covar_module.raw_outputscale"]).item()),
            noise=float(sp(sd["likelihood.noise_covar.raw_noise"]).item() + sd["likelihood.noise_covar.raw_noise_constraint.lower_bound"].item()),
        )
    with open(os.path.join(HERE, "hyperparameters.json"), "w") as fh:
        json.dump(hp, fh, indent=1)
    print(json.dumps(hp, indent=1))


if __name__ == "__main__":
    main()
