"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/mgp_hip.h
declares, the ctypes table mirrors the header, the plugin-surface compatibility layer behaves like
the reference's base classes, and the product path refuses to run without a device."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    hdr = open(os.path.join(ROOT, "include", "mgp_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mgp_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_header_symbol():
    from manifold_gp_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(handle, name), "libmgp_hip.so does not export %s" % name
    # the ctypes signature table and the header name the same entry points
    assert sorted(_lib.SIGNATURES) == names
    assert _lib.lib().mgp_version() >= 100


def test_struct_layouts_match_c_abi(tmp_path):
    """mgp_csr_t / mgp_operator_t / params structs: the ctypes mirrors against what the C compiler makes of include/mgp_hip.h
    (gcc compiles a probe that prints sizeof / offsetof)."""
    import subprocess
    from manifold_gp_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "probe.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "mgp_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mgp_csr_t), sizeof(mgp_operator_t), sizeof(mgp_cg_params_t),
         sizeof(mgp_lanczos_params_t), offsetof(mgp_operator_t, pre), offsetof(mgp_operator_t, nu), offsetof(mgp_csr_t, mt_sptr),
         offsetof(mgp_csr_t, mt_tiles), offsetof(mgp_csr_t, mt_steps));
  return 0;
}
""")
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)])
    c = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    assert c == [ctypes.sizeof(_lib.CsrT), ctypes.sizeof(_lib.OperatorT), ctypes.sizeof(_lib.CgParamsT),
                 ctypes.sizeof(_lib.LanczosParamsT), _lib.OperatorT.pre.offset, _lib.OperatorT.nu.offset, _lib.CsrT.mt_sptr.offset,
                 _lib.CsrT.mt_tiles.offset, _lib.CsrT.mt_steps.offset], c
    assert ctypes.sizeof(_lib.CsrT) == 144 and ctypes.sizeof(_lib.CgParamsT) == 28 and ctypes.sizeof(_lib.LanczosParamsT) == 24


def test_argument_errors_without_gpu():
    """Entry points validate arguments before touching the device."""
    from manifold_gp_amd import _lib
    lib = _lib.lib()
    assert lib.mgp_knn_workspace_bytes(0, 0, 0, 0) == 0
    assert lib.mgp_graph_workspace_bytes(-1, 5, 0) == 0
    assert lib.mgp_spmm_dot_blocks(0, 1) == -1
    assert lib.mgp_spmm_set_group_hint(3) == -1 and lib.mgp_spmm_set_group_hint(8) == 0
    assert lib.mgp_spmm_set_rows_in_flight(3) == -1 and lib.mgp_spmm_set_rows_in_flight(2) == 0
    assert lib.mgp_spmm_fused(None, None, 1, None, 0.0, 1.0, None, None, None, 0.0, 1.0, None, None, None) == -1
    assert lib.mgp_laplacian_build(10, None, None, None, 1.0, 1, None, None, None, None, None, None, None) == -1
    assert lib.mgp_cg_plan_destroy(None) == -1
    with pytest.raises(_lib.MgpError):
        _lib.check(-2, "x")


def test_knn_workspace_sizing_follows_the_filter_switch():
    """mgp_knn_workspace_bytes is host arithmetic (no device): the candidate filter's scratch (sampled keys, log, lists, a
    fail-over slab) replaces the whole-matrix key slab for large matrix-core searches and nothing else; a chunk of the filtered
    pipeline is bounded by its sampled-key slab; the switch itself validates its argument."""
    from manifold_gp_amd import _lib
    L = _lib.lib()
    try:
        assert L.mgp_knn_set_filter(3) != 0 and L.mgp_knn_set_filter(-1) != 0
        L.mgp_knn_set_filter(0)
        slab60 = L.mgp_knn_workspace_bytes(60000, 60000, 784, 50)
        small0 = L.mgp_knn_workspace_bytes(60000, 600, 784, 50)
        lowd0 = L.mgp_knn_workspace_bytes(1000000, 1000000, 3, 64)
        L.mgp_knn_set_filter(1)
        filt60 = L.mgp_knn_workspace_bytes(60000, 60000, 784, 50)
        assert slab60 > 14e9 and 3e9 < filt60 < 6e9, (slab60, filt60)          # 60k x 60k x 4 bytes against ~4.9 GB
        assert L.mgp_knn_workspace_bytes(60000, 600, 784, 50) == small0          # few queries: the slab pipeline either way
        assert L.mgp_knn_workspace_bytes(1000000, 1000000, 3, 64) == lowd0       # d <= 3: the low-dimensional path
        assert L.mgp_knn_workspace_bytes(8000, 8000, 784, 50) == L.mgp_knn_workspace_bytes(8000, 8000, 784, 50)
        # k beyond what the lists serve at the smallest sampling stride (K' x 4 > 1536): the slab again
        L.mgp_knn_set_filter(0)
        big_k0 = L.mgp_knn_workspace_bytes(60000, 60000, 784, 400)
        L.mgp_knn_set_filter(1)
        assert L.mgp_knn_workspace_bytes(60000, 60000, 784, 400) == big_k0
        # 1M points: chunks of 16 384 query rows (4 GiB of sampled keys), far below one list + log per query
        w1m = L.mgp_knn_workspace_bytes(1000000, 1000000, 64, 20)
        assert 8e9 < w1m < 20e9, w1m
        assert L.mgp_knn_last_filter_failover() == -1                             # no search has run in this process
    finally:
        L.mgp_knn_set_filter(1)


def test_product_path_refuses_cpu_tensors():
    import manifold_gp_amd as mgp
    x = torch.randn(50, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        mgp.utils.NearestNeighbors(x)
    idx = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(RuntimeError, match="no CPU path"):
        mgp.operators.GraphLaplacianOperator(torch.rand(2), idx, 3, torch.tensor([[0.5]]))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from manifold_gp_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libmgp_hip.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_settings_context_managers():
    from manifold_gp_amd import settings
    assert settings.cg_tolerance.value() == 1.0 and settings.max_cholesky_size.value() == 800
    with settings.cg_tolerance(1e-2), settings.max_cg_iterations(77):
        assert settings.cg_tolerance.value() == 1e-2
        assert settings.max_cg_iterations.value() == 77
        with settings.cg_tolerance(1e-3):
            assert settings.cg_tolerance.value() == 1e-3
        assert settings.cg_tolerance.value() == 1e-2
    assert settings.cg_tolerance.value() == 1.0 and settings.max_cg_iterations.value() == 1000


def test_protocol_linear_operator_surface():
    """The stand-in base offers the entry points the reference's tests call
    (test/_test_functions.py: matmul, .T, diagonal, solve, inv_quad_logdet, diagonalization)."""
    from manifold_gp_amd._compat import LinearOperator

    class Dense(LinearOperator):
        def __init__(self, A):
            super().__init__(A)
            self.A = A

        def _matmul(self, rhs):
            return self.A @ rhs

        def _size(self):
            return self.A.shape

        def _transpose_nonbatch(self):
            return Dense(self.A.t())

        def _diagonal(self):
            return self.A.diagonal()

    A = torch.tensor([[2.0, 1.0], [0.5, 3.0]])
    op = Dense(A)
    v = torch.tensor([1.0, -1.0])
    assert torch.allclose(op.matmul(v), A @ v)
    assert torch.allclose(op.matmul(v.view(-1, 1)), (A @ v).view(-1, 1))
    assert torch.allclose(op.T.matmul(v), A.t() @ v)
    assert torch.allclose(op.diagonal(), A.diagonal())
    assert tuple(op.shape) == (2, 2)
    for name in ("solve", "inv_quad_logdet", "diagonalization", "to_dense", "logdet", "inv_quad"):
        assert hasattr(op, name)


def test_kernel_parameter_surface_without_device():
    """RiemannMaternKernel's constructor signature (riemann_kernel.py:28-37,
    riemann_matern_kernel.py:13-19) and the Positive (softplus) constraint round trip."""
    import inspect
    import manifold_gp_amd as mgp
    from manifold_gp_amd._compat import Positive
    sig = inspect.signature(mgp.kernels.RiemannMaternKernel.__init__)
    assert list(sig.parameters)[:2] == ["self", "nu"]
    base = inspect.signature(mgp.kernels.riemann_kernel.RiemannKernel.__init__)
    assert list(base.parameters)[1:9] == ["x", "nearest_neighbors", "laplacian_normalization", "num_modes",
                                          "bump_scale", "bump_decay", "graphbandwidth_prior",
                                          "graphbandwidth_constraint"]
    assert base.parameters["nearest_neighbors"].default == 10
    assert base.parameters["laplacian_normalization"].default == "symmetric"
    assert base.parameters["num_modes"].default == 100
    p = Positive()
    v = torch.tensor([0.05, 0.5, 3.0])
    assert torch.allclose(p.transform(p.inverse_transform(v)), v, atol=1e-6)
    assert abs(float(p.transform(torch.zeros(1))) - 0.6931) < 1e-3       # default bandwidth softplus(0)


def test_operator_exports_match_reference_names():
    import manifold_gp_amd as mgp
    assert mgp.operators.__all__ == ["GraphLaplacianOperator", "PrecisionMaternOperator", "ScaleWrapperOperator",
                                     "NoiseWrapperOperator", "SchurComplementOperator"]
    assert mgp.kernels.__all__ == ["RiemannMaternKernel"]
    base = mgp.install_as_manifold_gp(force=True)
    import importlib
    assert importlib.import_module("manifold_gp.kernels").RiemannMaternKernel is mgp.kernels.RiemannMaternKernel
    assert importlib.import_module("manifold_gp.operators").GraphLaplacianOperator is mgp.operators.GraphLaplacianOperator
    assert importlib.import_module("manifold_gp.models").RiemannGP is mgp.models.RiemannGP
    assert callable(importlib.import_module("manifold_gp.utils").manifold_informed_train)
    assert callable(importlib.import_module("manifold_gp.utils").vanilla_train)
    assert callable(importlib.import_module("manifold_gp.utils").test_model)
    # RiemannGP surface of manifold_gp/models/riemann_gp.py:10-75
    for name in ("precision", "modulation", "posterior", "posterior_mean", "posterior_covar", "posterior_stddev",
                 "base_kernel", "eval"):
        assert hasattr(mgp.models.RiemannGP, name), name
    lik = mgp.models.GaussianLikelihood(0.03)
    assert abs(float(lik.noise.detach()) - 0.03) < 1e-6
    with pytest.raises(RuntimeError):                          # host tensors: no CPU path
        mgp.models.RiemannGP(torch.zeros(4, 2), torch.zeros(4), lik, object())
    import sys
    for k in [k for k in sys.modules if k == "manifold_gp" or k.startswith("manifold_gp.")]:
        del sys.modules[k]


def test_bump_function_matches_golden(golden):
    from manifold_gp_amd.utils import bump_function
    g = golden("bump")
    for a in (0.5, 1.0):
        for b in (0.01, 1.0):
            out = bump_function(torch.from_numpy(g["x"]), a, b).numpy()
            np.testing.assert_allclose(out, g[f"a{a}_b{b}"], rtol=2e-5, atol=1e-7)


def test_synthetic_workloads_are_deterministic():
    from tools import synth
    x1, y1 = synth.rmnist_like(3, 10, seed=7)
    x2, y2 = synth.rmnist_like(3, 10, seed=7)
    assert x1.shape == (30, 784) and np.array_equal(x1, x2) and np.array_equal(y1, y2)
    assert x1.min() >= -0.5 and x1.max() <= 0.5
    eps, eps_min = synth.bandwidth_rule(np.array([0.1, 0.4]), 0.05)
    assert abs(eps_min - np.sqrt(0.4 / (-4 * np.log(1e-4)))) < 1e-9 and eps == eps_min


@pytest.mark.parametrize("n", [1, 2, 3, 7, 33, 64, 100, 125, 128, 200])
def test_host_symeig_matches_lapack(n):
    """The small dense fp64 eigensolver inside the block eigensolver (host-only entry of the C-ABI):
    eigenvalues, orthonormality and reconstruction against numpy, including a rank-deficient Gram
    matrix (clustered / zero eigenvalues, the case Rayleigh-Ritz hands it)."""
    from manifold_gp_amd import _lib
    lib = _lib.lib()
    rng = np.random.default_rng(n)
    mats = [rng.standard_normal((n, n))]
    B = rng.standard_normal((n, max(1, n // 3)))
    mats.append(B @ B.T)                                        # rank-deficient PSD
    mats.append(np.diag(np.repeat([1.0, 1.0 + 1e-12, 5.0], -(-n // 3))[:n]))   # (nearly) degenerate diagonal
    mats.append(np.zeros((n, n)))                               # every Householder step is the identity
    mats.append(np.diag(np.arange(1.0, n + 1)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1))   # tridiagonal already
    mats.append(1e150 * rng.standard_normal((n, n)))            # the scaled norms / the guarded sqrt(a^2 + b^2)
    for M in mats:
        A = np.ascontiguousarray(0.5 * (M + M.T))
        ev = np.empty(n)
        V = np.empty((n, n))
        rc = lib.mgp_host_symeig(n, A.ctypes.data, ev.ctypes.data, V.ctypes.data)
        assert rc == 0
        ref = np.linalg.eigvalsh(A)
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.all(np.diff(ev) >= 0)
        np.testing.assert_allclose(ev, ref, rtol=0, atol=1e-12 * scale * n)
        np.testing.assert_allclose(V.T @ V, np.eye(n), rtol=0, atol=1e-12 * n)
        np.testing.assert_allclose(V @ np.diag(ev) @ V.T, A, rtol=0, atol=1e-12 * scale * n)
        # the vector update runs over column slices on host threads: same bits on every call
        ev2 = np.empty(n)
        V2 = np.empty((n, n))
        assert lib.mgp_host_symeig(n, A.ctypes.data, ev2.ctypes.data, V2.ctypes.data) == 0
        assert np.array_equal(ev, ev2) and np.array_equal(V, V2)


def test_struct_fields_mirror_header_and_integration_stub():
    """Field for field: the typedef structs of include/mgp_hip.h, the ctypes Structures of _lib.py and the
    binding shown in INTEGRATION.md (a stale stub makes the C side read past a shorter struct)."""
    from manifold_gp_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "mgp_hip.h")).read()

    def c_fields(name):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\}\s*" + name + r"\s*;", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        return [re.search(r"(\w+)\s*$", decl.strip()).group(1) for decl in body.split(";") if decl.strip()]

    for cname, cls in (("mgp_csr_t", _lib.CsrT), ("mgp_operator_t", _lib.OperatorT), ("mgp_cg_params_t", _lib.CgParamsT),
                       ("mgp_lanczos_params_t", _lib.LanczosParamsT)):
        assert c_fields(cname) == [f[0] for f in cls._fields_], cname
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = re.search(r"class mgp_csr_t\(ctypes\.Structure\):(.*?)\n\n", doc, re.S).group(1)
    assert re.findall(r'\("(\w+)",', stub) == [f[0] for f in _lib.CsrT._fields_]


def test_slq_quadrature_batched_equals_scalar_and_spectral_function():
    """slq._quadrature_log_sum: one batched eigh over the probes == the per-probe routine (including a probe whose
    Krylov space is exhausted), and with a spectral function it integrates log f(theta): for a diagonal-dominant
    tridiagonal built from a known matrix the quadrature of e_1^T log(p(T)) e_1 matches numpy's logm route."""
    import numpy as np
    from manifold_gp_amd.slq import _quadrature_log, _quadrature_log_sum
    rng = np.random.default_rng(0)
    k, P = 20, 12
    a = rng.random((k, P)) + 2.0
    b = rng.random((k, P)) * 0.5
    b[5, 3] = 0.0                                   # exhausted Krylov space: truncated at step 6
    ref = sum(_quadrature_log(a[:, p], b[:, p]) for p in range(P))
    assert abs(_quadrature_log_sum(a, b) - ref) <= 1e-12 * abs(ref)
    s = 0.05
    fun = lambda th: th - s * th * th + s * s * th ** 3          # noqa: E731
    got = _quadrature_log_sum(a[:, :1], b[:, :1], fun)
    T = np.diag(a[:, 0]) + np.diag(b[:k - 1, 0], 1) + np.diag(b[:k - 1, 0], -1)
    w, V = np.linalg.eigh(T)
    want = float(np.sum(V[0] ** 2 * np.log(fun(w))))
    assert abs(got - want) <= 1e-12 * max(1.0, abs(want))


def test_operators_rebuild_from_their_constructor_record(monkeypatch):
    """linear_operator rebuilds an operator as cls(*args, **kwargs) from what its constructor forwarded to
    LinearOperator.__init__ (representation_tree; the reference forwards every argument, graph_laplacian_operator.py:35-43).
    Construct the five operators (no compute: the device check is stubbed out), rebuild each from its record and
    compare every attribute the reference's constructors set."""
    import manifold_gp_amd as mgp
    from manifold_gp_amd import _lib
    monkeypatch.setattr(_lib, "require_device", lambda *a: None)
    O = mgp.operators
    val = torch.rand(3)
    idx = torch.tensor([[0, 0, 1], [1, 2, 2]])
    eps, ls = torch.tensor([[0.5]]), torch.tensor([[1.3]])
    lap = O.GraphLaplacianOperator(val, idx, 3, eps, "randomwalk", True, False)
    Q = O.PrecisionMaternOperator(lap, 2, ls)
    mask = torch.tensor([True, False, True])
    ops = [lap, lap._transpose_nonbatch(), Q, O.ScaleWrapperOperator(Q, torch.tensor(0.7), inverse_scale=True),
           O.NoiseWrapperOperator(Q, torch.tensor(1e-2)), O.SchurComplementOperator(Q, mask)]

    def same(a, b):
        if torch.is_tensor(a):
            return torch.is_tensor(b) and a.data_ptr() == b.data_ptr()
        return a is b or a == b

    for op in ops:
        re_op = type(op)(*op._args, **op._kwargs)
        assert type(re_op) is type(op) and tuple(re_op.shape) == tuple(op.shape)
        names = [k for k in vars(op) if not k.startswith("_")]
        assert names, type(op).__name__
        for k in names:
            assert same(getattr(op, k), getattr(re_op, k)), (type(op).__name__, k)
        # the tensors the record holds are what representation() hands to autograd
        assert all(torch.is_tensor(t) for t in op.representation())
    assert lap._kwargs["normalization"] == "randomwalk" and "graph" not in lap._kwargs
    assert ops[1].transposed is True and ops[1]._kwargs["transposed"] is True


def test_generic_block_lanczos_and_inverse_quadrature_on_cpu():
    """slq._lanczos_block_generic (batched Gram-Schmidt, coefficients read back once) on a small dense SPD matrix on the
    CPU: the tridiagonals' quadrature of log reproduces log det A to the stochastic error, and the same runs over A^-1
    with log(1 / mu) at the Ritz values (what the Schur-complement log-determinant does) give the same number."""
    import numpy as np
    import torch
    from manifold_gp_amd.slq import _lanczos_block_generic, _quadrature_log_sum
    rng = np.random.default_rng(4)
    n, P, steps = 160, 16, 40
    M = rng.normal(size=(n, n))
    A = torch.from_numpy((M @ M.T / n + 0.5 * np.eye(n)).astype(np.float32))
    Ainv = torch.linalg.inv(A.double()).float()

    class Op:
        def __init__(self, mat):
            self.mat, self.shape = mat, mat.shape

        def matmul(self, V):
            return self.mat @ V

    Z = torch.from_numpy(rng.choice([-1.0, 1.0], size=(n, P)).astype(np.float32))
    a, b = _lanczos_block_generic(Op(A), Z, steps)
    ai, bi = _lanczos_block_generic(Op(Ainv), Z, steps)
    want = float(torch.linalg.slogdet(A.double())[1])
    direct = n * _quadrature_log_sum(a, b) / P
    inverse = n * _quadrature_log_sum(ai, bi, lambda mu: 1.0 / np.maximum(mu, 1e-30)) / P
    assert abs(direct - inverse) < 2e-3 * abs(want) + 1e-2, (direct, inverse)
    assert abs(direct - want) < 0.05 * abs(want) + 0.5, (direct, want)


def test_factorised_solve_is_chosen_only_for_unmasked_chains():
    """solvers._factorisable: form 0, nu >= 2, pre / post absent (symmetric) or THE node scaling D^1/2 of the graph
    (random walk); masks folded into pre / post, the noise forms, nu = 1 and refinement requests keep the whole-chain CG."""
    import types
    import torch
    from manifold_gp_amd import solvers
    from manifold_gp_amd.operators._descriptor import Descriptor
    sq = torch.rand(10) + 0.5
    data = types.SimpleNamespace(dsqrt=sq, dinvsqrt=1.0 / sq, graph=types.SimpleNamespace(n=10))
    sym = Descriptor(data, 2, 1.3)
    rw = Descriptor(data, 2, 1.3, pre=sq, post=sq)
    assert solvers._factorisable(sym, {}) and solvers._factorisable(rw, {})
    assert not solvers._factorisable(sym.with_(nu=1), {})
    assert not solvers._factorisable(rw.with_(form=2, noise=0.1), {}) and not solvers._factorisable(rw.with_(form=1, noise=0.1), {})
    mask = torch.ones(10)
    mask[::2] = 0
    assert not solvers._factorisable(rw.masked(mask, mask), {}) and not solvers._factorisable(sym.masked(mask, mask), {})
    assert not solvers._factorisable(rw, {"refine": 2})
    solvers.FACTORISED_SOLVES[0] = False
    try:
        assert not solvers._factorisable(rw, {})
    finally:
        solvers.FACTORISED_SOLVES[0] = True


def test_kernel_block_pp_instruction_stream_respects_its_own_waits():
    """kernel_block_pp's staging loads are inline asm the compiler does not track (features.hip): tools/check_kblock_isa.py
    compiles the file for gfx950 and verifies that nothing touches a destination register of such a load before the kernel's own
    vmcnt wait, and that the tile's 16 stores keep their data registers until the s_nop; for kernel_block_res, whose waits are
    `vmcnt(N)` with N > 0, it replays each loop against an in-order queue of its loads and stores.  The checker itself is
    exercised on doctored streams first."""
    import importlib.util
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_kblock_isa", os.path.join(root, "tools", "check_kblock_isa.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    body = ["\tbuffer_load_dwordx4 v[64:67], v5, s[20:23], s30 offen", "\tv_mfma_f32_32x32x2_f32 v[0:15], v100, v101, v[0:15]",
            "\ts_waitcnt vmcnt(0)", "\tds_write2_b32 v6, v64, v65 offset1:1"] + \
           ["\tbuffer_store_dwordx4 v[%d:%d], v90, s[16:19], s56 offen" % (4 * q, 4 * q + 3) for q in range(16)] + ["\ts_nop 4", "\tv_mov_b32_e32 v0, 0"]
    def kernels(lines):
        out = []
        for ts in (8, 6, 4, 2):
            out += ["_ZN12_GLOBAL__N_115kernel_block_ppILi%dEEEvPKflS2_lifPfliii:" % ts] + lines + ["\t.end_amdhsa_kernel"]
            out += ["_ZN12_GLOBAL__N_116kernel_block_oneILi%dEEEvPKflS2_lifPflii:" % ts] + lines + ["\t.end_amdhsa_kernel"]
        return "\n".join(out)
    assert chk.check(kernels(body)) == []
    early_read = body[:1] + ["\tv_mov_b32_e32 v152, v65"] + body[1:]
    assert any("in-flight" in p for p in chk.check(kernels(early_read)))
    branch = body[:1] + ["\ts_cbranch_vccnz .LBB10_12"] + body[1:]
    assert any("branch" in p for p in chk.check(kernels(branch)))
    clobber = body[:-2] + ["\tv_pk_mul_f32 v[60:61], s[6:7], v[2:3]"] + body[-2:]
    assert any("before the s_nop" in p for p in chk.check(kernels(clobber)))
    assert any("expected the tile's 16" in p for p in chk.check(kernels(body[:10] + body[-2:])))
    # kernel_block_res: waits with vmcnt(N > 0) against an in-order queue of the loop's loads and stores (check_res)
    def res_kernel(wait2=32, wait1=16, extra=(), stores2=32):
        loads = ["\tbuffer_load_dwordx4 v[%d:%d], v1, s[20:23], 0 offen offset:%d" % (40 + 4 * j, 43 + 4 * j, 32 * j) for j in range(3)]
        mf = ["\tv_mfma_f32_32x32x2_f32 v[0:15], v100, v%d, v[0:15]" % (40 + e) for e in range(12)]
        def walk(label, wait, nst, br):
            return loads + ["\ts_waitcnt vmcnt(0)", "%s:" % label, "\ts_waitcnt vmcnt(%d)" % wait] + mf + list(extra) + loads + \
                   ["\tbuffer_store_dword v%d, v90, s[16:19], 0 offen nt" % (r % 16) for r in range(nst)] + ["\t%s %s" % (br, label)]
        return "\n".join(["_ZN12_GLOBAL__N_116kernel_block_resILi5EEEvPKflS2_lfPfliiiii:"] + walk(".LBB9_2", wait2, stores2, "s_cbranch_vccnz") +
                         walk(".LBB9_4", wait1, 16, "s_cbranch_scc0") + ["\t.end_amdhsa_kernel"])
    assert chk.check_res(res_kernel(), expect=1) == []
    assert any("still in flight" in p for p in chk.check_res(res_kernel(wait2=33), expect=1))       # one operation too many allowed
    assert any("still in flight" in p for p in chk.check_res(res_kernel(wait1=19), expect=1))
    assert any("loads /" in p for p in chk.check_res(res_kernel(stores2=31, wait2=31), expect=1))
    assert any("not straight-line" in p for p in chk.check_res(res_kernel(extra=("\ts_cbranch_scc1 .LBB9_9",)), expect=1))
    assert any("expected 29" in p for p in chk.check_res(res_kernel()))
    if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("hipcc not found: only the checker's own logic was tested")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_kblock_isa.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    # spmm_mt_kernel (entered with three blocks of loads in flight): the replay is clean on the compiled stream and catches a wait
    # that lets one operation too many stay in flight, and a missing wait in front of the epilogue
    asm = chk.compile_to_asm(os.path.join(root, "manifold_gp_amd", "csrc", "spmm.hip"), ["-mllvm", "-amdgpu-mfma-vgpr-form=1"])
    assert asm is not None and chk.check_replay(asm, chk.MT_PATTERN, 2, "spmm_mt_kernel") == []
    import re
    k0 = [m.start() for m in re.finditer(r"^_ZN\d+_GLOBAL__N_1\d+spmm_mt_kernelILb0EE.*:", asm, re.M)][0]
    k1 = asm.index(".end_amdhsa_kernel", k0)
    body = asm[k0:k1]
    assert "s_waitcnt vmcnt(17)" in body and "s_waitcnt vmcnt(16)" in body
    loose = asm[:k0] + body.replace("s_waitcnt vmcnt(17)", "s_waitcnt vmcnt(18)", 1) + asm[k1:]
    assert any("still in flight" in p for p in chk.check_replay(loose, chk.MT_PATTERN, 2, "spmm_mt_kernel"))
    last0 = body.rindex("s_waitcnt vmcnt(0)")
    nowait = asm[:k0] + body[:last0] + "s_nop 0" + body[last0 + len("s_waitcnt vmcnt(0)"):] + asm[k1:]
    assert chk.check_replay(nowait, chk.MT_PATTERN, 2, "spmm_mt_kernel") != []
