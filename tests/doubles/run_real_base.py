"""Child process of tests/test_real_base_doubles.py: runs with tests/doubles on sys.path so that `import linear_operator`
and `import gpytorch` succeed (test doubles, see README.md) and manifold_gp_amd/_compat.py takes its REAL-BASE branch:
`class LinearOperator(_HipEntryPoints, linear_operator.operators.LinearOperator)`, `Kernel = gpytorch.kernels.Kernel`.

mode "cpu": construction only (device check stubbed), representation / representation_tree rebuild, MRO, settings.
mode "gpu": the same objects on cuda:0 with compute: base-class matmul / to_dense / diagonal / mT dispatch into the HIP
hooks after a rebuild from the representation, solve / inv_quad_logdet through _HipEntryPoints, the `_solve(rhs,
preconditioner, num_tridiag)` convention, the kernel under gpytorch's registration contract.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import linear_operator  # noqa: E402  (the double)
import gpytorch  # noqa: E402         (the double)
import manifold_gp_amd as mgp  # noqa: E402
from manifold_gp_amd import _compat, _lib  # noqa: E402


def common_checks():
    assert linear_operator.__version__.endswith("double") and gpytorch.__version__.endswith("double")
    assert _compat.HAVE_LINEAR_OPERATOR and _compat.HAVE_GPYTORCH
    mro = _compat.LinearOperator.__mro__
    assert mro[1] is _compat._HipEntryPoints and mro[2] is linear_operator.operators.LinearOperator, mro
    assert _compat.Kernel is gpytorch.kernels.Kernel and _compat.Positive is gpytorch.constraints.Positive
    # settings fall through to the library's values unless overridden here
    assert mgp.settings.max_cholesky_size.value() == 800
    with gpytorch.settings.max_cholesky_size(123):
        assert mgp.settings.max_cholesky_size.value() == 123
    with mgp.settings.max_cholesky_size(5):
        assert mgp.settings.max_cholesky_size.value() == 5
    for cls in (mgp.operators.GraphLaplacianOperator, mgp.operators.PrecisionMaternOperator,
                mgp.operators.ScaleWrapperOperator, mgp.operators.NoiseWrapperOperator,
                mgp.operators.SchurComplementOperator):
        assert issubclass(cls, linear_operator.operators.LinearOperator)
        # solve / inv_quad_logdet / _solve resolve to the HIP entry points, not to the library's
        assert cls.solve is _compat._HipEntryPoints.solve and cls.inv_quad_logdet is _compat._HipEntryPoints.inv_quad_logdet
    assert issubclass(mgp.kernels.RiemannMaternKernel, gpytorch.kernels.Kernel)


def build_ops(val, idx, n, dev):
    O = mgp.operators
    eps, ls = torch.tensor([[0.5]], device=dev), torch.tensor([[1.3]], device=dev)
    lap = O.GraphLaplacianOperator(val, idx, n, eps, "randomwalk", True, False)
    Q = O.PrecisionMaternOperator(lap, 2, ls)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    mask[::3] = True
    return [lap, lap._transpose_nonbatch(), Q, O.ScaleWrapperOperator(Q, torch.tensor(0.7, device=dev), inverse_scale=True),
            O.NoiseWrapperOperator(O.ScaleWrapperOperator(Q, torch.tensor(0.7, device=dev)), torch.tensor(1e-2, device=dev)),
            O.SchurComplementOperator(Q, mask)]


def representation_checks(ops):
    for op in ops:
        rep = op.representation()
        assert rep and all(torch.is_tensor(t) for t in rep), type(op).__name__
        re_op = op.representation_tree()(*rep)              # what Matmul / Solve / InvQuadLogdet do in forward()
        assert type(re_op) is type(op) and tuple(re_op.shape) == tuple(op.shape)
        for k in [k for k in vars(op) if not k.startswith("_")]:
            a, b = getattr(op, k), getattr(re_op, k)
            if torch.is_tensor(a):
                assert torch.is_tensor(b) and a.data_ptr() == b.data_ptr(), (type(op).__name__, k)
            elif isinstance(a, linear_operator.operators.LinearOperator):
                assert type(a) is type(b) and tuple(a.shape) == tuple(b.shape), (type(op).__name__, k)
            else:
                assert a is b or a == b, (type(op).__name__, k)
    lap = ops[0]
    assert lap._nondifferentiable_kwargs["normalization"] == "randomwalk" and "graph" not in lap._kwargs
    assert "idx" in lap._differentiable_kwargs and "graphbandwidth" in lap._differentiable_kwargs
    assert ops[1].transposed is True and ops[1]._kwargs["transposed"] is True


def main_cpu():
    common_checks()
    _lib.require_device = lambda *a: None                    # construction only: no compute follows
    val = torch.rand(6)
    idx = torch.tensor([[0, 0, 1, 2, 3, 4], [1, 2, 2, 3, 4, 5]])
    representation_checks(build_ops(val, idx, 6, torch.device("cpu")))
    print("REAL_BASE_CPU_OK")


def main_gpu():
    common_checks()
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
    x, y = torch.from_numpy(g["train_x"]).to(dev), torch.from_numpy(g["train_y"]).to(dev)
    n = x.shape[0]
    idx = torch.from_numpy(g["edge_index"].astype(np.int64)).to(dev)
    val = torch.from_numpy(g["edge_value"]).to(dev)
    ops = build_ops(val, idx, n, dev)
    representation_checks(ops)
    v = torch.randn(n, 3, device=dev)
    with torch.no_grad(), mgp.settings.cg_tolerance(1e-6), mgp.settings.cg_stop_mode(1), mgp.settings.max_cg_iterations(4000):
        for op in ops:
            vv = v[: op.shape[0]]
            a = op.matmul(vv)                                # library entry point: rebuild from representation, then _matmul
            b = op._matmul(vv)
            assert torch.equal(a, b), type(op).__name__
            assert tuple(op.mT.shape) == tuple(op.shape)
        lap, lapT, Q, Qs, A, S = ops
        # base-class dispatch: diagonal() -> _diagonal, to_dense() -> matmul(eye) -> _matmul
        assert torch.equal(lap.diagonal(), lap._diagonal())
        small = mgp.operators.GraphLaplacianOperator(val, idx, n, torch.tensor([[0.05]], device=dev), "symmetric")
        dense = small.to_dense()
        ref = small._matmul(torch.eye(n, device=dev))
        assert torch.equal(dense, ref) and float((dense - dense.t()).abs().max()) < 1e-5 * float(dense.abs().max())
        # the HIP entry points ahead of the library's in the MRO
        sol = Q.solve(y)
        res = float((Q.matmul(sol) - y).norm() / y.norm())
        assert res < 1e-4, res
        iq, ld = A.inv_quad_logdet(y.unsqueeze(-1), logdet=True)
        assert torch.isfinite(iq).all() and torch.isfinite(ld).all()
        # `_solve(rhs, preconditioner, num_tridiag)` as InvQuadLogdet.forward calls it: (solves, tridiagonals)
        rhs = torch.cat([torch.randint(0, 2, (n, 3), device=dev).float() * 2 - 1, y.view(-1, 1)], 1)
        plain = A._solve(rhs, None)
        both = A._solve(rhs, None, num_tridiag=3)
        assert torch.is_tensor(plain) and isinstance(both, tuple) and torch.equal(plain, both[0])
        assert both[1].shape[0] == 3 and both[1].shape[1] == both[1].shape[2]
    # the kernel under gpytorch's registration contract (riemann_kernel.py:28-63)
    try:
        mgp.kernels.RiemannMaternKernel(nu=1, x=x, graphbandwidth_prior=object())
    except TypeError as e:
        assert "gpytorch.priors.Prior" in str(e)
    else:  # pragma: no cover
        raise AssertionError("a non-Prior graphbandwidth_prior must raise TypeError (riemann_kernel.py:57-58)")
    kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=10, laplacian_normalization="symmetric", num_modes=20,
                                           bump_scale=1.0, bump_decay=0.01,
                                           graphbandwidth_prior=gpytorch.priors.NormalPrior(0.0, 1.0)).to(dev)
    names = dict(kern.named_parameters())
    assert set(names) == {"raw_lengthscale", "raw_graphbandwidth"}, names.keys()
    assert isinstance(kern.raw_graphbandwidth_constraint, gpytorch.constraints.Positive)
    assert "graphbandwidth_prior" in kern._priors
    kern.initialize(graphbandwidth=0.05, lengthscale=0.5)
    assert abs(float(kern.graphbandwidth) - 0.05) < 1e-7 and abs(float(kern.lengthscale) - 0.5) < 1e-6
    kern.graphbandwidth = 0.06
    assert abs(float(kern.graphbandwidth) - 0.06) < 1e-7
    kern.initialize(graphbandwidth=0.05)
    kern.eval()
    K = kern(x, x)
    assert isinstance(K, linear_operator.operators.LowRankRootLinearOperator)
    Zx = kern.features(x)
    assert torch.equal(K.root, Zx)
    xt = torch.from_numpy(g["test_x"]).to(dev)
    Kc = kern(xt, x)
    assert isinstance(Kc, linear_operator.operators.MatmulLinearOperator) and tuple(Kc.shape) == (xt.shape[0], n)
    d = kern(x, x, diag=True)
    assert float((d - (Zx * Zx).sum(-1)).abs().max()) < 1e-5 * float(d.abs().max())
    print("REAL_BASE_GPU_OK")


if __name__ == "__main__":
    {"cpu": main_cpu, "gpu": main_gpu}[sys.argv[1]]()
