"""CPU ORACLE (test infrastructure) -- spectral (eigenfeature) side of the Riemann kernel.

Follows manifold_gp/kernels/riemann_kernel.py:79-149, riemann_matern_kernel.py:21-25,
manifold_gp/utils/torch_utils.py:38-41 and the dense out-of-sample formula of
test/_test_functions.py:134-150.
"""
import numpy as np


def bump_function(x, alpha, beta):                            # torch_utils.py:38-41
    x = np.asarray(x)
    y = np.zeros_like(x)
    m = np.abs(x) < alpha
    a2 = alpha * alpha
    y[m] = np.exp(beta / (x[m] * x[m] - a2)) / np.exp(-beta / a2)
    return y


def eval_eigenpairs(lap, num_modes, dtype=None):
    """riemann_kernel.py:117-130: dense eigh of L_sym (always the symmetric matrix), keep the
    first num_modes, eigval[0]=0, eigvec *= D^-1/2, column-normalise (both normalisations)."""
    L = lap.dense_symmetric()
    if dtype is not None:
        L = L.astype(dtype)
    w, U = np.linalg.eigh(L)
    w, U = w[:num_modes].copy(), U[:, :num_modes].copy()
    w[0] = 0.0
    U = U * (lap.degree.astype(U.dtype) ** -0.5)[:, None]
    U = U / np.maximum(np.linalg.norm(U, axis=0, keepdims=True), 1e-12)   # F.normalize(p=2, dim=0)
    return w, U


def spectral_density(eigval, nu, lengthscale):                # riemann_matern_kernel.py:21-22
    return (2.0 * nu / (lengthscale * lengthscale) + eigval) ** (-float(nu))


def features_insample(eigval, eigvec, nu, lengthscale):       # riemann_kernel.py:134-136
    s = spectral_density(eigval, nu, lengthscale)
    s = s / s.sum()
    return np.sqrt(s * eigvec.shape[0])[None, :] * eigvec


def features_oos(lap, eigval, eigvec, nu, lengthscale, edge_value, edge_index,
                 bump_scale, bump_decay):                     # riemann_kernel.py:138-147
    eps = float(lap.eps)
    dt = eigvec.dtype
    T = edge_value.shape[0]
    feats = np.zeros((T, eigvec.shape[1]), dt)
    d1 = np.sqrt(edge_value[:, 0])
    within = d1 < bump_scale * eps
    if within.sum() != 0:
        s = spectral_density(eigval, nu, lengthscale) / (1.0 - eps * eps * eigval) ** 2
        s = s / s.sum()
        s = s * lap.n
        oos = lap.out_of_sample(eigvec, edge_value[within], edge_index[within])
        b = bump_function(d1[within].astype(dt), dt.type(bump_scale * eps), dt.type(bump_decay))
        feats[within] = (np.sqrt(s)[None, :] * oos * b[:, None]).astype(dt)
    return feats


def dense_oos_extension(edge_value, edge_index, degree_unnorm, degree, eps, normalization, n):
    """test/_test_functions.py:135-145: dense Nystrom extension matrix [T, n]."""
    T, k = edge_index.shape
    W = np.zeros((T, n), edge_value.dtype)
    np.add.at(W, (np.repeat(np.arange(T), k), edge_index.reshape(-1)),
              np.exp(edge_value.reshape(-1) / (-4 * eps * eps)))
    dext = W.sum(axis=1)
    A = (W / dext[:, None]) / degree_unnorm[None, :]
    dA = A.sum(axis=1)
    if normalization == "symmetric":
        return (A / np.sqrt(dA)[:, None]) / np.sqrt(degree)[None, :]
    return A / dA[:, None]


def kernel_block(z1, z2):                                     # riemann_kernel.py:92-100
    return z1 @ z2.T
