"""CPU ORACLE (test infrastructure) -- Matern precision and its wrappers, matrix-free + dense.

Follows manifold_gp/operators/precision_matern_operator.py:26-53,
scale_wrapper_operator.py:27, noise_wrapper_operator.py:22,
schur_complement_operator.py:26-30 and the dense twins test/_dense_operators.py:27-57.
"""
import numpy as np


class PrecisionMaternOracle:
    """Q = (2 nu / kappa^2 I + L)^nu  (x D for random-walk)."""

    def __init__(self, laplacian, nu, lengthscale):
        self.laplacian = laplacian
        self.nu = int(nu)
        self.dtype = laplacian.dtype
        self.lengthscale = self.dtype(lengthscale)
        self.n = laplacian.n

    def matmul(self, rhs):                                    # precision_matern_operator.py:26-37
        dt = self.dtype
        out = np.asarray(rhs, dtype=dt)
        squeeze = out.ndim == 1
        if squeeze:
            out = out[:, None]
        diag = self.lengthscale * self.lengthscale / dt(2 * self.nu)
        for _ in range(self.nu):
            out = out + diag * self.laplacian.matmul(out)
            out = (out / diag).astype(dt)
        if self.laplacian.normalization == "randomwalk":
            out = out * self.laplacian.degree[:, None]
        out = out.astype(dt)
        return out[:, 0] if squeeze else out

    def dense(self):
        return self.matmul(np.eye(self.n, dtype=self.dtype))


class ScaleWrapperOracle:
    def __init__(self, operator, scale, inverse_scale=False):
        self.operator, self.n, self.dtype = operator, operator.n, operator.dtype
        self.scale, self.inverse_scale = self.dtype(scale), inverse_scale

    def matmul(self, rhs):                                    # scale_wrapper_operator.py:27
        out = self.operator.matmul(rhs)
        return (out / self.scale if self.inverse_scale else out * self.scale).astype(self.dtype)

    def dense(self):
        return self.matmul(np.eye(self.n, dtype=self.dtype))


class NoiseWrapperOracle:
    def __init__(self, operator, noise):
        self.operator, self.n, self.dtype = operator, operator.n, operator.dtype
        self.noise = self.dtype(noise)

    def matmul(self, rhs):                                    # noise_wrapper_operator.py:22
        rhs = np.asarray(rhs, dtype=self.dtype)
        Q = self.operator.matmul
        return Q(rhs - self.noise * Q(rhs - self.noise * Q(rhs))).astype(self.dtype)

    def dense(self):
        return self.matmul(np.eye(self.n, dtype=self.dtype))


class SchurComplementOracle:
    """Q_ll - Q_lu Q_uu^-1 Q_ul with the inner solve done densely in fp64 (converged)."""

    def __init__(self, base, mask):
        self.base, self.mask, self.dtype = base, np.asarray(mask, bool), base.dtype
        self.n = int(self.mask.sum())
        self._quu = None

    def matmul(self, rhs):                                    # schur_complement_operator.py:26-30
        rhs = np.asarray(rhs, dtype=self.dtype)
        squeeze = rhs.ndim == 1
        if squeeze:
            rhs = rhs[:, None]
        m = self.mask
        full = np.zeros((self.base.n, rhs.shape[1]), self.dtype)
        full[m] = rhs
        tmp = self.base.matmul(full)                          # Q[:, l] v
        if self._quu is None:
            eye_u = np.zeros((self.base.n, int((~m).sum())), self.dtype)
            eye_u[np.nonzero(~m)[0], np.arange(eye_u.shape[1])] = 1
            self._quu = self.base.matmul(eye_u)[~m].astype(np.float64)
        sol = np.linalg.solve(self._quu, tmp[~m].astype(np.float64)).astype(self.dtype)
        full2 = np.zeros_like(full)
        full2[~m] = sol
        out = tmp[m] - self.base.matmul(full2)[m]
        out = out.astype(self.dtype)
        return out[:, 0] if squeeze else out

    def dense(self):
        return self.matmul(np.eye(self.n, dtype=self.dtype))


# dense twins (test/_dense_operators.py) -- used to cross-check the matrix-free oracles
def dense_matern_precision(L, nu, lengthscale, degree=None):          # :27-33
    n = L.shape[0]
    P = np.linalg.matrix_power(np.eye(n, dtype=L.dtype) * 2 * nu / (lengthscale * lengthscale) + L, nu)
    if degree is not None:
        P = np.diag(degree) @ P
    return P


def dense_labeled_precision(P, mask):                                  # :36-49
    xx = P[np.ix_(mask, mask)]
    xz = P[np.ix_(mask, ~mask)]
    zz = P[np.ix_(~mask, ~mask)]
    zx = P[np.ix_(~mask, mask)]
    return xx - xz @ np.linalg.solve(zz, zx)


def dense_noisy_precision(P, noise):                                   # :56-57
    return P - noise * (P @ P) + noise * noise * (P @ P @ P)
