"""Synthetic workloads of SURVEY.md section 8(d) (bench / test infrastructure, CPU numpy).

S3 "RMNIST-like": `bases` smooth random 28x28 blob images (uint8 range) x (1 original + 99 rotations
drawn from U(-45, 45) degrees, bilinear rotation about the centre) flattened to 784 features and
scaled (x - 127.5) / 255, i.e. the recipe of manifold_gp/utils/rotate_mnist.py:11-31 and
load_dataset.py:75-77 with synthetic digits (MNIST itself needs a network download).  Target = the
rotation angle, standardised.  S5: points on a swiss-roll surface in R^3 with N(0, 1e-3) jitter.
S3g "C3-size manifold with a conditioned spectrum": a swiss roll (area-uniform samples) pushed through a fixed
orthonormal map into R^784 with a small ambient jitter -- the workload on which "within 1e-4 of the reference" is
decidable at N = 60 000, d = 784 (the RMNIST-like set's ~600 rotation orbits give >= 128 eigenvalues below 2e-6
lambda_max, so its 100-mode cut is not a function of the data).
"""
import numpy as np


def blob_images(count, rng, size=28):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    imgs = np.zeros((count, size, size), np.float32)
    for i in range(count):
        k = rng.integers(3, 7)
        cx, cy = rng.uniform(6, size - 6, k), rng.uniform(6, size - 6, k)
        sx, sy = rng.uniform(1.5, 4.0, k), rng.uniform(1.5, 4.0, k)
        amp = rng.uniform(0.5, 1.0, k)
        img = sum(a * np.exp(-((xx - x0) ** 2) / (2 * s0 ** 2) - ((yy - y0) ** 2) / (2 * s1 ** 2))
                  for a, x0, y0, s0, s1 in zip(amp, cx, cy, sx, sy))
        imgs[i] = img / img.max() * 255.0
    return np.round(imgs).astype(np.uint8)


def _rotate_batch(img, angles_deg):
    """Bilinear rotation of one image about its centre by each angle (zero fill), vectorised."""
    size = img.shape[0]
    c = (size - 1) / 2.0
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    a = np.deg2rad(angles_deg).astype(np.float32)[:, None, None]
    ys = np.cos(a) * (yy - c) + np.sin(a) * (xx - c) + c
    xs = -np.sin(a) * (yy - c) + np.cos(a) * (xx - c) + c
    y0, x0 = np.floor(ys).astype(np.int32), np.floor(xs).astype(np.int32)
    fy, fx = ys - y0, xs - x0
    pad = np.zeros((size + 2, size + 2), np.float32)
    pad[1:-1, 1:-1] = img

    def at(yi, xi):
        ok = (yi >= -1) & (yi <= size) & (xi >= -1) & (xi <= size)
        return np.where(ok, pad[np.clip(yi + 1, 0, size + 1), np.clip(xi + 1, 0, size + 1)], 0.0)

    out = (at(y0, x0) * (1 - fy) * (1 - fx) + at(y0, x0 + 1) * (1 - fy) * fx +
           at(y0 + 1, x0) * fy * (1 - fx) + at(y0 + 1, x0 + 1) * fy * fx)
    return out


def _rmnist_like_torch(imgs, angles, device):
    """All rotations at once on `device` (bilinear grid_sample about the image centre)."""
    import torch
    import torch.nn.functional as F
    bases, per_base = angles.shape
    out = torch.empty(bases * per_base, 28 * 28, dtype=torch.float32, device=device)
    img_t = torch.from_numpy(imgs).to(device)
    ang_t = torch.from_numpy(np.deg2rad(angles)).to(device)
    step = 256
    for b0 in range(0, bases, step):
        b1 = min(bases, b0 + step)
        a = ang_t[b0:b1].reshape(-1)
        cos, sin = torch.cos(a), torch.sin(a)
        theta = torch.zeros(a.shape[0], 2, 3, device=device)
        theta[:, 0, 0], theta[:, 0, 1], theta[:, 1, 0], theta[:, 1, 1] = cos, -sin, sin, cos
        grid = F.affine_grid(theta, (a.shape[0], 1, 28, 28), align_corners=True)
        src = img_t[b0:b1].repeat_interleave(per_base, dim=0).unsqueeze(1)
        rot = F.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        out[b0 * per_base:b1 * per_base] = rot.reshape(-1, 28 * 28)
    return out


def rmnist_like(bases=600, per_base=100, seed=1337, max_angle=45.0, device=None):
    """Returns x [bases*per_base, 784] f32 in [-0.5, 0.5], y [N] f32 (standardised angle).
    device=None: numpy on the host (returns numpy arrays); device=torch device: the rotations run
    there (returns torch tensors on that device) -- same blobs and angles, bilinear interpolation."""
    rng = np.random.default_rng(seed)
    if device is not None:
        import torch
        imgs = blob_images(bases, rng).astype(np.float32)
        angles = np.empty((bases, per_base), np.float32)
        for b in range(bases):
            angles[b] = np.concatenate([[0.0], rng.uniform(-max_angle, max_angle, per_base - 1)])
        x = _rmnist_like_torch(imgs, angles, device)
        x = (torch.round(x) - 127.5) / 255.0
        ang = torch.from_numpy(angles.reshape(-1)).to(device)
        y = (ang - ang.mean()) / ang.std(unbiased=False)
        return x.contiguous(), y.contiguous()
    imgs = blob_images(bases, rng).astype(np.float32)
    n = bases * per_base
    x = np.empty((n, 28 * 28), np.float32)
    ang = np.empty(n, np.float32)
    for b in range(bases):
        angles = np.concatenate([[0.0], rng.uniform(-max_angle, max_angle, per_base - 1)]).astype(np.float32)
        x[b * per_base:(b + 1) * per_base] = _rotate_batch(imgs[b], angles).reshape(per_base, -1)
        ang[b * per_base:(b + 1) * per_base] = angles
    x = (np.round(x) - 127.5) / 255.0          # uint8 range like the reference's arrays
    y = (ang - ang.mean()) / ang.std()
    return x.astype(np.float32), y.astype(np.float32)


def morton_order(x, bits=10):
    """Permutation that sorts points by the Morton (Z-order) code of their quantised coordinates
    (first three features): consecutive indices are spatial neighbours."""
    q = x[:, :3].astype(np.float64)
    q = (q - q.min(0)) / np.maximum(q.max(0) - q.min(0), 1e-30)
    q = np.minimum((q * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    code = np.zeros(len(x), np.uint64)
    for b in range(bits):
        for a in range(q.shape[1]):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(code, kind="stable")


def swiss_roll(n, seed=1337, order="random"):
    """order: "random" (i.i.d. samples in generation order) or "morton" (the same points stored in
    Z-order, i.e. what a locality-aware loader would hand over)."""
    rng = np.random.default_rng(seed)
    t = 1.5 * np.pi * (1 + 2 * rng.random(n))
    h = 21 * rng.random(n)
    x = np.stack([t * np.cos(t), h, t * np.sin(t)], 1) + rng.normal(scale=1e-3, size=(n, 3))
    y = np.sin(t) + 0.05 * h + rng.normal(scale=0.1, size=n)
    x, y = x.astype(np.float32), y.astype(np.float32)
    if order == "morton":
        p = morton_order(x)
        x, y = np.ascontiguousarray(x[p]), np.ascontiguousarray(y[p])
    return x, y


def manifold_784(n, seed=2024, d=784, height=21.0, jitter=2e-3):
    """S3g: n points of a swiss roll t in [1.5 pi, 4.5 pi] x [0, height] sampled uniformly in AREA (arc length along the
    spiral, so the graph has no density gradient for the alpha = 1 normalisation to remove), embedded isometrically in
    R^d by the first three columns of a seeded random orthogonal matrix, plus N(0, jitter^2) noise in every ambient
    coordinate (the data has full rank d; the squared distances gain ~2 d jitter^2 = 6e-3 against a k = 50 neighbourhood
    radius^2 of ~0.5).  The surface is a flat 89.4 x 21 rectangle: simple Neumann spectrum, relative gaps ~1 % at
    mode 100.  Target: sin(t) + 0.05 h + N(0, 0.1^2), standardised.  Returns x [n, d] f32, y [n] f32, (t, h)."""
    rng = np.random.default_rng(seed)
    tt = np.linspace(1.5 * np.pi, 4.5 * np.pi, 20001)
    arc = np.concatenate([[0.0], np.cumsum(np.sqrt(1.0 + (0.5 * (tt[1:] + tt[:-1])) ** 2) * np.diff(tt))])
    t = np.interp(rng.random(n) * arc[-1], arc, tt)
    h = height * rng.random(n)
    x3 = np.stack([t * np.cos(t), h, t * np.sin(t)], 1)
    q, _ = np.linalg.qr(rng.standard_normal((d, 3)))
    x = (x3 @ q.T).astype(np.float32)
    step = 8192                                              # jitter in slabs: no second n x d float64 array
    for s in range(0, n, step):
        x[s:s + step] += (jitter * rng.standard_normal((min(step, n - s), d))).astype(np.float32)
    y = np.sin(t) + 0.05 * h + rng.normal(scale=0.1, size=n)
    y = (y - y.mean()) / y.std()
    return x, y.astype(np.float32), (t, h)


def bandwidth_rule(knn_d2_first, floor):
    """Notebook rule (examples/RMNIST_supervised_learning.ipynb:123-125): the smallest eps for which
    every node keeps a nearest-neighbour weight >= 1e-4: eps_min = sqrt(max_i d2_i,1nn / (-4 ln 1e-4))."""
    eps_min = float(np.sqrt(np.max(knn_d2_first) / (-4.0 * np.log(1e-4))))
    return max(float(floor), eps_min), eps_min


def dumbbell_resampled(n, path=None):
    """S2 of SURVEY.md section 8(d): the closed dumbbell polyline of the reference's 1-D dataset (1556 nodes, data
    fixture tests/golden/dumbbell.npz = manifold_gp/data/dumbbell.msh) resampled uniformly by arc length to n points;
    y = 2 sin(1.5 * geodesic distance from node 0) (load_dataset.py:100-104; on a closed curve the geodesic is the
    shorter of the two arcs).  Returns x [n, 2] f32, y [n] f32, total length."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dumbbell.npz")
    d = np.load(path)
    pts, seg = d["x"].astype(np.float64), d["segments"].astype(np.int64)
    nbr = [[] for _ in range(len(pts))]
    for a, b in seg:
        nbr[a].append(b)
        nbr[b].append(a)
    order, prev, cur = [0], -1, 0
    while True:
        nxt = [v for v in nbr[cur] if v != prev]
        nxt = nxt[0] if nxt else nbr[cur][0]
        if nxt == 0:
            break
        order.append(nxt)
        prev, cur = cur, nxt
    assert len(order) == len(pts), "the polyline is not one closed curve"
    loop = pts[order + [0]]
    segl = np.linalg.norm(np.diff(loop, axis=0), axis=1)
    cum = np.concatenate([[0.0], np.cumsum(segl)])
    total = cum[-1]
    s = np.arange(n) * (total / n)
    j = np.minimum(np.searchsorted(cum, s, side="right") - 1, len(segl) - 1)
    t = (s - cum[j]) / segl[j]
    x = loop[j] * (1 - t)[:, None] + loop[j + 1] * t[:, None]
    geo = np.minimum(s, total - s)
    y = 2.0 * np.sin(1.5 * geo)
    return x.astype(np.float32), y.astype(np.float32), float(total)
