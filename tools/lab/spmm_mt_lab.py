#!/usr/bin/env python
"""Lab: wide SpMM on the matrix cores over dense 16-row tiles (spmm_mt_lab.hip) against the production SpMM on the C3 graph.
spmm_mt_lab.py [workload] [C ...] -- builds the dense tile image on the CPU, checks Y against fp64, times both."""
import ctypes, os, sys, argparse, subprocess, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE)); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
so = os.path.join(HERE, "libspmm_mt_lab.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "spmm_mt_lab.hip")):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-o", so, os.path.join(HERE, "spmm_mt_lab.hip")])
if len(sys.argv) > 1 and sys.argv[1] == "build":
    sys.exit(0)
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
data = lap.data.relabelled() or lap.data
csr = data.csr()
lab = ctypes.CDLL(so)
n = g.n
gg = data.graph
rowptr = gg.rowptr.cpu().numpy().astype(np.int64); col = gg.col.cpu().numpy().astype(np.int64); vals = data.vals.cpu().numpy()
R = 16
T = (n + R - 1) // R
sptr = np.zeros(T + 1, np.int32); cols = []; imgs = []
for t in range(T):
    r0, r1 = t * R, min(n, (t + 1) * R)
    e0, e1 = rowptr[r0], rowptr[r1]
    c = col[e0:e1]; v = vals[e0:e1]
    rows = np.repeat(np.arange(r0, r1), np.diff(rowptr[r0:r1 + 1])) - r0
    ok = v != 0
    u = np.unique(c[ok]) if ok.any() else np.array([0], np.int64)
    S = (len(u) + 15) // 16 * 4          # steps, padded to whole blocks of four
    m = np.zeros((4 * S, R), np.float32)
    np.add.at(m, (np.searchsorted(u, c[ok]), rows[ok]), v[ok])
    u = np.concatenate([u, np.full(4 * S - len(u), u[-1])])
    cols.append(u.astype(np.int32)); imgs.append(m.reshape(S, 64))       # [s][kq * 16 + i]
    sptr[t + 1] = sptr[t] + S
steps = int(sptr[-1])
dcol = torch.from_numpy(np.concatenate(cols + [np.zeros(192, np.int32)])).to(dev)          # the kernel requests up to 6 blocks / 2 dictionary batches past a tile's end
img = torch.from_numpy(np.concatenate(imgs + [np.zeros((32, 64), np.float32)])).to(dev).contiguous()
sp = torch.from_numpy(sptr).to(dev)
D = np.diff(sptr) * 4
print("16-row tiles %d, steps %d (mean %.1f per tile, max %d), image %.1f MB, dictionary %.1f MB; MFMA pipe floor at C = 128: %.1f us"
      % (T, steps, steps / T, D.max() // 4, img.numel() * 4 / 1e6, dcol.numel() * 4 / 1e6, steps * 8 * 32 / 1024 / 2.4e3), flush=True)
idx = torch.from_numpy(np.stack([np.repeat(np.arange(n), np.diff(rowptr)), col])).to(dev)
A64 = torch.sparse_coo_tensor(idx, torch.from_numpy(vals).to(dev).double(), (n, n)).coalesce()
for C in [int(a) for a in sys.argv[2:]] or (128, 64, 100):
    X = torch.randn(n, C, device=dev); Y = torch.empty_like(X); Y2 = torch.full_like(X, float("nan"))
    ref = torch.sparse.mm(A64, X.double())
    ms = ctypes.c_float(0.0)
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
    _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
    t_ref = ms.value / 30 * 1e3
    best = 1e9
    for rep in range(3):
        rc = lab.lab_spmm_mt(ctypes.c_void_p(sp.data_ptr()), ctypes.c_void_p(dcol.data_ptr()), ctypes.c_void_p(img.data_ptr()), ctypes.c_void_p(X.data_ptr()),
                             ctypes.c_void_p(Y2.data_ptr()), n, T, C, 30, ctypes.byref(ms), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                             img.numel() * 4, dcol.numel() * 4)
        assert rc == 0, rc
        best = min(best, ms.value / 30 * 1e3)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    print("  C %3d: production %.1f us (its Y: raw SpMM? max |Y - ref| %.2e)   matrix-core tiles %.1f us   max |Y2 - ref| %.2e of %.2e, nan %d"
          % (C, t_ref, float((Y.double() - ref).abs().max()), best, float((Y2.double() - ref).abs().max()), sc, int(torch.isnan(Y2).sum())), flush=True)
