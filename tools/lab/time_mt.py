"""Lab: the matrix-core tile SpMM (spmm_mt_kernel, taken when the CSR carries the dense 16-row tile image) against the library's
other choice on a bench graph: time per launch of Y = L X (mgp_spmm_repeat) and max difference of the results."""
import ctypes, os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
data = (lap.data.relabelled() if os.environ.get("MGP_NO_CHAIN") else lap.data.wide_relabelled()) or lap.data
torch.cuda.synchronize(); t0 = time.perf_counter()
plan = data.mt_plan()
torch.cuda.synchronize()
print("image build %.1f ms: tiles %d steps %d fill %.3f image %.1f MB" % ((time.perf_counter() - t0) * 1e3, plan.tiles, plan.steps, plan.fill, plan.img.numel() * 4 / 1e6), flush=True)
csr = data.csr(wide=True)
for C in [int(a) for a in sys.argv[2:]] or (20, 32, 64, 84, 100, 128, 192, 256):
    X = torch.randn(g.n, C, device=dev)
    res = {}
    for mode in (0, 1):
        lib.mgp_spmm_set_mt_mode(mode)
        Y = torch.full_like(X, float("nan"))
        ms = ctypes.c_float(0.0)
        _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 5, None, _lib.stream()), "repeat")
        best = 1e9
        for _ in range(3):
            _lib.check(lib.mgp_spmm_repeat(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 30, ctypes.byref(ms), _lib.stream()), "repeat")
            best = min(best, ms.value)
        res[mode] = (best / 30 * 1e3, Y.clone(), lib.mgp_spmm_kernel_choice(ctypes.byref(csr), C, 0, 0))
    lib.mgp_spmm_set_mt_mode(1)
    B = bench.spmm_bytes(g.n, g.M, C)
    diff = float((res[0][1] - res[1][1]).abs().max()); scale = float(res[0][1].abs().max())
    print("C %3d  kernel %d: %.1f us   matrix-core tiles (kernel %d): %.1f us (%.0f GB/s algorithmic, %.3f of 8 TB/s)   max |diff| %.2e of %.2e, nan %d"
          % (C, res[0][2], res[0][0], res[1][2], res[1][0], B / res[1][0] / 1e3, B / res[1][0] / 1e3 / 8000, diff, scale, int(torch.isnan(res[1][1]).sum())), flush=True)
