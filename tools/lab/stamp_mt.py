"""Lab: per-wave timeline of spmm_mt_kernel (build: bash tools/lab/build_variant.sh mtstamp spmm.hip -DMGP_MT_STAMP; run with
MGP_LAB_LIB=tools/lab/_kb_mtstamp/libmgp_hip.so).  Every wave records {start, loop entry, loop exit, end} (100 MHz wall clock), its
block count, HW_ID and XCC_ID; prints what a launch's time is made of."""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from manifold_gp_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("MGP_LAB_LIB", "tools/lab/_kb_mtstamp/libmgp_hip.so"))
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
handle = ctypes.CDLL(_lib.LIB_PATH)
handle.mgp_mt_set_stamp_buffer.argtypes = [ctypes.c_void_p]
data = lap.data.relabelled() or lap.data
csr = data.csr(wide=True)
for C in [int(a) for a in sys.argv[1:]] or (128,):
    NCB = (C + 63) // 64
    W = csr.mt_tiles * NCB
    X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
    def launch():
        _lib.check(lib.mgp_spmm_fused(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 0.0, 1.0, None, None, None, 0.0, 1.0, None, None,
                                      _lib.stream()), "spmm")
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    buf = torch.zeros((W + 8) * 8, dtype=torch.int64, device=dev)
    handle.mgp_mt_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    launch(); torch.cuda.synchronize()
    handle.mgp_mt_set_stamp_buffer(None)
    s = buf.cpu().numpy().reshape(-1, 8)[:W].astype(np.int64)
    t0 = s[:, 0].min()
    T = (s[:, :4] - t0) * 0.01           # us
    NB = s[:, 4]; xcc = s[:, 6] & 15; hw = s[:, 5]
    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
    dur = T[:, 3] - T[:, 0]
    print("C = %d: %d waves, makespan %.1f us; wave start times: median %.1f, 90%% %.1f, max %.1f us" %
          (C, W, T[:, 3].max(), np.median(T[:, 0]), np.percentile(T[:, 0], 90), T[:, 0].max()))
    A = np.stack([np.ones(W), NB], 1)
    coef = np.linalg.lstsq(A, dur, rcond=None)[0]
    print("wave duration = %.2f us + %.3f us per block (blocks: mean %.1f, max %d); prologue median %.2f us, loop per block median %.3f us, "
          "epilogue median %.2f us" % (coef[0], coef[1], NB.mean(), NB.max(), np.median(T[:, 1] - T[:, 0]),
                                        np.median((T[:, 2] - T[:, 1]) / np.maximum(NB, 1)), np.median(T[:, 3] - T[:, 2])))
    # early / late halves of the launch: per-block loop time (contention)
    early = T[:, 0] < np.median(T[:, 0])
    for name, sel in (("first half of the waves", early), ("second half", ~early)):
        print("  %s: loop per block %.3f us, prologue %.2f us" % (name, np.median(((T[:, 2] - T[:, 1]) / np.maximum(NB, 1))[sel]),
                                                                   np.median((T[:, 1] - T[:, 0])[sel])))
    # concurrency over time
    grid = np.arange(0, T[:, 3].max(), 2.0)
    act = [(int(((T[:, 0] <= t) & (T[:, 3] > t)).sum())) for t in grid]
    print("  resident waves every 2 us:", act)
    for x in range(8):
        sel = xcc == x
        if sel.any():
            print("  XCC %d: %d waves, %d blocks, first start %.1f, last end %.1f us" % (x, sel.sum(), NB[sel].sum(), T[sel, 0].min(), T[sel, 3].max()))
    # the 10 waves that end last
    last = np.argsort(-T[:, 3])[:10]
    print("  last waves (end, start, blocks, xcc):", [(round(float(T[i, 3]), 1), round(float(T[i, 0]), 1), int(NB[i]), int(xcc[i])) for i in last])
    mf = NB.sum() * 16 * 32 / 1024 / 2.4e3     # MFMA pipe us if spread evenly over 1024 SIMDs at 2.4 GHz
    print("  matrix pipe work spread evenly: %.1f us" % mf)
