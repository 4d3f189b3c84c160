"""Lab: phase timeline of workgroup 0 of spmm_dict8_kernel (build with tools/lab/build_stamp_d8.sh first).
Phases: 0 start, 1 prologue issued, 2 first barrier passed, 3 runs found; per stage: 10 top barrier passed, 11 ordinary loads
issued, 12 DMAs issued + walk done, 13 epilogue done, 14 hand-over barrier, 15 next tile's stream decoded, 16 barrier, 17 stage end."""
import ctypes, os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
from manifold_gp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "build_variants", "stamp_d8", "libmgp_hip.so")
import bench
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g, lap = wl["graph"], wl["lap"]
lib = _lib.lib(); lib.mgp_spmm_set_group_hint(g.spmv_lanes)
lib.mgp_spmm_set_dict8_mode(1)      # (off by default)
handle = ctypes.CDLL(_lib.LIB_PATH)
csr = lap.data.csr()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
X = torch.randn(g.n, C, device=dev); Y = torch.empty_like(X)
buf = torch.zeros(2 * 4096, dtype=torch.int64, device=dev)
handle.mgp_d8_set_stamp_buffer.argtypes = [ctypes.c_void_p]
for rep in range(3):
    buf.zero_()
    handle.mgp_d8_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    _lib.check(lib.mgp_spmm_fused(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 0.0, 1.0, None, None, None, 0.0, 1.0, None, None, _lib.stream()), "spmm")
    torch.cuda.synchronize()
b = buf.cpu().numpy()
n = int(b[2 * 4095])
t0 = b[1]
prev = t0
names = {0: "start", 1: "prologue issued", 2: "barrier", 3: "runs found", 10: "stage top: barrier passed", 11: "ordinary loads issued",
         12: "DMAs issued + walk done", 13: "epilogue done", 14: "hand-over barrier", 15: "next stream decoded", 16: "barrier", 17: "stage end"}
for i in range(n):
    code, t = int(b[2 * i]), b[2 * i + 1]
    print("%8.2f us  +%6.2f  %s" % ((t - t0) / 100.0, (t - prev) / 100.0, names.get(code, str(code))))
    prev = t
