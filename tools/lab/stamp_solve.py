"""Lab: GPU-side timeline of one C3 CG solve WITHOUT a profiler.  Needs a library built with -DMGP_STAMP (block 0 of every
kernel of the solve leaves wall_clock64 at its start and end behind the CG state words):
    python tools/lab/stamp_solve.py build_variants/stamp/libmgp_hip.so"""
import ctypes, os, sys, argparse, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
use_graph = not (len(sys.argv) > 2 and sys.argv[2] == "eager")
plan = CgPlan(wl["desc"], 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=0, use_graph=use_graph)
y = wl["y"].view(-1, 1).contiguous()
for _ in range(12):
    plan.solve(y, copy=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    plan.solve(y, copy=False)
torch.cuda.synchronize()
print("ms per solve (host clock, stamped build): %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
h = ctypes.CDLL(_lib.LIB_PATH)
h.mgp_stamp_enable(1)       # the tile kernel stamps behind CgPlan's state words only when asked (other plans reserve no room)
h.mgp_cg_plan_debug_stamps.restype = ctypes.c_int
h.mgp_cg_plan_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_int)]
buf = (ctypes.c_ulonglong * 256)(); cnt = ctypes.c_int(0)
assert h.mgp_cg_plan_debug_stamps(plan.handle, buf, ctypes.byref(cnt)) == 0
n = cnt.value
vals = [(buf[(n - 1 - i) & 255]) for i in range(min(n, 250))][::-1]       # oldest .. newest of the ring
ev = [(v >> 3, v & 7) for v in vals]
kinds = {0: "spmv  start", 1: "spmv  end", 2: "update start", 3: "update end", 4: "decide"}
# the last solve ends with the last 'decide'
last = max(i for i, e in enumerate(ev) if e[1] == 4)
prev = max([i for i, e in enumerate(ev[:last]) if e[1] == 4] or [max(0, last - 40)])
sol = ev[prev + 1:last + 1]
t0 = sol[0][0]
tick_ns = 10.0        # wall_clock64: 100 MHz
for t, k in sol:
    print("%8.2f us  %s" % ((t - t0) * tick_ns / 1e3, kinds[k]))
print("previous decide -> first kernel of this solve: %.2f us" % ((sol[0][0] - ev[prev][0]) * tick_ns / 1e3))

# per-block start / end of the LAST SpMV launch of the last solve
import numpy as np
g = wl["graph"]
nb = int(_lib.lib().mgp_spmm_dot_blocks_csr(ctypes.byref(wl["lap"].data.csr()), 1))
h.mgp_cg_plan_debug_block_stamps.restype = ctypes.c_int
h.mgp_cg_plan_debug_block_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
bb = (ctypes.c_ulonglong * (2 * nb))()
assert h.mgp_cg_plan_debug_block_stamps(plan.handle, bb, nb) == 0
a = np.array(list(bb), dtype=np.int64).reshape(nb, 2)
st, en = (a[:, 0] - a[:, 0].min()) * 0.01, (a[:, 1] - a[:, 0].min()) * 0.01
dur = en - st
print("blocks %d: start min/median/p90/max %.2f %.2f %.2f %.2f us; duration min/median/p90/max %.2f %.2f %.2f %.2f us; last end %.2f us"
      % (nb, st.min(), np.median(st), np.percentile(st, 90), st.max(), dur.min(), np.median(dur), np.percentile(dur, 90), dur.max(), en.max()))
rp = g.rowptr.cpu().numpy().astype(np.int64)
ent = np.array([rp[min(len(rp) - 1, (t + 1) * 64)] - rp[t * 64] for t in range(nb)])
# launch index -> tile: mgp_xcd_block
per, rem = nb // 8, nb % 8
def lb(pb):
    x, i = pb % 8, pb // 8
    return x * per + min(x, rem) + i
tile_of = np.array([lb(b) for b in range(nb)])
e = ent[tile_of]
print("corr(duration, tile entries) = %.3f; mean duration for tiles <= 4096 entries %.2f us (%d), > 4096 entries %.2f us (%d), > 6000 %.2f us (%d)"
      % (np.corrcoef(dur, e)[0, 1], dur[e <= 4096].mean(), (e <= 4096).sum(), dur[e > 4096].mean(), (e > 4096).sum(),
         dur[e > 6000].mean() if (e > 6000).any() else 0.0, (e > 6000).sum()))
order = np.argsort(en)[-8:]
print("last 8 blocks to end: start, duration, entries:", [(round(float(st[i]), 2), round(float(dur[i]), 2), int(e[i])) for i in order])
