#!/usr/bin/env python
"""Three warm 60k x 784 self-searches in the given filter mode (for rocprofv3 --kernel-trace --stats).  Usage: knn_filter_time.py [mode]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
from tools import synth
dev = torch.device("cuda:0")
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
x = x.contiguous()
_lib.lib().mgp_knn_set_filter(mode)
knn = mgp.utils.NearestNeighbors(x)
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); knn.search(x, 50); torch.cuda.synchronize()
    print("search ms %.2f" % ((time.perf_counter() - t0) * 1e3), knn.last_stats, flush=True)

if os.environ.get("MGP_KNN_STAMPS"):
    import ctypes
    h = ctypes.CDLL(_lib.LIB_PATH)
    out = (ctypes.c_uint64 * 8)()
    h.mgp_knn_lab_stamps(out, 1)
    knn.search(x, 50); torch.cuda.synchronize()
    h.mgp_knn_lab_stamps(out, 1)
    names = ["main loop", "bounds+keys+preds", "scan + barrier", "range drawn (atomic) + barrier", "-", "stores issued", "-", "stores acked"]
    waves = 4 * (469 * 470 // 2 + 469 * 30)      # filtered pass + sample pass (both stamp slot 0)
    tot = sum(out)
    for nme, v in zip(names, out):
        print("stamp %-24s %14d cycles  %5.1f %%   per wave %8.0f" % (nme, v, 100.0 * v / tot, v / (4 * 469 * 470 // 2 / 61)), flush=True)

if os.environ.get("MGP_SEL_STAMPS"):
    import ctypes
    h = ctypes.CDLL(_lib.LIB_PATH)
    out = (ctypes.c_uint64 * 8)()
    h.mgp_sel_lab_stamps(out, 1)
    knn.search(x, 50); torch.cuda.synchronize()
    h.mgp_sel_lab_stamps(out, 1)
    names = ["list loaded", "radix select", "candidates collected", "query row staged", "re-rank", "ordered", "-", "-"]
    tot = sum(out)
    for nme, v in zip(names, out):
        print("sel stamp %-24s %14d cycles  %5.1f %%   per row %8.0f" % (nme, v, 100.0 * v / max(tot, 1), v / (60000 / 61)), flush=True)
