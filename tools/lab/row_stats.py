import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch, numpy as np
import bench
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g = wl["graph"]
rp = g.rowptr.cpu().numpy().astype(np.int64)
ln = np.diff(rp)
print("rows", len(ln), "mean", ln.mean(), "pcts 50/90/99/99.9/max", np.percentile(ln, [50, 90, 99, 99.9, 100]))
n = len(ln); nt = -(-n // 64)
pad = np.zeros(nt * 64, dtype=np.int64); pad[:n] = ln
t = pad.reshape(nt, 64)
tmax = t.max(1); tsum = t.sum(1)
print("tile entries mean", tsum.mean(), "max", tsum.max(), " tile max-row mean", tmax.mean(), "pcts", np.percentile(tmax, [50, 90, 99, 100]))
w = t.reshape(nt, 4, 16).max(2)   # per-wave max row
print("per-wave max row mean", w.mean(), " per-wave mean row", t.mean())
