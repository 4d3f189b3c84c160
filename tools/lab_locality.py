import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
class A: workload, nodes, gpus = "c3", 0, 1
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g = wl["graph"]
idx = g.edge_index.cpu().numpy()
d = np.abs(idx[0] - idx[1])
for w in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192):
    print("within %5d: %.3f" % (w, (d <= w).mean()))
rp = g.rowptr.cpu().numpy(); ln = np.diff(rp)
print("row len (padded): mean %.1f median %d p90 %d p99 %d max %d; frac>64 %.3f frac>128 %.3f" % (ln.mean(), np.median(ln), np.percentile(ln, 90), np.percentile(ln, 99), ln.max(), (ln > 64).mean(), (ln > 128).mean()))
# same-base fraction
print("same 100-block:", ((idx[0] // 100) == (idx[1] // 100)).mean())
