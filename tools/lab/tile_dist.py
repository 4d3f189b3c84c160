import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
for wname in ("c3", "s5"):
    wl = bench.build_workload(argparse.Namespace(workload=wname, nodes=0, s5_order="morton"), dev, 0, 1)
    g = wl["graph"]
    tp = g.tiles["tile_ptr"].cpu().numpy().astype(np.int64)
    D = np.diff(tp)
    rp = g.rowptr.cpu().numpy().astype(np.int64)
    nt = len(D)
    ent = np.array([rp[min(g.n, (t + 1) * 64)] - rp[t * 64] for t in range(nt)])
    print(wname, "tiles", nt, "D pcts 50/75/90/95/99/max", np.percentile(D, [50, 75, 90, 95, 99, 100]).astype(int),
          "entries pcts", np.percentile(ent, [50, 75, 90, 95, 99, 100]).astype(int))
    for cap in (384, 512, 640, 768, 1024):
        print("   D >", cap, ":", int((D > cap).sum()), "tiles (%.1f %%)" % (100.0 * (D > cap).mean()))
    del wl; torch.cuda.empty_cache()
