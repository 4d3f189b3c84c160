"""Lab: how unevenly a tile's entries fall into dictionary slices.  For the C3 graph and slice widths S: per tile,
sum over slices of the LONGEST per-row run (what a barrier-per-slice walk pays, in entries per lane group) against the mean and
the max row length (what a walk without slices would pay)."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload=sys.argv[1] if len(sys.argv) > 1 else "c3", nodes=0, s5_order="morton"), dev, 0, 1)
g = wl["graph"]
rg = g.relabelled() if g.has_locality_order() else None
rp = (rg.rowptr if rg else g.rowptr).cpu().numpy().astype(np.int64)
t = rg.tiles if rg else g.tiles
lid = t["lid"].cpu().numpy().view(np.uint16).astype(np.int64)
vals = wl["lap"].data.vals_t.cpu().numpy() if rg else wl["lap"].data.vals.cpu().numpy()
n = g.n
ntiles = (n + 63) // 64
for S in (96, 208, 100000):
    tot_barrier, tot_mean, tot_max, tot_wave = 0.0, 0.0, 0.0, 0.0
    for tile in range(0, ntiles, 7):
        r0, r1 = tile * 64, min(n, tile * 64 + 64)
        runs = np.zeros((64, 16), np.int64)
        lens = np.zeros(64, np.int64)
        for r in range(r0, r1):
            e = np.arange(rp[r], rp[r + 1])
            real = vals[e] != 0
            tg = np.minimum(lid[e][real] // S, 15)
            lens[r - r0] = real.sum()
            np.add.at(runs[r - r0], tg, 1)
        tot_barrier += runs.max(0).sum()
        tot_mean += lens.mean()
        tot_max += lens.max()
        # 8 contiguous rows per wave in lockstep, waves independent otherwise (no barrier): per wave sum over slices of its max run
        tot_wave += max(runs[8 * w:8 * w + 8].max(0).sum() for w in range(8))
    print("S %6d: per tile  mean row %.1f  max row %.1f  | sum over slices of the longest run (barrier per slice) %.1f  | slowest wave without "
          "barriers (8 contiguous rows in lockstep) %.1f" % (S, tot_mean / len(range(0, ntiles, 7)), tot_max / len(range(0, ntiles, 7)),
                                                             tot_barrier / len(range(0, ntiles, 7)), tot_wave / len(range(0, ntiles, 7))))
