"""Lab: if the key kernel also wrote the MIN of every 64-key segment of a slab row, how many segments would the select's
one pass over the row still have to read?  (tau = the ~(16 x 75)-th smallest key of the row, what the 1/16 sample gives.)"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tools import synth
dev = torch.device("cuda:0")
for which in ("rmnist", "gauss"):
    if which == "rmnist":
        x, _ = synth.rmnist_like(600, 100, seed=1337, device=dev)
    else:
        x = torch.randn(60000, 784, device=dev)
    n = x.shape[0]
    rows = torch.randperm(n, device=dev)[:512]
    d2 = torch.cdist(x[rows].double(), x.double()) ** 2          # [512, n]
    for mult in (4, 16):
        tau = d2.kthvalue(75 * mult, dim=1).values                # [512]
        nseg = (n + 63) // 64
        pad = nseg * 64 - n
        dd = torch.cat([d2, torch.full((512, pad), float("inf"), device=dev, dtype=d2.dtype)], 1) if pad else d2
        segmin = dd.view(512, nseg, 64).min(2).values
        active = (segmin <= tau[:, None]).sum(1).float()
        print("%s: keys under tau %d -> active 64-key segments per row: mean %.1f of %d (%.1f %%), max %d" % (which, 75 * mult, active.mean(), nseg, 100 * active.mean() / nseg, int(active.max())))
