"""Lab: out-of-sample k-NN search (600 queries against the 60k x 784 index), repeated: what a prediction call pays."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 101, seed=1337, device=dev)
mask = torch.zeros(x.shape[0], dtype=torch.bool, device=dev); mask[100::101] = True
xt, xq = x[~mask].contiguous(), x[mask].contiguous()
knn = mgp.utils.NearestNeighbors(xt)
for _ in range(3): knn.search(xq, 50)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): knn.search(xq, 50)
torch.cuda.synchronize(); print("oos search of %d queries: %.3f ms" % (xq.shape[0], (time.perf_counter() - t0) / 20 * 1e3))
