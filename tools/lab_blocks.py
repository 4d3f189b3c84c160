import os, sys, numpy as np, torch, scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
class A: workload, nodes, gpus = "c3", 0, 1
wl = bench.build_workload(A(), torch.device("cuda:0"), 0, 1)
g = wl["graph"]
idx = g.edge_index.cpu().numpy()
b0, b1 = idx[0] // 100, idx[1] // 100
far = b0 != b1
print("far edges", far.mean())
A_ = sp.coo_matrix((np.ones(far.sum()), (b0[far], b1[far])), shape=(600, 600)).tocsr()
A_ = A_ + A_.T
deg = np.diff(A_.indptr)
print("block graph: nnz", A_.nnz, "deg mean %.1f median %d max %d" % (deg.mean(), np.median(deg), deg.max()))
w = A_.data
print("edge weight (edges per block pair): mean %.1f median %.1f; top-5%% pairs carry %.2f of far edges" % (w.mean(), np.median(w), np.sort(w)[-len(w)//20:].sum() / w.sum()))
perm = reverse_cuthill_mckee(A_.tocsr(), symmetric_mode=True)
inv = np.empty(600, int); inv[perm] = np.arange(600)
d = np.abs(inv[b0[far]] - inv[b1[far]])
print("block distance after RCM: median %d p90 %d max %d" % (np.median(d), np.percentile(d, 90), d.max()))
d0 = np.abs(b0[far] - b1[far])
print("block distance before: median %d p90 %d" % (np.median(d0), np.percentile(d0, 90)))
# weighted spectral ordering (Fiedler vector of the block graph)
import scipy.sparse.linalg as sla
L = sp.diags(np.asarray(A_.sum(1)).ravel()) - A_
vals, vecs = sla.eigsh(L.asfptype(), k=3, sigma=-1e-3, which="LM")
order = np.argsort(vecs[:, 1]); inv2 = np.empty(600, int); inv2[order] = np.arange(600)
d2 = np.abs(inv2[b0[far]] - inv2[b1[far]])
print("block distance after Fiedler ordering: median %d p90 %d" % (np.median(d2), np.percentile(d2, 90)), "eigs", vals)
