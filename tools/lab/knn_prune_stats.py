"""Lab: how many 128 x 128 tile pairs of the k-NN distance matrix can centroid / radius bounds exclude?
For a query tile Q and a point tile P every pair is at least ||c_Q - c_P|| - r_Q - r_P apart; with tau_q the k-th
neighbour distance of row q (here taken from the finished search: the best any scheme could know), P can be dropped
for Q when that lower bound exceeds max_q tau_q.  Also the two-pass scheme: tau from the S nearest tiles only."""
import os, sys, argparse, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from tools import synth
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "rmnist"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
if which == "rmnist":
    x_np, _ = synth.rmnist_like(600, 100, seed=1337)
elif which == "gauss":
    x_np = np.random.default_rng(0).standard_normal((60000, 784)).astype(np.float32)
else:
    rng = np.random.default_rng(0)
    cent = rng.standard_normal((300, 64)) * 4
    x_np = (cent[rng.integers(0, 300, 60000)] + rng.standard_normal((60000, 64))).astype(np.float32)
x = torch.from_numpy(x_np).to(dev)
n, d = x.shape
knn = mgp.utils.NearestNeighbors(x)
D, I = knn.search(x, k)
tau = D[:, -1].double()                     # squared k-th neighbour distance
T = (n + 127) // 128
pad = T * 128 - n
xp = torch.cat([x, x[-1:].expand(pad, d)]) if pad else x
xt = xp.view(T, 128, d).double()
c = xt.mean(1)
r = (xt - c[:, None, :]).norm(dim=2).max(1).values
cd = torch.cdist(c, c)
lb = (cd - r[:, None] - r[None, :]).clamp(min=0) ** 2
taup = torch.cat([tau, tau[-1:].expand(pad)]) if pad else tau
tmax = taup.view(T, 128).max(1).values
keep = lb <= tmax[:, None] * (1 + 1e-5)
print("%s: n %d d %d k %d tiles %d; radius mean %.3f centroid distance median %.3f sqrt(tau) mean %.3f" % (which, n, d, k, T, r.mean(), cd.median(), tau.sqrt().mean()))
print("ideal (tau known): surviving tile pairs %.2f %% (mean %.1f point tiles per query tile, max %d)" % (100 * keep.float().mean(), keep.sum(1).float().mean(), int(keep.sum(1).max())))
for S in (4, 8, 16, 32):
    # pass A: the S nearest point tiles by lower bound; tau_A = k-th smallest true distance among their points
    near = lb.argsort(1)[:, :S]
    surv = []
    for t0 in range(0, T, 16):
        t1 = min(T, t0 + 16)
        q = xp.view(T, 128, d)[t0:t1]                                  # [tt, 128, d]
        pts = xp.view(T, 128, d)[near[t0:t1]].reshape(t1 - t0, S * 128, d)
        dist = torch.cdist(q.double(), pts.double()) ** 2             # [tt, 128, S*128]
        ta = dist.topk(k, dim=2, largest=False).values[:, :, -1]      # [tt, 128]
        tm = ta.max(1).values
        surv.append((lb[t0:t1] <= tm[:, None] * (1 + 1e-5)).sum(1))
    surv = torch.cat(surv).float()
    print("two passes, S = %2d: pass B tiles per query tile mean %.1f max %d -> %.2f %% of the pairs (+ pass A %.2f %%)" % (S, surv.mean(), int(surv.max()), 100 * surv.mean() / T, 100 * S / T))
