"""Lab: where the 600-point posterior of the manifold_784 workload spends its time (cProfile of model.posterior with the training
cache dropped, device synchronised around every call of interest).  posterior_parts.py"""
import cProfile, io, os, pstats, sys, time, warnings
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import solvers
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from tools import synth
dev = torch.device("cuda:0")
n_all = 60600
x_np, y_np, _ = synth.manifold_784(n_all)
perm = np.random.default_rng(11).permutation(n_all)
te, tr = np.sort(perm[:600]), np.sort(perm[600:])
x, y = torch.from_numpy(x_np[tr]).to(dev), torch.from_numpy(y_np[tr]).to(dev)
xt = torch.from_numpy(x_np[te]).to(dev)
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100, bump_scale=3.0, bump_decay=0.01).to(dev)
kern.initialize(graphbandwidth=0.3, lengthscale=3.0)
model = RiemannGP(x, y, GaussianLikelihood(1e-2).to(dev), ScaleKernel(kern, 1.0).to(dev)).to(dev)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    model.eval()
def sync_wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); r = f(*a, **k); torch.cuda.synchronize(); return r
    setattr(mod, name, g)
for _ in range(3):
    model._cache = None; model.posterior(xt); m = model.posterior_mean
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model._cache = None; model.posterior(xt); m = model.posterior_mean
torch.cuda.synchronize()
print("posterior ms (no instrumentation):", (time.perf_counter() - t0) / 5 * 1e3)
for mod, name in ((solvers, "woodbury"), (solvers, "gram_f64"), (solvers, "kernel_block"), (torch.linalg, "cholesky"), (torch, "cholesky_solve"), (type(kern), "features")):
    sync_wrap(mod, name)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model._cache = None; model.posterior(xt); m = model.posterior_mean
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l[:170] for l in s.getvalue().splitlines()))
import time as _t
for k in (1, 50):
    torch.cuda.synchronize(); t0 = _t.perf_counter()
    for _ in range(5):
        D, I = kern.knn.search(xt, k)
    torch.cuda.synchronize()
    print("search of 600 held-out points, k = %d: %.3f ms" % (k, (_t.perf_counter() - t0) / 5 * 1e3), kern.knn.last_stats)
xj = x[:600] + 1e-3 * torch.randn(600, 784, device=dev)
torch.cuda.synchronize(); t0 = _t.perf_counter()
for _ in range(5):
    D, I = kern.knn.search(xj, 50)
torch.cuda.synchronize()
print("search of 600 jittered training points, k = 50: %.3f ms" % ((_t.perf_counter() - t0) / 5 * 1e3), kern.knn.last_stats)
