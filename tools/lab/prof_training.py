"""Lab: host-side profile (cProfile) of training epochs at C3 / C4 size after warm epochs.  `semisup` as first argument: 10 % labelled."""
import cProfile, json, os, pstats, sys, time, io
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from manifold_gp_amd.utils import manifold_informed_train
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
D1, _ = kern.knn.search(x[:20000], 2)
eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
SEMI = len(sys.argv) > 1 and sys.argv[1] == "semisup"
if SEMI:
    torch.manual_seed(1337)
    labeled = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
    labeled[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
    model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev), labeled=labeled).to(dev)
else:
    model = RiemannGP(x, y, GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev)).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
def run(n):
    manifold_informed_train(model, opt, max_iter=n, tolerance=0.0, num_rand_vec=32 if SEMI else 100, max_cholesky=800, cg_tolerance=1e-2, cg_max_iter=1000)
    torch.cuda.synchronize()
NE = 4 if SEMI else 10
from manifold_gp_amd import solvers as _sv
if os.environ.get("FACTOR_DIV"):
    _sv.FACTOR_TOL_DIVISOR[0] = float(os.environ["FACTOR_DIV"])
run(2 if SEMI else 3)
t0 = time.perf_counter(); run(NE); print("epoch ms (no profiler): %.2f" % ((time.perf_counter() - t0) / NE * 1e3))
_sv.FACTOR_ROUNDS_LOG = []
run(2)
log = _sv.FACTOR_ROUNDS_LOG; _sv.FACTOR_ROUNDS_LOG = None
import collections
print("factorised solves per epoch %.1f, rounds histogram %s, mean iterations %.1f, mean worst/tol %.2f" % (len(log) / 2, dict(collections.Counter(r for r, _, _ in log)), sum(i for _, i, _ in log) / max(1, len(log)), sum(w for _, _, w in log) / max(1, len(log))))
if os.environ.get("NO_PROFILE"):
    sys.exit(0)
pr = cProfile.Profile(); pr.enable(); run(NE); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(34); print(s.getvalue()[:7000])
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.print_callers("method 'item'"); st.print_callers("built-in method torch.tensor"); st.print_callers("solvers.py:99"); print(s.getvalue()[:9000])
