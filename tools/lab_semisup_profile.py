"""Where an epoch of the semi-supervised loop (C4) spends its solves: every cg_solve call (columns, iterations, ms)."""
import os, sys, time, json, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd import solvers
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
torch.manual_seed(1337)
labeled = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
labeled[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev), labeled=labeled).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
log = []
orig = solvers.cg_solve
def traced(desc, rhs, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(desc, rhs, **kw)
    torch.cuda.synchronize()
    log.append((rhs.shape[1] if rhs.dim() == 2 else 1, out[1], (time.perf_counter() - t0) * 1e3, desc.form, desc.pre is not None and desc.post is not None))
    return out
solvers.cg_solve = traced
import manifold_gp_amd.operators.schur_complement_operator as sco
times = []
class Rec:
    def step(self, loss):
        torch.cuda.synchronize(); times.append((time.perf_counter(), len(log), float(loss.detach())))
torch.cuda.synchronize(); t0 = time.perf_counter()
jac = len(sys.argv) > 1 and sys.argv[1] == "jacobi"
with mgp.settings.cg_jacobi_preconditioner(jac):
  manifold_informed_train(model, opt, max_iter=2, tolerance=0.0, num_rand_vec=32, max_cholesky=800, cg_tolerance=1e-2, cg_max_iter=1000, scheduler=Rec())
prev_t, prev_n = t0, 0
for t, nlog, loss in times:
    ent = log[prev_n:nlog]
    by = collections.Counter()
    for C, its, ms, form, _ in ent:
        by[(C, form)] += 1
    print("epoch %.1f ms loss %.5f: %d cg solves, %d iterations total, %.1f ms inside solves; (columns, form) -> count %s; iterations per solve %s"
          % ((t - prev_t) * 1e3, loss, len(ent), sum(e[1] for e in ent), sum(e[2] for e in ent), dict(by), [e[1] for e in ent][:40]))
    prev_t, prev_n = t, nlog
