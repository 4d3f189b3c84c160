import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import manifold_gp_amd as mgp
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
D1, _ = kern.knn.search(x[:20000], 2)
eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
Q = kern.precision()
torch.manual_seed(1337)
mask = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
mask[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
S = mgp.operators.SchurComplementOperator(Q, mask)
v = y[mask]
from manifold_gp_amd import solvers
orig = solvers.cg_solve
def traced(desc, rhs, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = orig(desc, rhs, **kw)
    torch.cuda.synchronize()
    print("   cg_solve C=%d its=%d %.2f ms" % (rhs.shape[1] if rhs.dim() == 2 else 1, out[1], (time.perf_counter() - t0) * 1e3))
    return out
solvers.cg_solve = traced
import manifold_gp_amd.operators.schur_complement_operator as sco
with mgp.settings.cg_tolerance(1e-4), mgp.settings.cg_stop_mode(1), mgp.settings.max_cg_iterations(5000), mgp.settings.cg_jacobi_preconditioner(True), torch.no_grad():
    for i in range(6):
        w = torch.randn_like(v) if i % 2 else v
        torch.cuda.synchronize(); t0 = time.perf_counter(); sv = S.matmul(w); torch.cuda.synchronize()
        print("matvec %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
