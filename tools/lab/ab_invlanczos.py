"""Lab: stochastic log-determinant of the semi-supervised model's operator, Lanczos over the Schur complement (nested
solves) against Lanczos over its inverse (one factorised full-precision solve per step): values and times at 60k."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import manifold_gp_amd as mgp
from manifold_gp_amd import slq
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
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
A = model.precision()
n = A.shape[0]
for tol in (1e-2, 1e-4):
    for inv in (False, True, False, True):
        slq.INVERSE_LANCZOS[0] = inv
        with mgp.settings.cg_tolerance(tol), mgp.settings.max_cg_iterations(2000), torch.no_grad():
            for probes, steps in ((12, 20), (48, 30)):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ld = float(slq.slq_logdet(A, num_probes=probes, steps=steps))
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print("cg tol %g  %-22s probes %2d steps %2d: logdet / n = %.6f   %.1f ms" % (tol, "Lanczos over S^-1" if inv else "Lanczos over S", probes, steps, ld / n, dt * 1e3))
slq.INVERSE_LANCZOS[0] = True
