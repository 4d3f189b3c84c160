#!/usr/bin/env python
"""Time epochs of the semi-supervised (config C4) precision-form training loop: 10 % labelled."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from manifold_gp_amd.utils import manifold_informed_train
from tools import synth

bases = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(bases, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
D1, _ = kern.knn.search(x[:20000], 2)
eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
torch.manual_seed(1337)
labeled = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
labeled[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev),
                  labeled=labeled).to(dev)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
times, losses = [], []
class Rec:
    def step(self, loss):
        torch.cuda.synchronize(); times.append(time.perf_counter()); losses.append(float(loss.detach())); print("epoch", len(times), losses[-1], flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
manifold_informed_train(model, opt, max_iter=2, tolerance=0.0, num_rand_vec=32, max_cholesky=800, cg_tolerance=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2,
                        cg_max_iter=1000, scheduler=Rec())
torch.cuda.synchronize(); t_total = time.perf_counter() - t0
ep = [round((b - a) * 1e3, 1) for a, b in zip([t0] + times[:-1], times)]
print(json.dumps(dict(n=x.shape[0], labelled=int(labeled.sum()), total_s=round(t_total, 2), epoch_ms=ep, losses=losses), indent=1))
