import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import manifold_gp_amd as mgp
from manifold_gp_amd.operators import noise_wrapper_operator as nw
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from manifold_gp_amd.utils import manifold_informed_train
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
for mode in (False, True, False, True):
    nw._NEUMANN_FOR_CHAINS[0] = mode
    kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
    D1, _ = kern.knn.search(x[:20000], 2)
    eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
    kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
    model = RiemannGP(x, y, GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev)).to(dev)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-2)
    times, losses = [], []
    class Rec:
        def step(self, loss):
            torch.cuda.synchronize(); times.append(time.perf_counter()); losses.append(float(loss.detach()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    manifold_informed_train(model, opt, max_iter=5, tolerance=0.0, num_rand_vec=100, max_cholesky=800, cg_tolerance=1e-2, cg_max_iter=1000, scheduler=Rec())
    ep = [round((b - a) * 1e3, 1) for a, b in zip([t0] + times[:-1], times)]
    grads = [float(p.grad.reshape(-1)[0]) for p in model.parameters() if p.requires_grad and p.grad is not None]
    print("neumann" if mode else "cg on p(Q)", "epoch_ms", ep, "losses", [round(l, 5) for l in losses], "last grads", [round(g, 5) for g in grads])
