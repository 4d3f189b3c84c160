"""Lab: is one loss + gradient evaluation of manifold_informed_train's objective bitwise reproducible from the same state?
determinism.py <sup|semisup>"""
import json, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import manifold_gp_amd as mgp
from manifold_gp_amd._compat import settings
from manifold_gp_amd.models import GaussianLikelihood, RiemannGP, ScaleKernel
from tools import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "sup"
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
D1, _ = kern.knn.search(x[:20000], 2)
eps, _ = synth.bandwidth_rule(D1[:, 1].cpu().numpy(), hp["graphbandwidth"])
kern.initialize(graphbandwidth=eps, lengthscale=hp["lengthscale"])
if mode == "semisup":
    torch.manual_seed(1337)
    labeled = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
    labeled[torch.randperm(x.shape[0], device=dev)[: x.shape[0] // 10]] = True
    model = RiemannGP(x[labeled], y[labeled], GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev), labeled=labeled).to(dev)
else:
    model = RiemannGP(x, y, GaussianLikelihood(hp["noise"]).to(dev), ScaleKernel(kern, hp["outputscale"]).to(dev)).to(dev)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
yy = model.train_targets
for rep in range(4):
    for p in params:
        p.grad = None
    op = model.precision()
    with settings.max_cholesky_size(800), settings.cg_tolerance(1e-2), settings.max_cg_iterations(1000):
        t1 = torch.dot(yy, op.matmul(yy.view(-1, 1)).squeeze(-1))
        t2 = op.inv_quad_logdet(logdet=True)[1]
        loss = 0.5 * (t1 - t2 + yy.shape[0] * math.log(2 * math.pi)) / yy.shape[0]
    loss.backward()
    print(rep, "quad %r logdet %r loss %r" % (t1.item(), t2.item(), loss.item()), "grads", [None if p.grad is None else float(p.grad.reshape(-1)[0]) for p in params])
