"""Lab: eval() cold, then after a 1 % / 5 % bandwidth change warm and cold: ms, rounds, block products, residuals, eigenvalue agreement."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import manifold_gp_amd as mgp
from tools import synth
warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "manifold"
if which == "manifold":
    x_np, y_np, _ = synth.manifold_784(60000)
    x = torch.from_numpy(x_np).to(dev); eps0, kappa = 0.3, 3.0
else:
    x, y = synth.rmnist_like(600, 100, seed=1337, device=dev); eps0, kappa = 0.255, 1.9
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
def ev(eps, warm):
    kern.warm_start = warm
    kern.initialize(graphbandwidth=eps, lengthscale=kappa)
    torch.cuda.synchronize(); t0 = time.perf_counter(); kern.eval(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, kern.eigen_info, max(kern.eigen_residuals), kern.eigval.clone()
for rep in range(2):
    t, info, r, e0 = ev(eps0, False)
    print("cold eps %.4f: %.1f ms, rounds %d applies %d conv %d, max resid %.2e" % (eps0, t, info[0], info[1], info[2], r))
ev(eps0, True)                                   # leaves a warm block behind
for f in (1.01, 1.05):
    tw, iw, rw, ew = ev(eps0 * f, True)
    ev(eps0, True)
    tc, ic, rc, ec = ev(eps0 * f, False)
    print("eps x %.2f: warm %.1f ms (rounds %d, applies %d, conv %d, resid %.2e) | cold %.1f ms (rounds %d, applies %d, conv %d, resid %.2e) | max |dlambda| / lambda_100 %.2e"
          % (f, tw, iw[0], iw[1], iw[2], rw, tc, ic[0], ic[1], ic[2], rc, float((ew - ec).abs().max() / ec[-1])))
    ev(eps0, True)
