"""Lab: the S5 posterior-mean system (I + c B^2) x = y, B = tau I + L_sym (symmetric normalisation, nu = 2), solved through its
complex factorisation: 1 / (1 + c b^2) = Re[1 / (1 + i sqrt(c) b)], so x = Re[(I + i sigma B)^-1 y], sigma = sqrt(c) -- a complex
SYMMETRIC system whose spectrum {1 + i sigma b} has condition sqrt(cond(A)) -- by COCG (CG with the unconjugated inner product),
one 2-column product with B per iteration.  Prototype in torch ops over the HIP SpMM; prints iterations / residuals / time."""
import os, sys, time, argparse, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
from manifold_gp_amd.solvers import CgPlan

dev = torch.device("cuda:0")
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
wl = bench.build_workload(argparse.Namespace(workload="s5", nodes=nodes, s5_order="morton"), dev, 0, 1)
desc = wl["desc"]
assert desc.form == 2 and desc.nu == 2 and desc.pre is None
c = desc.noise * desc.scale
sigma = math.sqrt(c)
dB = desc.with_(nu=1, kappa=desc.kappa / math.sqrt(2.0), scale=1.0, form=0, noise=0.0)      # B = (2 nu / kappa^2) I + L_sym
y = wl["y"].view(-1, 1).contiguous()
n = y.shape[0]

def applyA_c(Z):           # Z [n, 4] = (re, im, 0, 0) -> (I + i sigma B) Z
    BZ = dB.apply(Z)
    out = Z.clone()
    out[:, 0] -= sigma * BZ[:, 1]
    out[:, 1] += sigma * BZ[:, 0]
    return out

def cdot(a, b):            # unconjugated a . b of (re, im) columns -> (re, im) as python floats on device tensors
    re = (a[:, 0] * b[:, 0]).sum() - (a[:, 1] * b[:, 1]).sum()
    im = (a[:, 0] * b[:, 1]).sum() + (a[:, 1] * b[:, 0]).sum()
    return torch.complex(re.double(), im.double())

def cscale(alpha, v):      # complex scalar times (re, im) columns
    out = torch.zeros_like(v)
    ar, ai = float(alpha.real), float(alpha.imag)
    out[:, 0] = ar * v[:, 0] - ai * v[:, 1]
    out[:, 1] = ar * v[:, 1] + ai * v[:, 0]
    return out

def cocg(b, tol, max_iter=2000):
    x = torch.zeros(n, 4, device=dev)
    r = torch.zeros(n, 4, device=dev); r[:, 0] = b.view(-1)
    p = r.clone()
    rr = cdot(r, r)
    bn = float(b.norm())
    for it in range(1, max_iter + 1):
        q = applyA_c(p)
        alpha = rr / cdot(p, q)
        x += cscale(alpha, p)
        r -= cscale(alpha, q)
        if float(r[:, :2].norm()) <= tol * bn:
            break
        rr_new = cdot(r, r)
        p = r + cscale(rr_new / rr, p)
        rr = rr_new
    return x, it

for tol in (1e-6, 1e-7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    z, its = cocg(y, tol)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    x = z[:, :1].contiguous()
    r = desc.apply(x) - y
    print("COCG tol %.0e: %d iterations (%d products with B, 2 columns), %.1f ms eager; true residual of (I + c B^2) x = y (fp32 apply): %.3e"
          % (tol, its, its, dt * 1e3, float(r.norm() / y.norm())), flush=True)
plan = CgPlan(desc, 1, tol=1e-6, max_iter=5000, stop_mode=1, check_every=8, refine=3)
plan.solve(y, copy=False); torch.cuda.synchronize(); t0 = time.perf_counter()
xr = plan.solve(y, copy=False); torch.cuda.synchronize()
print("plan CG (refine 3): %d iterations, %.1f ms, true residual %.3e; |x_cocg - x_cg| / |x| = %.3e"
      % (plan.iters, (time.perf_counter() - t0) * 1e3, max(plan.resid), float((x - xr).norm() / xr.norm())))
