"""Lab: CG iteration counts on the C3 precision Q (form 0) -- plain, Jacobi, and on the symmetric normalisation."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import bench
import manifold_gp_amd as mgp
from manifold_gp_amd.solvers import CgPlan
dev = torch.device("cuda:0")
wl = bench.build_workload(argparse.Namespace(workload="c3", nodes=0, s5_order="morton"), dev, 0, 1)
g = wl["graph"]
desc = wl["desc"].with_(scale=1.0, form=0, noise=0.0)
torch.manual_seed(0)
B = torch.randn(g.n, 12, device=dev)
idx = torch.randint(0, g.n - 1, (1, 12), device=dev)
B1 = torch.zeros(g.n, 12, device=dev).scatter_(0, idx, 1.0)
dpre = desc.pre
print("pre range", float(dpre.min()), float(dpre.max()), "kappa", desc.kappa, "nu", desc.nu)
for name, rhs in (("gaussian", B), ("one-hot", B1)):
    for stop_mode, tol in ((0, 1e-2), (1, 1e-2), (1, 1e-6)):
        row = []
        for jac in (False, True):
            plan = CgPlan(desc, 12, tol=tol, max_iter=5000, stop_mode=stop_mode, jacobi=jac)
            plan.solve(rhs)
            row.append(plan.iters)
            plan.close()
        # symmetric normalisation: Q_rw = D^1/2 Q_sym D^1/2  ->  Q_sym y = D^-1/2 b, x = D^-1/2 y
        dsym = desc.with_(pre=None, post=None)
        plan = CgPlan(dsym, 12, tol=tol, max_iter=5000, stop_mode=stop_mode, jacobi=False)
        plan.solve((rhs / dpre.view(-1, 1)).contiguous())
        row.append(plan.iters)
        plan.close()
        print("%-9s stop_mode %d tol %g: iterations plain %d, jacobi %d, symmetric-normalised system %d" % (name, stop_mode, tol, *row))

# ---- factorised solve: Q = scale * D^1/2 B^nu D^1/2, B = tau I + L_sym  ->  nu sequential CG solves with B (1 SpMM per iteration)
import math, time
nu = int(desc.nu)
dB = desc.with_(nu=1, kappa=desc.kappa / math.sqrt(nu), scale=1.0, pre=None, post=None)
for name, rhs in (("gaussian", B), ("one-hot", B1)):
    for stop_mode, tol in ((0, 1e-2), (1, 1e-2), (1, 1e-6)):
        planQ = CgPlan(desc, 12, tol=tol, max_iter=5000, stop_mode=stop_mode, jacobi=False)
        for _ in range(2): xq = planQ.solve(rhs).clone()
        torch.cuda.synchronize(); t0 = time.perf_counter(); xq = planQ.solve(rhs).clone(); torch.cuda.synchronize(); tq = time.perf_counter() - t0
        itq = planQ.iters; planQ.close()
        for tol2 in (tol, tol / 4, tol / 16):
            planB = CgPlan(dB, 12, tol=tol2, max_iter=5000, stop_mode=stop_mode, jacobi=False)
            def fsolve():
                y = (rhs / dpre.view(-1, 1)).contiguous()
                its = []
                for _ in range(nu):
                    y = planB.solve(y).clone(); its.append(planB.iters)
                return y / dpre.view(-1, 1) / desc.scale, its
            for _ in range(2): xf, its = fsolve()
            torch.cuda.synchronize(); t0 = time.perf_counter(); xf, its = fsolve(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
            planB.close()
            rq = (desc.apply(xq) - rhs).norm(dim=0) / rhs.norm(dim=0)
            rf = (desc.apply(xf) - rhs).norm(dim=0) / rhs.norm(dim=0)
            print("%-9s mode %d tol %g: CG on Q %d its %.2f ms true res mean %.2e max %.2e | factorised (tol %g) its %s %.2f ms true res mean %.2e max %.2e"
                  % (name, stop_mode, tol, itq, tq * 1e3, float(rq.mean()), float(rq.max()), tol2, its, tf * 1e3, float(rf.mean()), float(rf.max())))
