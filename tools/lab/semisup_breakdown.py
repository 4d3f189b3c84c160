#!/usr/bin/env python
"""Where a semi-supervised (or supervised) training epoch goes: every CgPlan.solve (columns, iterations, relabelled?) and every
Descriptor.apply of the steady epochs, counted; epoch times.  semisup_breakdown.py <sup|semisup> [epochs]   (MGP_CHAIN_MIN_C=48: the
small-column solves in the caller's order)"""
import collections, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tools import bench_stages, synth
from manifold_gp_amd import solvers
from manifold_gp_amd.operators import _descriptor

if os.environ.get("MGP_CHAIN_MIN_C"):
    solvers.CHAIN_SOLVE_MIN_C[0] = int(os.environ["MGP_CHAIN_MIN_C"])
if os.environ.get("MGP_NO_REPEAT"):
    def _rep(self, B, times):
        its, worst = 0, 0
        for _ in range(times):
            B = self.solve(B); its += self.iters; worst = max(worst, self.status)
        return B, its, worst
    solvers.CgPlan.solve_repeated = _rep
if os.environ.get("MGP_NO_REBIND"):
    solvers.REBIND_PLANS[0] = False
mode = sys.argv[1] if len(sys.argv) > 1 else "semisup"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
log = collections.Counter()
its = collections.Counter()
_solve = solvers.CgPlan.solve
def solve(self, B, out=None, copy=True):
    X = _solve(self, B, out=out, copy=copy)
    key = ("solve", self.C, "form%d" % self.desc.form, "nu%d" % self.desc.nu, "chain" if self._rg is not None else "given",
           "masked" if self.desc.pre is not None else "plain", "jacobi" if self.minv is not None else "")
    log[key] += 1; its[key] += self.iters
    return X
solvers.CgPlan.solve = solve
_apply = _descriptor.Descriptor.apply
def apply(self, X):
    log[("apply", X.shape[-1] if X.dim() > 1 else 1, "form%d" % self.form, "nu%d" % self.nu)] += 1
    return _apply(self, X)
_descriptor.Descriptor.apply = apply
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
hp = json.load(open(os.path.join(ROOT, "tests", "golden", "hyperparameters.json")))["srmnist_manifold_semisupervised"]
if os.environ.get("MGP_BUILD_CHAIN"):
    # the chain order up front (otherwise only a product of 48 columns and more builds it)
    from manifold_gp_amd import graph as _graph
    _from_knn = _graph.KnnGraph.from_knn.__func__
    def from_knn(cls, *a, **k):
        g = _from_knn(cls, *a, **k); g.wide_relabelled(); return g
    _graph.KnnGraph.from_knn = classmethod(from_knn)
out = bench_stages.training_stage(x, y, hp, dev, semisup=(mode == "semisup"), epochs=epochs)
print(json.dumps(dict(mode=mode, chain_min_c=solvers.CHAIN_SOLVE_MIN_C[0], epoch_ms_all=out["epoch_ms_all"], epoch_ms=out["epoch_ms"],
                      last_loss=out["last_loss"])))
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print("%6.1f per epoch  %s  iterations per call %.1f" % (v / epochs, k, its[k] / v if k in its else 0))
