"""Lab: which output row does register r of lane l of v_mfma_f64_16x16x4_f64 hold?  mgp_gram_f64 on a 64 x 16 block against numpy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
from manifold_gp_amd import _lib
from manifold_gp_amd.solvers import gram_f64
A = torch.randn(64, 16, generator=torch.Generator().manual_seed(1)).cuda()
ref = A.double().cpu().numpy().T @ A.double().cpu().numpy()
for on in (0, 1):
    _lib.lib().mgp_gram_set_mfma(on)
    G = gram_f64(A).cpu().numpy()
    print("mfma", on, "max err", np.abs(G - ref).max())
    if on:
        perm = [int(np.argmin(np.abs(ref - G[i][None, :]).sum(1))) for i in range(16)]
        print("row i of the kernel's output is row", perm, "of the reference")
        permc = [int(np.argmin(np.abs(ref.T - G[:, j][None, :]).sum(1))) for j in range(16)]
        print("column j of the kernel's output is column", permc)
