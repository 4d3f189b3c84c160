"""Lab: the resident-operand kernel (knob 5) run twice on the same finite data / with one NaN row in each operand: where do
the two outputs differ?  (replays tests/test_gpu_parity.py::test_kernel_block_resident_operand_kernel's launch sequence)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
lib = _lib.lib()
n1, n2, m = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (600, 20004, 100)
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(n1 * 5 + n2 * 11 + m)
Z1 = torch.randn(n1, m, generator=g).to(dev); Z2 = torch.randn(n2, m, generator=g).to(dev)
L = lambda a, b, k: lib.mgp_kernel_block(_lib.ptr(a), n1, _lib.ptr(b), n2, m, 0.6, _lib.ptr(k), _lib.stream())
for rep in range(2):
    lib.mgp_kernel_block_set_pipe(5)
    K = torch.full((n1, n2), float("nan"), device=dev)
    for _ in range(3): L(Z1, Z2, K)
    torch.cuda.synchronize()
    lib.mgp_kernel_block_set_pipe(1); K1 = torch.empty_like(K); L(Z1, Z2, K1)
    lib.mgp_kernel_block_set_pipe(6); Kn = torch.full((n1, n2), float("nan"), device=dev); L(Z1, Z2, Kn); torch.cuda.synchronize()
    lib.mgp_kernel_block_set_pipe(5)
    Z2n = Z2.clone(); Z2n[n2 // 2 + 1] = float("nan")
    Z1n = Z1.clone(); Z1n[n1 // 3 + 1] = float("nan")
    L(Z1n, Z2n, Kn); torch.cuda.synchronize()
    expect = torch.zeros(n1, n2, dtype=torch.bool, device=dev); expect[n1 // 3 + 1, :] = True; expect[:, n2 // 2 + 1] = True
    bad = torch.isnan(Kn)
    d = (Kn != K) & ~expect
    print("max |resident - lean|", float((K - K1).abs().max()), "of", float(K1.abs().max()))
    print("rep", rep, "nan pattern ok", bool(torch.equal(bad, expect)), "finite entries that differ:", int(d.sum()))
    if d.any():
        idx = d.nonzero()
        rows, cols = idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist()
        print("  rows", rows[:70], "\n  cols", cols[:70], "n cols", len(cols))
        i, j = idx[0].tolist()
        print("  first", i, j, float(Kn[i, j]), float(K[i, j]), "lean", float(K1[i, j]), " max |diff|", float((Kn - K)[d].abs().max()))
    K2 = torch.full((n1, n2), float("nan"), device=dev); L(Z1, Z2, K2); torch.cuda.synchronize()
    print("   clean rerun equal to first:", bool(torch.equal(K2, K)))
