"""Lab: 60 launches of mgp_kernel_block at one shape (for PMC passes): kblock_one.py n1 n2 m [knob]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
lib = _lib.lib()
n1, n2, m = (int(v) for v in sys.argv[1:4])
lib.mgp_kernel_block_set_pipe(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
torch.manual_seed(0)
Z1 = torch.randn(n1, m, device="cuda:0"); Z2 = torch.randn(n2, m, device="cuda:0"); K = torch.empty(n1, n2, device="cuda:0")
st = _lib.stream()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(60):
    if i == 10: e0.record()
    lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.0, _lib.ptr(K), st)
e1.record(); torch.cuda.synchronize()
print("us per launch %.1f" % (e0.elapsed_time(e1) / 50 * 1e3))
