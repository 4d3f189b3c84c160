#!/usr/bin/env python
"""Time mgp_kernel_block (K = s Z1 Z2^T on the fp32 MFMA) at posterior shapes.  GPU box only."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])      # A/B: another build of the library
lib = _lib.lib()
res = []
pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.mgp_kernel_block_set_pipe(pipe)
for (n1, n2, m) in ((600, 60000, 100), (1010, 60000, 100), (4096, 60000, 100), (60000, 128, 128), (8192, 8192, 256), (600, 60000, 124), (256, 60000, 100)):
    Z1 = torch.randn(n1, m, device="cuda:0"); Z2 = torch.randn(n2, m, device="cuda:0")
    K = torch.empty(n1, n2, device="cuda:0")
    st = _lib.stream()
    for _ in range(3):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.0, _lib.ptr(K), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.0, _lib.ptr(K), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ref = Z1[:64] @ Z2[:64].t()
    err = float((K[:64, :64] - ref).abs().max() / ref.abs().max())
    res.append(dict(n1=n1, n2=n2, m=m, us=round(us, 1), tflops=round(2.0 * n1 * n2 * m / us / 1e6, 1),
                    write_TBps=round(n1 * n2 * 4 / us / 1e6, 2), rel_err_vs_torch=err))
print(json.dumps(res, indent=1))
