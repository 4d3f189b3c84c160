"""Lab: long launches of mgp_kernel_block (6-7 ms each) for clock and MFMA-busy counters.  Run plain for the times, and under
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 tools/lab/clock_kblock.py
for the effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the MFMA pipe's share of those cycles.  KB_ZERO=1: all-zero
operands (the clock the chip holds without the MFMAs' switching power)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
lib = _lib.lib()
n1, n2, m = 32768, 60000, 128
torch.manual_seed(0)
Z1 = torch.randn(n1, m, device="cuda:0"); Z2 = torch.randn(n2, m, device="cuda:0")
if os.environ.get("KB_ZERO") == "1":
    Z1.zero_(); Z2.zero_()
K = torch.empty(n1, n2, device="cuda:0")
st = _lib.stream()
for mode in (0, 2, 3):
    lib.mgp_kernel_block_set_pipe(mode)
    for _ in range(2):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.0, _lib.ptr(K), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.0, _lib.ptr(K), st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print("mode %d: %.0f us per launch = %.1f TFLOP/s (%s operands)" % (mode, us, 2.0 * n1 * n2 * m / us / 1e6,
                                                                       "zero" if os.environ.get("KB_ZERO") == "1" else "random"))
lib.mgp_kernel_block_set_pipe(0)
