#!/usr/bin/env python
"""A/B of mgp_kernel_block: the general kernel (knob 4), the lean one-tile-per-workgroup kernel (0) and the two-half tile walk (2),
plus the walk with every store dropped (3); full-result check against torch fp64 on every shape.  GPU box only."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
lib = _lib.lib()
res = []
if os.environ.get("KB_ZERO") == "1":
    print("ALL-ZERO operands")
shapes = ((600, 60000, 100), (4096, 60000, 100), (8192, 8192, 256), (1000, 50000, 64), (777, 33333, 36), (128, 70001, 16),
          (60000, 128, 128), (131, 257, 8), (600, 60000, 128))
for (n1, n2, m) in shapes:
    torch.manual_seed(n1 + n2 + m)
    Z1 = torch.randn(n1, m, device="cuda:0"); Z2 = torch.randn(n2, m, device="cuda:0")
    if os.environ.get("KB_ZERO") == "1":      # all-zero operands: the clock the chip holds without the MFMAs' switching power
        Z1.zero_(); Z2.zero_()
    st = _lib.stream()
    row = dict(n1=n1, n2=n2, m=m)
    outs = {}
    for mode in (4, 0, 2):
        assert lib.mgp_kernel_block_set_pipe(mode) == 0
        K = torch.full((n1, n2), float("nan"), device="cuda:0")
        for _ in range(3):
            assert lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.7, _lib.ptr(K), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.7, _lib.ptr(K), st)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row[f"us_mode{mode}"] = round(us, 1)
        row[f"tflops_mode{mode}"] = round(2.0 * n1 * n2 * m / us / 1e6, 1)
        outs[mode] = K
    # full check in row slabs against fp64
    worst = 0.0
    scale = 0.0
    for r0 in range(0, n1, 512):
        ref = 1.7 * (Z1[r0:r0 + 512].double() @ Z2.double().t())
        scale = max(scale, float(ref.abs().max()))
        for mode in (4, 0, 2):
            worst = max(worst, float((outs[mode][r0:r0 + 512].double() - ref).abs().max()))
            assert not torch.isnan(outs[mode][r0:r0 + 512]).any(), (n1, n2, m, mode)
    lib.mgp_kernel_block_set_pipe(3)       # timing only: stores dropped
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    Kd = torch.empty(n1, n2, device="cuda:0")
    for _ in range(3):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.7, _lib.ptr(Kd), st)
    e0.record()
    for _ in range(20):
        lib.mgp_kernel_block(_lib.ptr(Z1), n1, _lib.ptr(Z2), n2, m, 1.7, _lib.ptr(Kd), st)
    e1.record(); torch.cuda.synchronize()
    row["us_no_stores"] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    row["max_abs_err"] = worst
    row["rel_to_max"] = worst / max(scale, 1e-30)
    row["modes_max_diff"] = max(float((outs[0] - outs[2]).abs().max()), float((outs[0] - outs[4]).abs().max()))
    res.append(row)
    print(json.dumps(row), flush=True)
lib.mgp_kernel_block_set_pipe(0)
