"""Lab: mgp_kernel_block time against the number of rows / columns / modes around the C3 posterior shape (600 x 60000 x 100):
does the time follow the tile count (tail rounds), the MFMA work or the bytes stored?"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from manifold_gp_amd import _lib
if os.environ.get("MGP_LAB_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["MGP_LAB_LIB"])
lib = _lib.lib()
pipe = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib.mgp_kernel_block_set_pipe(pipe)
shapes = [(600, 60000, mm) for mm in (80, 84, 88, 92, 96, 100, 104, 108, 112, 116, 120, 124, 128)] if len(sys.argv) > 2 and sys.argv[2] == "m" else None
if len(sys.argv) > 2 and sys.argv[2] == "small":
    shapes = [(600, 60000, mm) for mm in (16, 32, 48, 64, 72)] + [(4096, 30000, mm) for mm in (32, 48, 64, 72)]
if len(sys.argv) > 2 and sys.argv[2] == "res":     # where does the resident-operand kernel (knob 5) beat the lean one (knob 1)?
    shapes = [(1000, 50000, 64), (600, 60000, 52), (600, 60000, 76), (600, 60000, 104), (8192, 60000, 100), (16384, 30000, 100), (4096, 4096, 100),
              (128, 60000, 100), (64, 60000, 100), (2048, 2048, 64), (600, 6000, 100), (600, 600, 100), (20000, 20000, 88),
              (600, 60000, 16), (600, 60000, 32), (600, 60000, 48), (600, 60000, 112), (600, 60000, 124), (600, 60000, 128), (4096, 60000, 128), (8192, 8192, 128)]
for (n1, n2, m) in shapes or ((128, 60000, 100), (256, 60000, 100), (384, 60000, 100), (512, 60000, 100), (600, 60000, 100), (640, 60000, 100), (768, 60000, 100),
                    (1024, 60000, 100), (600, 30000, 100), (600, 120000, 100), (600, 60000, 16), (600, 60000, 48), (600, 60000, 96), (600, 60000, 112), (600, 60000, 128)):
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
    tiles = -(-n1 // 128) * -(-n2 // 128)
    print("%5d x %6d x %3d: %7.1f us  %6.1f TFLOP/s  stores %.2f TB/s  tiles %5d = %.2f rounds of 1024, %.2f us per tile-round"
          % (n1, n2, m, us, 2.0 * n1 * n2 * m / us / 1e6, n1 * n2 * 4 / us / 1e6, tiles, tiles / 1024, us / (tiles / 1024)), flush=True)
