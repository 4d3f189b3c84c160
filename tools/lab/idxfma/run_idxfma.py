"""Lab: VGPR-indexed accumulation micro-benchmark (gen_idxfma.py -> idxfma.hip -> libidxfma.so, built by build.sh next to this file).
Checks acc[row] += val * x with the row chosen at run time against numpy, and times entries per second."""
import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
lib = ctypes.CDLL(os.path.join(here, "libidxfma%s.so" % mode))
print("mode", mode, "(0 as designed, 1 no scalar loads in the loop, 2 one index change per record)")
dev = torch.device("cuda:0")
ROWS = 32
for blocks, threads, nrec in ((2, 64, 4), (256, 256, 512), (512, 256, 512), (1024, 256, 512)):
    waves = blocks * threads // 64
    rng = np.random.default_rng(0)
    vals = rng.normal(size=(waves, nrec * 8)).astype(np.float32)
    rows = rng.integers(0, ROWS, size=(waves, nrec * 8)).astype(np.uint32)
    stream = np.zeros((waves + 1, nrec * 8, 2), np.uint32)          # (+1 wave of zeros: the last record's prefetch reads one record past a wave's stream)
    stream[:waves, :, 0] = vals.view(np.uint32)
    stream[:waves, :, 1] = 2 * rows
    x = rng.normal(size=(64, 2)).astype(np.float32)
    st, xt = torch.from_numpy(stream.view(np.int32)).to(dev), torch.from_numpy(x).to(dev)
    out = torch.zeros(waves, ROWS, 64, 2, device=dev)
    ms = ctypes.c_float(0)
    rc = lib.idxfma_run(ctypes.c_void_p(st.data_ptr()), nrec, ctypes.c_void_p(xt.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, threads, 20, ctypes.byref(ms))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    w = min(waves, 8)
    ref = np.zeros((w, ROWS, 64, 2))
    for a in range(w):
        np.add.at(ref[a], rows[a], vals[a][:, None, None].astype(np.float64) * x[None, :, :].astype(np.float64))
    err = np.abs(got[:w] - ref).max()
    ent = waves * nrec * 8
    print("blocks %d x %d threads, %d records/wave: rc %d, %.2f us, %.2f G entries/s (%.1f cycles per entry per SIMD at 2.4 GHz, %d waves/SIMD), max err %.2e"
          % (blocks, threads, nrec, rc, ms.value * 1e3, ent / ms.value / 1e6, 2.4e9 * ms.value * 1e-3 / (ent / 1024), max(1, waves // 1024), err), flush=True)
