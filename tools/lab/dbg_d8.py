import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
from manifold_gp_amd.graph import KnnGraph, LaplacianData
dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "dumbbell_k10_loop.npz")))
idx, val, n = torch.from_numpy(g["edge_index"].astype(np.int64)).to(dev), torch.from_numpy(g["edge_value"]).to(dev), g["train_x"].shape[0]
graph = KnnGraph.from_coo(idx, val, n)
data = LaplacianData(graph, 0.1, True)
lib = _lib.lib()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
X = torch.randn(n, C, device=dev)
outs = {}
for mode in (0, 1):
    lib.mgp_spmm_set_dict8_mode(mode)
    csr = data.csr()
    Y = torch.full_like(X, float("nan"))
    _lib.check(lib.mgp_spmm_fused(ctypes.byref(csr), _lib.ptr(X), C, _lib.ptr(Y), 0.0, 1.0, None, None, None, 0.0, 1.0, None, None, _lib.stream()), "spmm")
    torch.cuda.synchronize()
    outs[mode] = Y.cpu().numpy()
d = np.abs(outs[0] - outs[1])
print("tiles", graph.tiles["max_cols"], graph.tiles["max_entries"], "max diff", d.max(), "scale", np.abs(outs[0]).max(), "nan", np.isnan(outs[1]).sum())
bad_rows = np.nonzero(d.max(1) > 1e-3 * np.abs(outs[0]).max())[0]
bad_cols = np.nonzero(d.max(0) > 1e-3 * np.abs(outs[0]).max())[0]
print("bad rows", len(bad_rows), bad_rows[:40]); print("bad cols", len(bad_cols), bad_cols[:70])
rp = graph.rowptr.cpu().numpy()
if len(bad_rows):
    r = bad_rows[0]
    print("row", r, "len", rp[r + 1] - rp[r], "diff per col", d[r][:16], "ref", outs[0][r][:8], "got", outs[1][r][:8])
col = graph.col.cpu().numpy(); vals = data.vals.cpu().numpy(); Xn = X.cpu().numpy()
tiles = graph.tiles; lid = tiles["lid"].cpu().numpy().view(np.uint16); tp = tiles["tile_ptr"].cpu().numpy(); tc = tiles["tile_cols"].cpu().numpy()
for r in bad_rows[:3]:
    e = np.arange(rp[r], rp[r + 1])
    A = (vals[e][:, None] * Xn[col[e]])            # contributions (subtracted in y)
    sd = (outs[1][r] - outs[0][r])                 # got - ref = -(acc_got - acc_ref)
    coef, *_ = np.linalg.lstsq(A.T, -sd, rcond=None)
    print("row", r, "cols", col[e], "vals", np.round(vals[e], 3), "lid", lid[e], "\n  lstsq coef of (acc_got - acc_ref) on entry contributions:", np.round(coef, 3))
    # maybe a wrong X row was used: try all dictionary rows of the tile
    t = r // 64
    dcols = tc[tp[t]:tp[t + 1]]
    B = Xn[dcols]
    coef2, res, *_ = np.linalg.lstsq(B.T, -sd, rcond=None)
    nz = np.nonzero(np.abs(coef2) > 1e-3)[0]
    print("  on dictionary rows: nonzero", [(int(k), int(dcols[k]), round(float(coef2[k]), 3)) for k in nz][:12])
