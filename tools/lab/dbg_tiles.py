import os, sys, json, ctypes
sys.path.insert(0, os.getcwd())
import torch
import manifold_gp_amd as mgp
from manifold_gp_amd import _lib
from tools import synth
dev = torch.device("cuda:0")
x, y = synth.rmnist_like(600, 100, seed=1337, device=dev)
kern = mgp.kernels.RiemannMaternKernel(nu=2, x=x, nearest_neighbors=50, laplacian_normalization="randomwalk", num_modes=100).to(dev)
g = kern.knn.knn_graph
print({k: v for k, v in (g.tiles or {}).items() if not torch.is_tensor(v)})
lap = kern.laplacian()
Q = mgp.operators.PrecisionMaternOperator(lap, 2, torch.tensor([[1.0]], device=dev))
desc = Q._descriptor()
op = desc.struct()
for C in (1, 4, 8, 12, 16, 100):
    print(C, _lib.lib().mgp_spmm_dot_blocks_csr(ctypes.byref(op.L), C))
print("tile fields", op.L.tile_rows, op.L.tile_max_cols, op.L.tile_max_entries, bool(op.L.lid), bool(op.L.tile_rowptr), bool(op.L.tile_vals), bool(op.L.tile_rowid))
