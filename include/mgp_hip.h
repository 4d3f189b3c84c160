/*
 * mgp_hip.h -- C-ABI of libmgp_hip.so: the MI355X (gfx950) implementation of manifold-gp's
 * sparse graph-Laplacian GP hot path.
 *
 * Boundary contract (every entry point):
 *   - extern "C", plain device pointers + sizes + a hipStream_t passed as void*; no torch types;
 *   - returns int: 0 = ok, <0 = argument error (MGP_ERR_*), >0 = hipError_t from the runtime;
 *   - never allocates or frees caller memory: scratch comes from an explicit `work` buffer whose
 *     size the matching *_workspace_bytes() query returns;
 *   - asynchronous on `stream` unless the comment says it synchronises (graph-build calls that
 *     must report a data-dependent size do);
 *   - all matrices of right-hand sides are row-major [N, C] fp32, indices int32, values fp32.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * nash169/manifold-gp checkout; third-party call sites are the reference's, the arithmetic
 * lived in faiss / torch_sparse / linear_operator / ATen).
 */
#ifndef MGP_HIP_H
#define MGP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGP_OK 0
#define MGP_ERR_ARG (-1)        /* null pointer / bad size / bad enum */
#define MGP_ERR_WORKSPACE (-2)  /* work buffer too small */
#define MGP_ERR_UNSUPPORTED (-3)
#define MGP_ERR_NOT_CONVERGED (-4)
#define MGP_ERR_KNN_AMBIGUOUS (-5)
#define MGP_ERR_TIMEOUT (-6)      /* a multi-rank solve's stream did not drain within MGP_DIST_TIMEOUT_S seconds (default 300):
                                      a peer rank is gone or a collective hangs; the caller must exit, not retry */

#define MGP_PAD 4 /* CSR rows are padded to a multiple of 4 entries (16-byte vector loads) */

/* library / device probes ------------------------------------------------------------------ */
int mgp_version(void);                       /* 100*major + minor */
int mgp_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes); /* hipGetDeviceProperties */

/* ---------------------------------------------------------------------------------------------
 * k-NN: exact brute-force squared-L2 top-k, ascending (d2, index).
 * Replaces faiss IndexFlatL2 / IndexIVFFlat(nlist=1) / Gpu* `search` as called at
 * manifold_gp/utils/nearest_neighbors.py:35-37 (index built at :17-33).
 *   db [N,d] f32 row-major (the indexed points), q [n,d] f32 (queries), k <= min(N, 1024)
 *   D [n,k] f32 = (float) of the fp64 distance, I [n,k] i32
 * Distances/ties follow the rule in oracle/knn_oracle.c (fp64 sum in ascending feature order,
 * no FMA contraction, lower index wins exact ties): fp32 tiles select a padded candidate set,
 * an fp64 re-rank orders it, a per-row bound check proves the set sufficient, rows that fail
 * are redone with a wider set and finally by an exact fp64 scan.  For d <= 3 and N >= 4096 no
 * N x n distance slab is formed: points are Morton-sorted, a window around the query gives a provable
 * upper bound on its k-th neighbour distance, one fused pass over the chunks of points whose bounding box
 * meets the workgroup's search box keeps the points under that bound and the fp64 re-rank orders them (knn_lowd.hip); rows whose candidate list overflows go
 * through the slab pipeline.  Same results bit for bit.  Synchronises `stream`.
 * stats (nullable, host int64[4]): rows redone wide, rows redone exact, chunks, candidates K'
 * (-1 when the low-dimensional path ran; [0] then counts its overflow rows). */
size_t mgp_knn_workspace_bytes(int64_t N, int64_t n, int d, int k);
int mgp_knn_search(const float* db, int64_t N, int d, const float* q, int64_t n, int k,
                   float* D, int32_t* I, void* work, size_t work_bytes, int64_t* stats,
                   void* stream);
/* d >= 32 and >= 1024 queries: the candidate keys of the slab pipeline come from the matrix cores (centred bf16 two-term
 * split, GEMM form, knn_mfma.hip) with an absolute error bound in the sufficiency check; a chunk in which
 * many rows fail that check is redone with the direct-difference tiles.  0 forces the direct tiles
 * (default 1).  mgp_knn_last_direct_chunks: chunks of the last slab search that were redone. */
int mgp_knn_set_mfma(int on);
int64_t mgp_knn_last_direct_chunks(void);
/* Self-search (q == db, n == N: the graph build, nearest_neighbors.py:35-37 called on the training inputs) with the whole
 * N x N key matrix inside the workspace (N <= 92k: 32 GiB): key(x, y) = key(y, x), so the matrix-core kernel computes the
 * tile pairs on and above the diagonal only and stores each off-diagonal tile twice (as it is and transposed): half the
 * MFMA work, the same selection, the same results bit for bit.  Default 1; 0 computes every tile (tests, A/B runs). */
int mgp_knn_set_symmetric(int on);
/* Candidate filter of the matrix-core searches (round 5).  A large search does not write its keys to an N x n slab for the
 * select kernel to read back (60k x 60k: 14.4 GB each way): the keys of every row to a 1/stride sample of the points
 * (stride 16 up to k = 64) give a per-row bound >= the row's K'-th smallest key; the key pass appends the ~stride K' keys
 * per row under the bounds to a log (one returning atomic per WORKGROUP and tile pair draws its range; a self-search logs
 * each key of an off-diagonal tile for both of its rows); regroup_kernel deals the log to per-row candidate lists (3840
 * entries of {key, index}) with LDS counters; the select kernel takes the exact K'-th smallest key, the candidates, the fp64
 * re-rank and the sufficiency check from the list.  Rows whose list overflows or whose check fails are gathered and redone
 * by the slab pipeline (which ends in the exact scan); a chunk whose log fills up is redone there whole: the results are
 * the oracle's bit for bit, as before.  Workspace at 60k x 784: 4.9 GB instead of 14.8.  mode 0: key slab (rounds 1-4);
 * 1 (default): searches of >= 4096 queries against >= 16384 points; 2: every matrix-core search whose k the lists can serve
 * (tests).  Lab / test switch: read once per call.
 * mgp_knn_last_filter_failover: rows of the last search handed to the slab pipeline; -1 when the search ran on the slab. */
int mgp_knn_set_filter(int mode);
int64_t mgp_knn_last_filter_failover(void);
/* Prepared index (d >= 32): the part of a search that depends on the points alone -- column means, the centred bf16 split
 * in the key kernel's tile layout, norms -- built once (faiss: index.train / index.add, nearest_neighbors.py:20-33) and
 * handed to every search.  Without it a search prepares the points itself (~1 ms at 60k x 784) and therefore keeps small
 * query batches (< 1024 rows) on the fp32 direct-difference tiles; with it the matrix cores rank the candidates of any batch
 * (600 queries against 60k x 784: 1.3 -> 0.5 ms).  The index is a SNAPSHOT of db at build time (faiss copies its vectors):
 * rebuild it when db changes.  Results are the oracle's bit for bit either way.  mgp_knn_index_bytes: 0 when d < 32. */
size_t mgp_knn_index_bytes(int64_t N, int d);
int mgp_knn_index_build(const float* db, int64_t N, int d, void* index, size_t index_bytes, void* stream);
int mgp_knn_search_indexed(const float* db, int64_t N, int d, const void* index, size_t index_bytes, const float* q,
                           int64_t n, int k, float* D, int32_t* I, void* work, size_t work_bytes, int64_t* stats,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * Graph: k-NN lists -> symmetrised graph.  Replaces NearestNeighbors.graph
 * (manifold_gp/utils/nearest_neighbors.py:39-55: drop column 0, orient row<col,
 * torch_sparse.coalesce(op='mean')).
 * Outputs (caller allocates the upper bounds, actual sizes returned through host pointers):
 *   reference view: tri_row/tri_col i32 [<= n(k-1)], tri_val f32 -- sorted unique (row<=col) COO,
 *                   value = fp32 mean of the duplicates;  *M = number of undirected edges
 *   kernel view   : full symmetric padded CSR of the same graph: rowptr i32 [n+1] (every row
 *                   start a multiple of MGP_PAD), col i32 / d2 f32 / eid i32 [<= 2n(k-1)+4n]
 *                   (eid = index into tri_*; padding entries: col = own row, d2 = +inf, eid = -1)
 *                   *nnz = rowptr[n].  A (i,i) edge (duplicate points) appears twice in row i,
 *                   exactly as the reference scatters it twice.
 * Synchronises `stream` (needs M on the host). */
size_t mgp_graph_workspace_bytes(int64_t n, int k);
/* The edge list of NearestNeighbors.graph for ANY flag combination (nearest_neighbors.py:39-55), without the CSR that
 * mgp_graph_build adds for the default one (symmetric, no self loops): self_loop keeps column 0 of the lists (an (i, i) entry
 * at distance 0), symmetric orients every pair as (min, max), sorts and merges duplicates with the fp32 mean
 * (torch_sparse.coalesce(op='mean')); not symmetric: the n (k - first) directed pairs in row-major order.  out_* hold
 * n (k - first) entries (first = 0 with self_loop, else 1); *M = the number written.  Workspace (symmetric only):
 * mgp_graph_workspace_bytes(n, k + 1). */
int mgp_graph_edges(const float* D, const int32_t* I, int64_t n, int k, int symmetric, int self_loop, int32_t* out_row,
                    int32_t* out_col, float* out_val, int64_t* M, void* work, size_t work_bytes, void* stream);
int mgp_graph_build(const float* D, const int32_t* I, int64_t n, int k, int32_t* tri_row, int32_t* tri_col,
                    float* tri_val, int64_t* M, int32_t* rowptr, int32_t* col, float* d2, int32_t* eid,
                    int64_t* nnz, void* work, size_t work_bytes, void* stream);
/* same CSR from an existing reference-style edge list (idx[2,M] given as two i32 arrays) */
size_t mgp_graph_coo_workspace_bytes(int64_t n, int64_t M);
int mgp_graph_from_coo(const int32_t* tri_row, const int32_t* tri_col, const float* tri_val,
                       int64_t M, int64_t n, int32_t* rowptr, int32_t* col, float* d2, int32_t* eid,
                       int64_t* nnz, void* work, size_t work_bytes, void* stream);

/* Row-tile column dictionaries for the C == 1 SpMV.  A tile is `tile_rows` consecutive rows of the padded
 * CSR; tile_cols lists the distinct columns the tile references (ascending), lid maps every entry to
 * its column's position in that list.  The SpMV stages x[tile_cols] in LDS once per tile, so the
 * texture path sees one access per DISTINCT column instead of one per entry and the matrix stream
 * shrinks to 6 bytes per entry.  No reference counterpart (the reference scatters with atomics,
 * graph_laplacian_operator.py:117-120).
 *   tile_ptr [ceil(n/tile_rows)+1], tile_cols [capacity nnz], lid [nnz]  (device, caller allocated)
 *   total_cols / max_cols / max_entries: host outputs.  MGP_ERR_UNSUPPORTED when a tile references
 *   more than 65536 distinct columns (use a smaller tile or leave the dictionaries off). */
size_t mgp_graph_tiles_workspace_bytes(int64_t n, int64_t nnz);
/* row_order (nullable): a permutation of the rows; tile t then holds rows row_order[64 t .. 64 t + 63]
 * (locality order for inputs that arrive unordered).  With it the tile view is a second CSR in that
 * order: tile_rowptr [n+1] (entry offsets), emap [nnz] (tile-order entry -> entry of the CSR: gather
 * the values through it, mgp_csr_t.tile_vals), lid in tile order; column ids stay the original ones. */
int mgp_graph_tiles(int64_t n, const int32_t* rowptr, const int32_t* col, int64_t nnz, int tile_rows,
                    const int32_t* row_order, int32_t* tile_rowptr, int32_t* emap, int32_t* tile_ptr,
                    int32_t* tile_cols, uint16_t* lid, int64_t* total_cols, int32_t* max_cols,
                    int32_t* max_entries, void* work, size_t work_bytes, void* stream);

/* Breadth-first (Cuthill-McKee style) locality order of the graph: order [n] = node ids level by level,
 * a level sorted by (position of the first-numbered parent, node id); components are appended one
 * after the other.  Deterministic.  For inputs whose node order carries no locality (random point
 * order) the tile dictionaries built on this order recover the reuse a spatially sorted input has.
 * Synchronises `stream` (one host round trip per BFS level). */
/* Morton (Z-curve) order of points with d <= 3: int32 [n] permutation, ascending code.  The cheaper and
 * tighter alternative to the BFS order when coordinates of low dimension are at hand.  Synchronises. */
size_t mgp_morton_order_workspace_bytes(int64_t n);
int mgp_morton_order(const float* x, int64_t n, int d, int32_t* order, void* work, size_t work_bytes, void* stream);
/* Nearest-neighbour chain order (round 5): from the node in hand step to its nearest not-yet-numbered neighbour (by d2), else to
 * that of one of the last 64 chain nodes, else to the unnumbered node of smallest index.  For k-NN graphs whose given order keeps
 * clusters together but not the order inside them (rotation orbits in random angle order) it makes the rows of a tile share most
 * of their columns: 2.5 x fewer distinct columns per 16-row tile on the 60k RMNIST-like graph, which is what the matrix-core SpMM's
 * work is proportional to.  Sequential walk on the HOST over a copy of the CSR (one-off per graph); deterministic; synchronises. */
int mgp_graph_chain_order(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2, int32_t* order, void* stream);
/* dst[p, :] = src[order[p], :] for a row-major float32 block [n, C] (order: int32 [n], any row list with entries in [0, n_src);
 * src != dst).  What the solvers that iterate on P A P^T (graph.RelabelledGraph) permute right-hand sides in and solutions out
 * with; no counterpart in the reference. */
int mgp_permute_rows(const float* src, const int32_t* order, int64_t n, int C, float* dst, void* stream);
size_t mgp_graph_bfs_workspace_bytes(int64_t n);
int mgp_graph_bfs_order(int64_t n, const int32_t* rowptr, const int32_t* col, int32_t* order, void* work,
                        size_t work_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Laplacian build: diffusion-maps normalisation of the kernel weights, fused row passes.
 * Replaces the cached properties adjacency_unnorm_mat / degree_unnorm_mat / adjacency_mat /
 * degree_mat / laplacian_diag / laplacian_triu of
 * manifold_gp/operators/graph_laplacian_operator.py:52-106.
 *   in : padded CSR (rowptr, col, d2), eps (graph bandwidth), self_loops
 *   out: degree_unnorm D~ [n], degree D [n], diag [n], dsqrt = sqrt(D) [n], dinvsqrt [n],
 *        vals [nnz] = S_ij = A_ij / (sqrt(D_i) sqrt(D_j)) / eps^2  (positive; L_ij = -S_ij) */
int mgp_laplacian_build(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2,
                        float eps, int self_loops, float* degree_unnorm, float* degree,
                        float* diag, float* dsqrt, float* dinvsqrt, float* vals, void* stream);
/* Forward-mode tangent of mgp_laplacian_build wrt the graph bandwidth: d/d eps of D~, D, diag, sqrt(D),
 * 1/sqrt(D) and the CSR values (same three row passes).  With it d(u^T L v)/d eps = u^T L' v is one more
 * fused SpMV on (d_vals, d_diag): the hyper-parameter gradient path the reference obtains by autograd
 * through graph_laplacian_operator.py:52-124 (pinned by test/_test_functions.py:59-74). */
int mgp_laplacian_tangent(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2,
                          float eps, int self_loops, const float* degree_unnorm, const float* degree,
                          const float* diag, float* d_degree_unnorm, float* d_degree, float* d_diag,
                          float* d_dsqrt, float* d_dinvsqrt, float* d_vals, void* stream);
/* The reductions of the differentiable fused SpMM's backward pass in one launch (autograd through graph_laplacian_operator.py:108-124 /
 * precision_matern_operator.py:26-37 wrt bandwidth, length scale and the node vectors; `_test_functions.py:59-104`).  Blocks are
 * [n, C] row-major float32: h = cov post (.) g (upstream gradient), xs = pre (.) X, dlx = L' xs (tangent product), lx = L xs,
 * gxs = a h + b L h.  Writes partial[mgp_spmm_backward_blocks(n)][2] = per-workgroup sums of <h, dlx> and <h, xs> (the caller adds
 * them up), gpre[r] = sum_c gxs X and gpost[r] = cov sum_c g (a xs + b lx).  Null dlx / gpre / gpost switch that output off. */
int mgp_spmm_backward_blocks(int64_t n);
int mgp_spmm_backward_sums(int64_t n, int C, const float* h, const float* dlx, const float* xs, const float* gxs, const float* X,
                           const float* g, const float* lx, float av, float bv, float cov, float* partial, float* gpre,
                           float* gpost, void* stream);
/* per-edge values in the reference's COO order: which = 0 W (adjacency_unnorm_mat :54-56),
 * 1 A (adjacency_mat :73-75), 2 S (laplacian_triu :104-106) */
int mgp_edge_values(const int32_t* tri_row, const int32_t* tri_col, const float* tri_val,
                    int64_t M, const float* degree_unnorm, const float* degree, float eps,
                    int which, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The hot kernel: fused CSR SpMV / SpMM with the symmetric Laplacian.
 *   xs_j = pre ? pre[j] * X[j,:] : X[j,:]
 *   lx_i = diag[i] * xs_i - sum_j vals[ij] * xs_j
 *   t_i  = (post ? post[i] : 1) * (a * xs_i + b * lx_i)
 *   Y[i,:] = (base ? cb * base[i,:] : 0) + co * t_i
 *   dot_partials (nullable) [dot_blocks, C]: per-workgroup partial sums of dotw[i,:] * Y[i,:]
 * Replaces GraphLaplacianOperator._matmul (graph_laplacian_operator.py:108-124: two
 * torch_sparse.spmm calls + diagonal + random-walk scalings) and is the building block of the
 * precision / wrapper operators below.  pre/post/base/dotw/dot_partials may be NULL. */
typedef struct {
  int64_t n;
  const int32_t* rowptr;
  const int32_t* col;
  const float* vals;
  const float* diag;
  int64_t ncols;           /* length of the vectors the columns index (0 = n; > n for a row slice) */
  /* row-tile column dictionaries (mgp_graph_tiles); all NULL / 0 = gather straight from memory */
  const int32_t* tile_ptr;   /* [ntiles + 1] offsets into tile_cols, ntiles = ceil(n / tile_rows) */
  const int32_t* tile_cols;  /* distinct column ids of each tile, ascending */
  const uint16_t* lid;       /* [nnz] position of every entry's column in its tile's list */
  int32_t tile_rows;
  int32_t tile_max_cols;     /* max over tiles of the list length (LDS floats per workgroup) */
  int32_t tile_max_entries;  /* max over tiles of the padded entry count */
  int32_t tile_reserved;
  /* tiles over a row ORDER (mgp_graph_tiles with row_order): all NULL = tile t is rows 64 t .. 64 t + 63 */
  const int32_t* tile_rowptr;  /* [n+1] entry offsets in tile order */
  const float* tile_vals;      /* [nnz] vals gathered through emap */
  const int32_t* tile_rowid;   /* [n] original row of every tile-order position (= row_order) */
  /* dense 16-row tiles for the matrix-core SpMM at 48 <= C <= 256 (mgp_spmm_mt_fill); all NULL / 0 = not built */
  const int32_t* mt_sptr;      /* [mt_tiles + 1] steps (of four distinct columns) before tile t; multiples of 16 */
  const int32_t* mt_dcol;      /* [4 mt_steps + 192] the tiles' distinct columns, padded per tile to whole bodies of 64 */
  const float* mt_img;         /* [64 (mt_steps + 32)] tile values in MFMA operand order */
  int32_t mt_tiles;            /* ceil(n / 16) */
  int32_t mt_steps;            /* mt_sptr[mt_tiles] */
} mgp_csr_t;

/* workgroups that write dot partials for this CSR (format aware; use this one to size dot_partials) */
int mgp_spmm_dot_blocks_csr(const mgp_csr_t* L, int C);
int mgp_spmm_set_tile_mode(int on);          /* C == 1: use the tile dictionaries when present (default 1) */
int mgp_spmm_set_tile_small_mode(int on);    /* C in {4,8,12,16}: LDS-dictionary multi-column kernel on 64-row tiles
                                                (default 1; 0 = the per-entry gather kernel) */
/* 16 < C <= 256 with C % 4 == 0 on 64-row tiles: LDS-dictionary kernel in 16-column chunks where it wins (C <= 32, or
 * an X block of 96 MB and more: default 1); 0 = always the per-entry X-row gather kernel; 2 = always the dictionary
 * kernel (tests, A/B runs). */
int mgp_spmm_set_tile_wide_mode(int on);
/* 16 < C <= 256 with C % 4 == 0 on 64-row tiles (round 3): the dictionary kernel with LANES OVER COLUMNS
 * (csrc/spmm.hip spmm_dict_kernel) -- a tile's distinct X rows cross the vector memory path once per tile, staged in
 * double-buffered slices of 256-byte-aligned LDS slots that every entry then reads conflict-free.  Default 1: taken
 * wherever the shape allows (it takes precedence over the two kernels above); 0 = never.  Requires what the graph
 * builders guarantee: within a row the entries' columns ascend (padding entries, value 0, at the row's end).
 * Replaces torch_sparse.spmm at manifold_gp/operators/graph_laplacian_operator.py:118-119 for the [N, 100] right-hand
 * sides of precision_matern_operator.py:50-53 and the eigensolver's blocks. */
int mgp_spmm_set_dict_mode(int on);
/* 48 <= C <= 256 with C % 4 == 0 (round 4): the SpMM on the fp32 MATRIX CORES over 16-row tiles stored dense in their own distinct
 * columns (csrc/spmm.hip spmm_mt_kernel): a tile's distinct X rows cross the vector memory path once per tile and 64-column
 * block, straight into the MFMA operand layout; no LDS, no barrier.  Taken (before every other wide kernel) when the CSR
 * carries mt_* and the call has no row offset (weighted dot-product partials included: one row of partials per workgroup of four
 * (tile, block) waves, mgp_spmm_dot_blocks_csr counts them); a row's sum is taken in ascending column order.
 * N = 60k, C = 128: 60 us against 94 for the gather kernel.  mgp_spmm_set_mt_mode(0) = never; returns the previous setting.
 * mgp_spmm_mt_fill builds mt_dcol / mt_img from a CSR in natural row order and its 16-row tile dictionaries
 * (mgp_graph_tiles with tile_rows = 16: tile_ptr16, tile_cols16, lid16) and the step offsets sptr (per tile
 * 16 ceil(D / 64) steps -- whole bodies of four blocks --, exclusive prefix sum, `steps` = the total).
 * Replaces torch_sparse.spmm at manifold_gp/operators/graph_laplacian_operator.py:118-119 for the eigensolver's blocks and
 * the [N, 100] right-hand sides of precision_matern_operator.py:50-53. */
int mgp_spmm_set_mt_mode(int on);
/* which kernel mgp_spmm_fused would launch for this CSR / width (tests, docs): 0 = gather, 1 = C == 1 tile kernel, 2 = small-C
 * tile kernel, 3 = matrix-core tiles, 5 = lanes-over-columns dictionary, 6 = chunked dictionary (4 was a round-4 kernel, removed) */
/* A CSR that carries the matrix-core image, called with a row offset: the other kernels run (the image is ignored) -- except
 * with dot partials, whose count mgp_spmm_dot_blocks_csr sized for the matrix-core kernel: MGP_ERR_UNSUPPORTED from this
 * function and from mgp_spmm_fused_rows alike (strip mt_* from the struct for such a call). */
int mgp_spmm_kernel_choice(const mgp_csr_t* L, int C, int with_dot, int64_t row_offset);
int mgp_spmm_mt_fill(int64_t n, const int32_t* rowptr, const float* vals, const uint16_t* lid16, const int32_t* tile_ptr16,
                     const int32_t* tile_cols16, const int32_t* sptr, int64_t steps, int32_t* dcol, float* img, void* stream);
/* Measurement hook (bench.py `roofline`): between begin and end every EAGER launch of the C == 1 tile kernel carries
 * its own start / stop event pair (hipExtLaunchKernelGGL: the dispatch's begin / end timestamps); end returns the sum of
 * the kernel durations and the number of launches timed (at most max_launches).  No effect on results. */
int mgp_spmm_timing_begin(int max_launches);
int mgp_spmm_timing_end(float* total_ms, int* launches);
/* 16 < C <= 256 with C % 4 == 0 on a quad-padded CSR (what the graph builder produces), 16-byte aligned operands: a
 * lane owns one float4 of the row instead of one column; used where it wins (C <= 64, or an X block of 96 MB and more:
 * default 1); 0 = never (the per-column gather kernel); 2 = always (tests, A/B runs). */
int mgp_spmm_set_v4_mode(int on);

int mgp_spmm_dot_blocks(int64_t n, int C);   /* workgroups that write dot partials */
int mgp_spmm_set_group_hint(int lanes);      /* C == 1: lanes per row, one of 4,8,16,32,64 */
int mgp_spmm_set_rows_in_flight(int rows);   /* C == 1 fallback kernel: rows a lane group loads at once: 1,2,4,8 */
int mgp_spmm_fused(const mgp_csr_t* L, const float* X, int C, float* Y, float a, float b,
                   const float* pre, const float* post, const float* base, float cb, float co,
                   const float* dotw, float* dot_partials, void* stream);

/* row-partitioned form: L_local = rows [row_offset, row_offset + L_local->n) of the operator (column
 * ids global), vectors of global length; writes only the local rows of Y (and local dot partials) */
int mgp_spmm_fused_rows(const mgp_csr_t* L_local, int64_t row_offset, const float* X, int C, float* Y,
                        float a, float b, const float* pre, const float* post, const float* base,
                        float cb, float co, const float* dotw, float* dot_partials, void* stream);

/* measurement helper: `reps` back-to-back launches of Y = L X replayed as one hipGraph; elapsed_ms
 * (nullable, host) = HIP-event time of the launches on `stream`.  Synchronises `stream`. */
int mgp_spmm_repeat(const mgp_csr_t* L, const float* X, int C, float* Y, int reps, float* elapsed_ms,
                    void* stream);

/* L @ X in the three flavours of graph_laplacian_operator.py:108-124:
 * mode 0 symmetric, 1 randomwalk (D^-1/2 L_sym D^1/2), 2 randomwalk transposed */
int mgp_laplacian_matmul(const mgp_csr_t* L, const float* dsqrt, const float* dinvsqrt, int mode,
                         const float* X, int C, float* Y, float* work_nc, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Precision-side operator family, applied matrix-free as a chain of fused SpMMs:
 *   Q2  = scale * diag(post) (tau I + L_sym)^nu diag(pre)          tau = 2 nu / kappa^2
 *         - symmetric normalisation: pre = post = NULL
 *           (PrecisionMaternOperator._matmul, precision_matern_operator.py:26-37)
 *         - random walk: pre = post = sqrt(D), because
 *           D (tau I + L_rw)^nu = D^1/2 (tau I + L_sym)^nu D^1/2   (same file, :36-37)
 *         - scale = outputscale or 1/outputscale (ScaleWrapperOperator,
 *           scale_wrapper_operator.py:27)
 *         - a 0/1 mask multiplied into pre / post gives the MaskedLinearOperator blocks
 *           Q_ll, Q_lu, Q_ul, Q_uu of schur_complement_operator.py:26-30 on full-length
 *           zero-padded vectors
 *   A   = form 0: Q2
 *         form 1: Q2 - s Q2^2 + s^2 Q2^3   (NoiseWrapperOperator, noise_wrapper_operator.py:22,
 *                                           evaluated as Q2(v - s Q2(v - s Q2 v)))
 *         form 2: I + s Q2                 ((K + s I) in precision form with K = Q2^-1:
 *                                           K (K + s I)^-1 y = (I + s Q2)^-1 y) */
typedef struct {
  mgp_csr_t L;
  const float* pre;   /* nullable [n] */
  const float* post;  /* nullable [n] */
  int32_t nu;
  float kappa;        /* lengthscale */
  float scale;        /* 1 = none */
  int32_t form;       /* 0, 1, 2 above */
  float noise;        /* s */
} mgp_operator_t;

size_t mgp_operator_workspace_bytes(const mgp_operator_t* op, int C);
int mgp_operator_apply(const mgp_operator_t* op, const float* X, int C, float* Y, void* work,
                       size_t work_bytes, void* stream);
/* as above, additionally writing per-workgroup partials of dotw . Y (used by CG/Lanczos) */
int mgp_operator_apply_dot(const mgp_operator_t* op, const float* X, int C, float* Y,
                           const float* dotw, float* dot_partials, void* work, size_t work_bytes,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * (Preconditioned) conjugate gradients on A x = b, multi right-hand side, single-reduction
 * (Chronopoulos-Gear) recurrence, whole iterations captured in a hipGraph.
 * Replaces linear_operator.utils.linear_cg as reached from
 * precision_matern_operator.py:53 (`inv_quad_logdet`), schur_complement_operator.py:28
 * (`.solve`) and train_model.py:68.
 *   B [n,C] rhs, X [n,C] solution (in: ignored, start from 0), minv (nullable) Jacobi 1/diag(A)
 *   stop: mode 0 = linear_cg's rule (rhs columns normalised; >= min_iter(10) iterations; stop
 *         when the mean over columns of ||r||_2 < tol; columns with ||r|| < 1e-10 freeze),
 *         mode 1 = every column ||r||_2 <= tol * ||b||_2
 *   out (host): iters, resid[C] relative residual norms (recurrence residual).
 * A solve that needs k steps is ONE hipGraph launch (cg_init + k x (operator apply, fused update);
 * the rhs pointer is patched into the graph, its length follows the previous solves).  The call
 * returns when the stopping decision has reached the host (host-mapped flag); X is complete in
 * stream order: consume it on `stream` or synchronise before reading it from elsewhere. */
typedef struct {
  float tol;
  int32_t max_iter;
  int32_t min_iter;
  int32_t stop_mode;
  int32_t check_every; /* iterations per graph launch / host convergence poll (0 = default) */
  int32_t use_graph;   /* 1 = hipGraph replay (default), 0 = eager launches */
  int32_t max_refine;  /* stop_mode 1 only: iterative-refinement rounds on the TRUE residual B - A x
                          (0 = off); `resid` then reports true relative residuals.  A refined solve stops (status 1)
                          once every true relative residual is <= 2 tol: the fp32 evaluation of B - A x scatters by about
                          that factor around the tolerance */
} mgp_cg_params_t;

size_t mgp_cg_workspace_bytes(const mgp_operator_t* op, int C);
/* ---- Lab-only switches (every mgp_*_set_* in this header): process-wide std::atomic<int> words for A/B measurements and tests.
 * The SpMM family copies them once per call into a thread-local snapshot that all of the call's shape tests consult; the CG
 * switches are read at plan creation; the kernel-block switch once per call; the k-NN / eigensolver switches where a call
 * branches on them.  They are not part of the path's contract and are not meant to be flipped while another thread is inside a
 * call of the same family: two calls that size and launch the same product (mgp_spmm_dot_blocks_csr, then mgp_spmm_fused; the
 * k-NN workspace query, then the search) must see the same setting. */
/* C == 1 plans: the LAST update launch of a plan's first graph also takes the stopping decision of the step behind it and
 * leaves the end-of-graph mark, instead of a single-workgroup decision launch + a marker launch behind it.  Hand-off inside the
 * launch (no release / acquire fence -- a fence writes back every dirty L2 line of the vectors the launch has just stored, measured
 * +3.4 us per solve): lane 0 of every workgroup stores its ||r||^2 partial with a write-through (sc1) relaxed agent-scope atomic
 * store, drains it (s_waitcnt vmcnt(0)) and makes one returning relaxed agent-scope atomic add on its group's arrival counter
 * (two levels: eight groups, one top word); the workgroup whose add completes the count reads all partials with sc1 relaxed
 * agent-scope atomic loads, sums them in the decision kernel's order and takes the decision (MI355X_MICROARCH.md, inter-workgroup
 * visibility: sc1 stores reach the shared level before vmcnt retires them, sc1 loads miss the non-coherent levels).  Same sums in
 * the same order, same rule, same flags: bit-identical to the separate launches (tests/test_gpu_parity.py::
 * test_cg_decide_in_update_matches_separate_launches).  Default 1; 0 = the separate launches; returns the previous setting; read
 * at plan creation. */
int mgp_cg_set_decide_in_update(int on);
/* Complex-shift solve (round 5).  C == 1 plans on A = I + s Q2 (form 2) with nu = 2 and no pre / post vectors (symmetric
 * normalisation), stop_mode 1, no preconditioner: A = I + c B^2 = (I + i sigma B)(I - i sigma B) with B = tau I + L_sym,
 * c = noise * scale, sigma = sqrt(c), and x = Re[(I + i sigma B)^-1 b].  The plan then runs COCG (the CG recurrences with the
 * unconjugated bilinear form) on the complex symmetric system, whose condition is ~sqrt(cond(A)): ONE product with B (the
 * 4-column tile SpMM over (re, im, re, im)) per iteration and about the square root of CG's iteration count -- 1M-node swiss
 * roll: 58 iterations against 688 of two SpMVs each.  `iters` counts COCG iterations, `resid` is the relative norm of the
 * COMPLEX residual (refined solves report the true residual of A x = b as before).  The reference's call is the unpreconditioned
 * linear_cg of precision_matern_operator.py:53 / riemann_gp.py:45-75 on the same system.
 * mgp_cg_set_complex_shift(0): CG on A as in rounds 1-4 (A/B runs, tests, the in-solve SpMV profile); returns the previous
 * setting; read at plan creation.  mgp_cg_plan_is_complex_shift: 1 when the plan took that form. */
int mgp_cg_set_complex_shift(int on);
int mgp_cg_plan_is_complex_shift(void* plan);
/* Plans with more than 16 columns sum the dot-product partials of a step ONCE (cg_reduce_kernel, one small launch
 * ahead of the update) instead of in every workgroup of the update kernel.  Default 1; 0 = the every-workgroup
 * scheme at any C (A/B measurements, tests); affects plans created afterwards. */
int mgp_cg_set_reduce_once(int on);
/* Plans whose column count is a multiple of four (the 12 probes, the 32 / 100 one-hot columns of training) run their vector update
 * on float4 streams (cg_update_q_kernel: a lane owns a column QUAD of a row, every access a dwordx4; cg_update_kernel gives a lane
 * one element and idles tile_cols(C) - C lanes of every row).  Default 1; 0 = the element form at every column count (A/B runs,
 * tests); read at plan creation. */
int mgp_cg_set_update_quads(int on);
/* The host learns of the end of a solve from a host-mapped word the deciding kernel writes.  While the first graph of
 * a plan runs -- it is sized to end in the stopping decision -- the host reads only that word, for up to twice the
 * time the previous solve took (`spins` reads between two looks at the clock, default 64); hipStreamQuery, the guard
 * against a chunk that ends undecided, comes after that window and for continuation chunks.  0: no window, one
 * query per read as in round 1. */
int mgp_cg_set_poll_spin(int spins);
/* C == 1 plans on the tile SpMV without a preconditioner start WITHOUT a cg_init launch: the first operator apply
 * reads the right-hand side itself, copies it to r and leaves ||b||^2 as partials; the first update treats p, s, x
 * as zero.  One launch (~3.8 us at N = 60k) less per solve.  Default 1; 0 restores the classic start (plans created
 * afterwards). */
int mgp_cg_set_init_free(int on);
/* reusable solver: owns the captured iteration graph; `work` must outlive the plan */
int mgp_cg_plan_create(const mgp_operator_t* op, int C, const float* minv,
                       const mgp_cg_params_t* params, void* work, size_t work_bytes, void* stream,
                       void** plan_out);
int mgp_cg_plan_solve(void* plan, const float* B, float* X, int32_t* iters, float* resid,
                      int32_t* status); /* status 1 converged, 2 max_iter, 3 breakdown (NaN) */
/* Point an existing single-GPU plan at another operator of the SAME structure -- same sizes, tile / dense-tile layouts, nu, form,
 * the same of pre / post / minv present --, e.g. the same graph at the next epoch's bandwidth and length scale (the reference
 * rebuilds its operators every epoch, train_model.py:63-67; a fresh plan costs ~0.9 ms of host time between its creation, its
 * first eager solve, the capture at its second and its destruction).  Pointers and scalars may all differ.  The workspace, the
 * host-mapped flags and the executable graphs are kept: the next solve records its launches again and updates the graphs in
 * place (hipGraphExecUpdate).  MGP_ERR_UNSUPPORTED when the structure differs (create a new plan), for distributed plans, and for a
 * complex-shift plan whose new operator no longer factorises.  The old operator's arrays are not touched after this call. */
int mgp_cg_plan_rebind(void* plan, const mgp_operator_t* op, const float* minv);
float* mgp_cg_plan_x(void* plan);    /* device pointer of the plan's own solution buffer [n,C]; pass
                                         X = NULL to mgp_cg_plan_solve to skip the copy into X */
/* refined solves (max_refine > 0, single GPU) accumulate the solution in float64 and form the true residual
 * from it; the float32 buffer above holds its rounding.  Device pointer of that float64 solution [n,C]
 * (valid after a refined solve, until the next solve); NULL for a distributed plan. */
double* mgp_cg_plan_x64(void* plan);
/* operator applies the last solve actually ran (refinement off).  A plan's first graph is captured for the
 * step count its previous solves needed; for C = 1 its last step -- the one that would only notice
 * ||r|| <= tol after running one more apply for nothing -- is a single-workgroup decision launch, so a
 * solve that takes k iterations runs k applies, not k + 1.  Same iterates, residuals and flags. */
int mgp_cg_plan_last_applies(void* plan);
int mgp_cg_plan_destroy(void* plan);
/* 1 after a solve of this plan returned MGP_ERR_TIMEOUT (a dead peer / hung collective of a multi-rank job): kernels and
 * collectives that reference the plan's buffers are still queued, so the plan is POISONED -- further solves return
 * MGP_ERR_TIMEOUT at once and mgp_cg_plan_destroy frees and synchronises nothing (the memory is leaked on purpose; the
 * caller must keep the workspace alive and exit non-zero). */
int mgp_cg_plan_poisoned(void* plan);
int mgp_cg_plan_poison(void* plan);     /* mark it so by hand: the caller has learnt by other means that a peer rank is gone */
int mgp_cg_solve(const mgp_operator_t* op, const float* B, int C, float* X, const float* minv,
                 const mgp_cg_params_t* params, int32_t* iters, float* resid, void* work,
                 size_t work_bytes, void* stream);
/* cheap Jacobi preconditioner: minv_i = 1 / A_ii with the polynomial's off-diagonal
 * contributions to the diagonal ignored for nu > 2 (exact for nu <= 2) */
int mgp_operator_jacobi(const mgp_operator_t* op, float* minv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Eigensolve: the m smallest eigenpairs of L_sym by a Chebyshev-filtered block Krylov iteration
 * with Rayleigh-Ritz on L_sym itself (csrc/eigen.hip explains why not single-vector Lanczos; the
 * plain Lanczos tridiagonalisation is mgp_lanczos_tridiag below).
 * Replaces torch.linalg.eigh on the densified N x N matrix at
 * manifold_gp/kernels/riemann_kernel.py:121-125 and
 * GraphLaplacianOperator.diagonalization (graph_laplacian_operator.py:132-144).
 *   evals [m] ascending (host), evecs [n,m] row-major (device, orthonormal), resid [m] (host)
 *   = ||L v - lambda v||_2.  Synchronises `stream`. */
typedef struct {
  int32_t max_basis;   /* block size b (0 = default: next multiple of 64 above m + max(m/8, 12)) */
  int32_t degree;      /* Chebyshev filter degree (0 = adaptive 8..200) */
  int32_t max_restarts;/* outer filter + Rayleigh-Ritz rounds (0 = 40) */
  float tol;           /* residual tolerance relative to lambda_max */
  uint64_t seed;
} mgp_lanczos_params_t;

/* host-only helper of the block eigensolver: eigendecomposition of a small dense symmetric matrix in
 * fp64 (Householder tridiagonalisation + implicit QL).  A [n x n] row-major, evals ascending,
 * eigenvectors = columns of V [n x n] row-major.  No device work. */
int mgp_host_symeig(int n, const double* A, double* evals, double* V);
size_t mgp_lanczos_workspace_bytes(int64_t n, int m, const mgp_lanczos_params_t* p);
int mgp_lanczos_smallest(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals,
                         float* evecs, float* resid, int32_t* info, void* work, size_t work_bytes,
                         void* stream);
/* The same solve, also handing out the WHOLE Rayleigh-Ritz block it ended with -- the m wanted pairs plus the guard
 * columns behind them (b = mgp_lanczos_block_size(m, p) columns, Ritz values ascending): block_evals [b] (host),
 * block_evecs [n, b] row-major (device), block_resid [b] (host); any of the three may be NULL.  What an independent
 * check of the spectral stage needs (tests/test_gpu_configs.py: float64 Rayleigh-Ritz of the block with the oracle's
 * matrix, the gap behind the kept modes, a Davis-Kahan bound).
 * Return value of both: MGP_OK when all m residuals are <= tol * lambda_max, and ALSO when the iteration has reached the
 * fp32 residual floor (a few ulp of |L|: a round with the filter at its degree cap that does not even halve the largest
 * residual and converges no further pair), or when the measured gap between the wanted block and its last guard column says
 * that a further round at the degree cap would not even halve the residual: info[2] (pairs under tol) < m tells these
 * from convergence and `resid` holds what was reached.  Both early exits require the largest residual to be within
 * max(50 tol, 2e-5) * lambda_max (2e-5 = 10 x the measured fp32 floor); the Python wrapper warns (EigenFloorWarning) whenever
 * info[2] < m.  MGP_ERR_NOT_CONVERGED: max_restarts rounds without either. */
int mgp_lanczos_block_size(int m, const mgp_lanczos_params_t* p);
int mgp_lanczos_smallest_ex(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals, float* evecs,
                            float* resid, int32_t* info, float* block_evals, float* block_evecs, float* block_resid,
                            void* work, size_t work_bytes, void* stream);
/* Warm start (round 5).  The reference re-runs the whole eigendecomposition on every eval() (riemann_kernel.py:117-130); while
 * the graph bandwidth moves a little per training step the previous Rayleigh-Ritz block is already close.  warm_block [n, b]
 * (device: the block_evecs of an earlier call on the SAME sparsity pattern and row order) replaces the random start, warm_evals
 * [b] (host: its block_evals) set the first round's filter, so that the first round is already a full-strength one.  Same
 * outputs, return values and tolerances as mgp_lanczos_smallest_ex. */
int mgp_lanczos_smallest_warm(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals, float* evecs, float* resid,
                              int32_t* info, float* block_evals, float* block_evecs, float* block_resid, const float* warm_block,
                              const float* warm_evals, void* work, size_t work_bytes, void* stream);
/* Upper end of the eigensolver's Chebyshev filter: 1 (default) = lambda_max estimated from a 32-dimensional Krylov space
 * (+ 3 % or more), checked against the Ritz values of every round, with the Gershgorin bound as the fallback; 0 = the
 * Gershgorin bound (rigorous; about twice lambda_max on k-NN graph Laplacians, i.e. ~40 % more filter applies).  The residual
 * test uses the Gershgorin bound as its norm of L in both modes.  2 (tests only): half the estimate, a bound that is
 * certainly short, so that the fallback is exercised.  Process-wide. */
int mgp_lanczos_set_bound_mode(int mode);

/* k-step Lanczos tridiagonalisation of a precision-family operator with full re-orthogonalisation
 * (classical Gram-Schmidt against all previous vectors, twice).  Replaces
 * linear_operator.utils.lanczos.lanczos_tridiag as reached from
 * GraphLaplacianOperator.diagonalization (graph_laplacian_operator.py:132-135, Lanczos branch)
 * and from the stochastic-Lanczos-quadrature log-determinant of inv_quad_logdet
 * (train_model.py:68).  q0 [n] device start vector; alpha[steps], beta[steps] host;
 * Q_out (nullable) device [steps, n], row j = q_j.  Synchronises `stream`. */
/* P independent Lanczos runs at once (the probes of the stochastic log-determinant): Q0 [n, P] row-major,
 * P <= 16, steps <= 47; alpha / beta host [steps][P].  Same arithmetic per column as mgp_lanczos_tridiag,
 * one P-column SpMM chain per step instead of P single-vector chains.  Synchronises `stream`. */
size_t mgp_lanczos_tridiag_block_workspace_bytes(const mgp_operator_t* op, int P, int steps);
int mgp_lanczos_tridiag_block(const mgp_operator_t* op, const float* Q0, int P, int steps, float* alpha, float* beta,
                              void* work, size_t work_bytes, void* stream);
/* The same P runs for an operator the CALLER applies (round 5) -- the inverse of a Schur complement, every product of which is a CG
 * solve of its own: the stochastic log-determinant of the semi-supervised loss (train_model.py:68 through
 * schur_complement_operator.py:12-36).  mgp_blz_begin normalises the start block Q0 [n, P]; per step j = 0 .. steps - 1 the caller
 * reads q_j (mgp_blz_q: device pointer of the [n, P] block inside `work`), applies its operator and hands W = A q_j [n, P] to
 * mgp_blz_step (two passes of classical Gram-Schmidt against q_0 .. q_j, alpha_j, beta_j, q_j+1; W is overwritten);
 * mgp_blz_end copies alpha / beta (host [steps][P]) and synchronises `stream` -- nothing before it synchronises or reads back.
 * `work` (mgp_blz_workspace_bytes; 0 = unsupported shape: P <= 16, steps <= 47) is laid out identically at every call and must
 * not be touched in between. */
size_t mgp_blz_workspace_bytes(int64_t n, int P, int steps);
int mgp_blz_begin(const float* Q0, int64_t n, int P, int steps, void* work, size_t work_bytes, void* stream);
float* mgp_blz_q(int64_t n, int P, int steps, int j, void* work, size_t work_bytes);
int mgp_blz_step(float* W, int64_t n, int P, int steps, int j, void* work, size_t work_bytes, void* stream);
int mgp_blz_end(int64_t n, int P, int steps, float* alpha, float* beta, void* work, size_t work_bytes, void* stream);
size_t mgp_lanczos_tridiag_workspace_bytes(const mgp_operator_t* op, int steps);
int mgp_lanczos_tridiag(const mgp_operator_t* op, const float* q0, int steps, float* alpha,
                        float* beta, float* Q_out, void* work, size_t work_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Spectral features and the kernel block.
 * mgp_features_insample: Z = sqrt(N s / sum s) * Phi, s = (2 nu/kappa^2 + lambda)^-nu
 *   (riemann_kernel.py:134-136, riemann_matern_kernel.py:21-22); also the eval()-time
 *   post-processing Phi = normalise_cols(D^-1/2 U), lambda_0 = 0 (riemann_kernel.py:126-128)
 *   via mgp_eigvec_postprocess.
 * mgp_features_oos: fused Nystrom extension (graph_laplacian_operator.py:146-157) + bump
 *   modulation (riemann_kernel.py:138-147, torch_utils.py:38-41) without the [T,k,m] temporary.
 * mgp_kernel_block: K = Z1 Z2^T on the fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   (MatmulLinearOperator / LowRankRootLinearOperator evaluation, riemann_kernel.py:92-100). */
size_t mgp_eigvec_postprocess_work_floats(int m);   /* size of colnorm_work in floats */
int mgp_eigvec_postprocess(float* evecs, int64_t n, int m, const float* degree, float* colnorm_work,
                           void* stream);
int mgp_features_insample(const float* evals_dev, const float* evecs, int64_t n, int m, int nu,
                          float kappa, float* Z, void* stream);
int mgp_features_oos(const float* evals_dev, const float* evecs, int64_t n, int m, int nu,
                     float kappa, float eps, int normalization, const float* degree_unnorm,
                     const float* degree, const float* knn_d2, const int32_t* knn_idx, int64_t T,
                     int k, float bump_scale, float bump_decay, float* Z, void* stream);
int mgp_kernel_block(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale,
                     float* K, void* stream);
/* K = scale * Z1 Z2^T on the fp32 MFMA.  Which kernel runs (A/B and test knob; results agree to fp32 rounding, the three
 * LDS kernels bit for bit with each other): 0 (default) = by shape: the resident-operand kernel where the operands allow it
 * (16 <= m <= 128, m and n2 multiples of 4, n1 >= 64, n2 >= 32, 16-byte aligned: a wave keeps 64 rows of Z1 in registers and
 * streams Z2, no LDS), else the lean one-tile kernel, else the general one; 1 = one 128x128 tile per workgroup, by the lean
 * kernel (single-instruction staging loads, 16-byte stores) where the operands allow it (n1, n2 >= 128, m and n2 multiples of
 * 4, 16-byte aligned) and by the general one elsewhere; 2 = where they allow it, one 512-thread workgroup per CU whose two
 * halves walk tiles and take turns on the matrix pipe; 3 = as 2 with every store dropped (timing only); 4 = the general kernel
 * always; 5 = as 0; 6 = as 5 with every store dropped (timing only). */
int mgp_kernel_block_set_pipe(int mode);
int mgp_kernel_diag(const float* Z1, const float* Z2, int64_t n, int m, float scale, float* out,
                    void* stream);
/* y = alpha * Z (Z^T x) + beta * x : the low-rank covariance K + sigma^2 I applied matrix-free */
size_t mgp_lowrank_workspace_bytes(int m, int C);
int mgp_lowrank_apply(const float* Z, int64_t n, int m, const float* X, int C, float alpha,
                      float beta, float* Y, void* work, size_t work_bytes, void* stream);
/* Woodbury solve of (s Z Z^T + noise I) x = v, the way gpytorch evaluates the reference's spectral kernel
 * (LowRankRootAddedDiagLinearOperator), in two device steps around an m x m fp64 system:
 *   mgp_gram_f64          G = A^T A for a tall block A [n,b] (b <= 512), fp64 accumulation -> device double [b,b];
 *                         with A = [Z | v] one pass gives Z^T Z, Z^T v (and v^T v)
 *   mgp_lowrank_residual  out = scale * (V - Z T), T device double [m,C] (m C <= 6144), row sums in fp64 */
size_t mgp_gram_workspace_bytes(int64_t n, int b);
int mgp_gram_f64(const float* A, int64_t n, int b, double* G, void* work, size_t work_bytes, void* stream);
/* the partial Gram blocks of mgp_gram_f64 and of the eigensolver on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; default 1) or by
 * fp64 vector FMAs (0: A/B runs, tests); returns the previous setting */
int mgp_gram_set_mfma(int on);
int mgp_lowrank_residual(const float* Z, int64_t n, int m, const double* T, const float* V, int C, double scale,
                         float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU (one process per GPU, RCCL over xGMI): row-partitioned operator apply and CG.
 * The reference has no distributed code; this is the north-star's
 * "CG SpMV row-partitions across the 8 GPUs of one node with one RCCL collective per iteration".
 *   - rank p owns global rows [p*n_loc, (p+1)*n_loc) of L (op_local->L: local CSR slice whose
 *     column ids are global; op_local->pre/post and every vector have GLOBAL length world*n_loc and
 *     are replicated on every rank);
 *   - per SpMM launch each rank computes its row slice, then ONE grouped ncclAllGather assembles
 *     the output on every rank; on the last launch of a chain the dot-product partials ride in the
 *     same group, so CG needs no separate scalar reduction (the vector updates are replicated and
 *     every rank takes bit-identical decisions);
 *   - the communicator comes from ncclCommInitRank over a unique id the host distributes. */
int mgp_dist_unique_id_bytes(void);
int mgp_dist_unique_id(void* id_out);                          /* rank 0 */
int mgp_dist_init(int rank, int world, const void* id_bytes, void** comm_out);
int mgp_dist_destroy(void* comm);
/* size of the communicator, this process's rank in it and its HIP device, as RCCL reports them (ncclCommCount /
 * ncclCommUserRank / ncclCommCuDevice): what bench.py's N > 1 line carries as `rccl_ranks` */
int mgp_dist_comm_info(void* comm, int32_t* count, int32_t* user_rank, int32_t* device);
int mgp_dist_allgather(void* comm, int rank, int world, float* buf, int64_t count_per_rank,
                       void* stream);                          /* in place, slice p at p*count */
int mgp_operator_apply_part(const mgp_operator_t* op_local, void* comm, int rank, int world,
                            const float* X, int C, float* Y, void* work, size_t work_bytes,
                            void* stream);                     /* work: 4 global-length buffers */
size_t mgp_cg_dist_workspace_bytes(const mgp_operator_t* op_local, int C, int world);
int mgp_cg_plan_create_dist(const mgp_operator_t* op_local, int C, const float* minv,
                            const mgp_cg_params_t* params, void* comm, int rank, int world,
                            void* work, size_t work_bytes, void* stream, void** plan_out);
/* solve / x / destroy: mgp_cg_plan_solve, mgp_cg_plan_x, mgp_cg_plan_destroy (B, X global length) */

/* ---------------------------------------------------------------------------------------------
 * Partitioned pipelined CG (csrc/pcg.hip): the multi-GPU form that can scale -- vectors AND rows partitioned,
 * ONE grouped RCCL all-gather per iteration (the w slices + the dot partials), ghost layers instead of a second
 * exchange for nu >= 2, the iteration loop captured in a hipGraph.  C = 1, forms 0 and 2.  Same call sites as
 * mgp_cg_plan_* above (the (K + s I) x = y solve in precision form).
 *   op->L        tile view of the WHOLE padded graph over this rank's row order [own rows, ghost layer 1, ...,
 *                rest] (mgp_graph_tiles with `order`: tile_rowptr / tile_vals / tile_rowid set); L.n = global
 *                (padded) node count; pre / post / diag at the global length
 *   launch_rows  [nu]: rows of that view launch s of the SpMV chain covers = own rows + (nu - 1 - s) ghost layers,
 *                rounded up to the tile height; launch_rows[nu - 1] >= n_loc
 *   row0, n_loc  this rank's contiguous row block (row0 = rank * n_loc, equal blocks); n_real: rows >= n_real are
 *                padding
 *   comm         ncclComm_t, or NULL: world == 1, or VIRTUAL ranks of one process sharing `shared`
 *                (mgp_pcg_shared_floats floats: gathered w and partials, double-buffered), driven phase by phase
 *                with mgp_pcg_plan_enqueue -- what the single-GPU tests of the partition logic use
 *   B            right-hand side at the global length on every rank; the solution's own rows are valid in
 *                mgp_pcg_plan_x()[row0 .. row0 + n_loc) and copied to X_loc when given */
size_t mgp_pcg_shared_floats(int64_t n_glob, int64_t n_loc, int world);
size_t mgp_pcg_workspace_bytes(int64_t n_glob, int64_t n_loc, int world);
/* recurrence: 0 = pipelined (ONE grouped all-gather per iteration; stagnates early on ill-conditioned systems in
 *             fp32: residual replacement for chunks >= 16 iterations, refinement rounds with max_refine),
 *             1 = Chronopoulos-Gear, the recurrence of mgp_cg_plan_* (robust; TWO collectives per iteration: the
 *             gathered vector + gamma partials, and ~n_loc / 64 delta partials per rank) */
int mgp_pcg_plan_create(const mgp_operator_t* op, const int64_t* launch_rows, int64_t row0, int64_t n_loc,
                        int64_t n_real, void* comm, int rank, int world, float* shared, int recurrence,
                        const mgp_cg_params_t* params, void* work, size_t work_bytes, void* stream,
                        void** plan_out);
int mgp_pcg_plan_solve(void* plan, const float* B, float* X_loc, int32_t* iters, float* resid, int32_t* status);
int mgp_pcg_plan_enqueue(void* plan, int phase, int par, const float* B); /* 0: start; 1: iteration (recurrence 1: its
                                                                             SpMV chain); 2: recurrence 1: its update */
int mgp_pcg_plan_poll(void* plan, int32_t* iters, float* resid, int32_t* status); /* 1 = undecided */
float* mgp_pcg_plan_x(void* plan);
int mgp_pcg_plan_destroy(void* plan);
int mgp_pcg_plan_poisoned(void* plan);   /* as mgp_cg_plan_poisoned */

#ifdef __cplusplus
}
#endif
#endif /* MGP_HIP_H */
