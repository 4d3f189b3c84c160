// Internal (non-exported) entry points shared between the translation units of libmgp_hip.
#pragma once
#include <vector>
#include "mgp_hip.h"

// mgp_spmm_fused plus: `skip` (device flag: the launch is a no-op when non-zero) and `tick`
// (device counter incremented once per non-skipped launch) -- used by the CG iteration graph.
int mgp_spmm_fused_ex(const mgp_csr_t* L, const float* X, int C, float* Y, float a, float b,
                      const float* pre, const float* post, const float* base, float cb, float co,
                      const float* dotw, float* dot_partials, const int* skip, int* tick, void* stream);

// row-partitioned form: L holds the LOCAL rows [0, L->n) of a larger operator whose vectors are
// global; local row r is global row r + row_offset (columns in L->col are global already)
int mgp_spmm_fused_part(const mgp_csr_t* L, int64_t row_offset, const float* X, int C, float* Y, float a, float b,
                        const float* pre, const float* post, const float* base, float cb, float co,
                        const float* dotw, float* dot_partials, const int* skip, int* tick, void* stream);

// Init-free CG solve (cg.hip): the FIRST operator apply of a solve reads the caller's right-hand side directly
// (no cg_init launch).  Launch 0 of the chain stores its raw input rows to copy_x (r = b); the chain's last launch
// also writes per-workgroup partials of sum dotw^2 (||b||^2) and resets the iteration state to {1, 0, 0}.
// record (nullable, MGP_SPMM_RECORD_BYTES): the launch arguments, for mgp_spmm_patch_node.  Tile kernel only.
#define MGP_SPMM_RECORD_BYTES 512
struct MgpFirst {
  float* copy_x;
  float* dot2_partials;
  int tick_reset;
  void* record;
};
int mgp_spmm_fused_first(const mgp_csr_t* L, int64_t row_offset, const float* X, int C, float* Y, float a, float b,
                         const float* pre, const float* post, const float* base, float cb, float co,
                         const float* dotw, float* dot_partials, const int* skip, int* tick,
                         const MgpFirst* first, void* stream);
int mgp_spmm_patch_node(void* exec, void* node, const void* record, const float* old_ptr, const float* new_ptr);
// first apply of an init-free solve: launch 0 reads `rhs` (pre-scaled in the kernel by op->pre) and copies it to
// r_copy; later launches take r_copy as base / dot weight; partials of r . A r and ||r||^2; state reset
int mgp_operator_apply_first(const mgp_operator_t* op, const float* rhs, float* r_copy, float* Y, float* dot_partials,
                             float* dot2_partials, int* state, void* record, void* work, size_t work_bytes, void* stream);
// 1 when the C == 1 tile kernel would run on L; launch geometry of that kernel
int mgp_tile_plan(const mgp_csr_t* L, int C, int* grid, int* tiles_per_block, size_t* lds_bytes);

// workgroups that write dot partials for this CSR (depends on whether the tile kernel is used)
int mgp_spmm_dot_blocks_for(const mgp_csr_t* L, int C);

// operator chain with the same hooks on its LAST SpMM
int mgp_operator_apply_ex(const mgp_operator_t* op, const float* X, int C, float* Y, const float* dotw,
                          float* dot_partials, const int* skip, int* tick, void* work, size_t work_bytes,
                          void* stream);

// as above with Xs = diag(op->pre) X precomputed by the caller (nullable)
int mgp_operator_apply_ex2(const mgp_operator_t* op, const float* X, const float* Xs, int C, float* Y,
                           const float* dotw, float* dot_partials, const int* skip, int* tick, void* work,
                           size_t work_bytes, void* stream);

// Row partition over the ranks of one node: rank p owns global rows [p * n_loc, (p+1) * n_loc);
// vectors are replicated (global length world * n_loc); after every local SpMM the output slices
// (and, on the last launch of a chain, the dot-product partials) are all-gathered over RCCL.
struct MgpDist {
  void* comm;          // ncclComm_t
  int rank, world;
  int64_t n_loc;
  int64_t row_offset;  // rank * n_loc
};

int mgp_operator_apply_dist(const mgp_operator_t* op, const MgpDist* d, const float* X, const float* Xs, int C,
                            float* Y, const float* dotw, float* dot_partials, int nb_loc, const int* skip, int* tick,
                            void* work, size_t work_bytes, void* stream);
int mgp_dist_allgather_f32(const MgpDist* d, float* buf, int64_t count_per_rank, void* stream);

// k-NN internals (knn.hip / knn_lowd.hip)
int mgp_knn_bruteforce(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D,
                       int32_t* I, void* work, size_t work_bytes, int64_t* stats, void* stream,
                       const void* index = nullptr, size_t index_bytes = 0, bool allow_filter = true);
int mgp_knn_lowd_eligible(int64_t N, int64_t n, int d, int k);
size_t mgp_knn_lowd_workspace_bytes(int64_t N, int64_t n, int d, int k);
int mgp_knn_lowd(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D, int32_t* I, void* work,
                 size_t work_bytes, std::vector<int32_t>* over_rows, void* stream);
int mgp_knn_gather_rows(const float* src, const int32_t* rows_dev, int64_t m, int w, float* dst, void* stream);
int mgp_knn_scatter_rows(const float* Ds, const int32_t* Is, const int32_t* rows_dev, int64_t m, int k, float* D,
                         int32_t* I, void* stream);

// candidate distances on the matrix cores (knn_mfma.hip): centred bf16 h/l split of points and queries
struct MgpKnnMfma {
  int dpad;
  uint16_t *Ph, *Pl, *Qh, *Ql;   // [N, dpad] / [chunk rows, dpad]
  float *pn2, *qn2;              // |c|^2
  float* mu;                     // [d] column means of the points
  float* partial;
  unsigned* r2max;               // bits of max |c_y|^2 over the points
  uint16_t *Sh, *Sl;             // candidate filter: split of every stride-th point [S, dpad]
  float* sn2;
};
int mgp_knn_mfma_dpad(int d);
size_t mgp_knn_mfma_bytes(int64_t N, int64_t qc, int d);
int mgp_knn_mfma_take(MgpArena& ar, int64_t N, int64_t qc, int d, MgpKnnMfma* m);
size_t mgp_knn_mfma_index_bytes(int64_t N, int d);
int mgp_knn_mfma_index_take(MgpArena& ar, int64_t N, int d, MgpKnnMfma* m);
size_t mgp_knn_mfma_query_bytes(int64_t qc, int d);
int mgp_knn_mfma_query_take(MgpArena& ar, int64_t qc, int d, MgpKnnMfma* m);
int mgp_knn_mfma_prepare_points(const float* db, int64_t N, int d, const MgpKnnMfma& m, hipStream_t st);
int mgp_knn_mfma_prepare_queries(const float* q, int64_t rows, int d, const MgpKnnMfma& m, hipStream_t st);
int mgp_knn_mfma_tiles(const MgpKnnMfma& m, int64_t rows, int64_t N, float* slab, int64_t ld, hipStream_t st, bool sym = false);
void mgp_knn_mfma_bound(int dpad, double* alpha, double* beta);
// candidate filter (no key slab): sampled points, their keys, the filtered key pass
size_t mgp_knn_mfma_sample_bytes(int64_t S, int d);
int mgp_knn_mfma_sample_take(MgpArena& ar, int64_t S, int d, MgpKnnMfma* m);
int mgp_knn_mfma_prepare_sample(const float* db, int64_t S, int64_t stride, int d, const MgpKnnMfma& m, hipStream_t st);
int mgp_knn_mfma_sample_tiles(const MgpKnnMfma& m, int64_t rows, int64_t S, float* samp, int64_t ld, hipStream_t st, bool sym);
int mgp_knn_mfma_tiles_filtered(const MgpKnnMfma& m, int64_t rows, int64_t N, const float* bounds, void* log, unsigned* cursor,
                                void* table, unsigned shard_cap, int* overflow, hipStream_t st, bool sym);
int mgp_knn_mfma_regroup(const void* table, const void* log, int64_t rows, int64_t N, void* lists, int* counts, int cap,
                         hipStream_t st, bool sym);
size_t mgp_knn_mfma_table_entries(int64_t rows, int64_t N);
int mgp_knn_mfma_log_shards(void);

// fp64 operator apply from the fp32 matrix (true residual of the CG refinement); work64 = 4 n C doubles
int mgp_operator_apply_f64(const mgp_operator_t* op, const double* X, int C, double* Y, double* work64, void* stream);
