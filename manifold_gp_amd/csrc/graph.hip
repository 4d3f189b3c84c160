// k-NN lists -> symmetrised graph: reference COO view + padded full-symmetric CSR.
//
// Replaces NearestNeighbors.graph (manifold_gp/utils/nearest_neighbors.py:39-55): drop column
// 0, orient every directed edge as (min,max), torch_sparse.coalesce(op='mean') = sort by
// (row,col) + merge duplicates with the fp32 mean.  The reference stops at the upper-triangular
// COO and pays for it with two atomic scatters per matvec; here the same edge set is also
// expanded once into a full symmetric CSR (both directions, columns ascending, rows padded to 4
// entries) so that every later pass is a gather-only row sweep.
//
// Sorting uses rocPRIM's device radix sort through hipcub (library primitive); everything else
// (key build, run heads, segmented mean, row search, fill) is hand written.  One-off setup work:
// clarity over the last percent.
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <vector>
#include <mutex>
#include "mgp_common.h"

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n) {
  int64_t g = mgp_cdiv(n, kBlock);
  return (int)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

__global__ void make_directed_keys(const float* __restrict__ D, const int32_t* __restrict__ I,
                                   int64_t n, int k, int first_col, uint64_t* __restrict__ keys,
                                   float* __restrict__ vals) {
  const int km1 = k - first_col;
  const int64_t total = n * km1;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / km1;
    const int c = (int)(t % km1) + first_col;  // nearest_neighbors.py:42-43 drops column 0 unless self_loop
    const uint32_t j = (uint32_t)I[i * k + c];
    const uint32_t r = (uint32_t)i;
    const uint32_t lo = r < j ? r : j, hi = r < j ? j : r;   // :48-50 orient row<col
    keys[t] = ((uint64_t)lo << 32) | hi;
    vals[t] = D[i * k + c];
  }
}

__global__ void mark_heads(const uint64_t* __restrict__ keys, int64_t total, int32_t* __restrict__ head) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x)
    head[t] = (t == 0 || keys[t] != keys[t - 1]) ? 1 : 0;
}

// one thread per run head: fp32 sum of the run in sorted (stable) order, divided by the count
__global__ void segment_mean(const uint64_t* __restrict__ keys, const float* __restrict__ vals,
                             const int32_t* __restrict__ head, const int32_t* __restrict__ uidx,
                             int64_t total, int32_t* __restrict__ tri_row, int32_t* __restrict__ tri_col,
                             float* __restrict__ tri_val) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * blockDim.x) {
    if (!head[t]) continue;
    const uint64_t key = keys[t];
    float s = vals[t];
    int cnt = 1;
    for (int64_t u = t + 1; u < total && keys[u] == key; ++u) { s += vals[u]; ++cnt; }
    const int32_t o = uidx[t];
    tri_row[o] = (int32_t)(key >> 32);
    tri_col[o] = (int32_t)(key & 0xffffffffu);
    tri_val[o] = s / (float)cnt;
  }
}

__global__ void expand_both_directions(const int32_t* __restrict__ tri_row, const int32_t* __restrict__ tri_col,
                                       int64_t M, uint64_t* __restrict__ keys, int32_t* __restrict__ eids) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < M;
       e += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t r = (uint32_t)tri_row[e], c = (uint32_t)tri_col[e];
    keys[2 * e] = (r << 32) | c;
    keys[2 * e + 1] = (c << 32) | r;
    eids[2 * e] = (int32_t)e;
    eids[2 * e + 1] = (int32_t)e;
  }
}

// rowstart[r] = first sorted position whose key >= (r, 0)
__global__ void row_search(const uint64_t* __restrict__ keys, int64_t total, int64_t n, int32_t* __restrict__ rowstart) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t target = (uint64_t)r << 32;
    int64_t lo = 0, hi = total;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    rowstart[r] = (int32_t)lo;
  }
}

__global__ void padded_counts(const int32_t* __restrict__ rowstart, int64_t n, int32_t* __restrict__ cnt) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= n; r += (int64_t)gridDim.x * blockDim.x) {
    if (r == n) { cnt[r] = 0; continue; }
    const int c = rowstart[r + 1] - rowstart[r];
    cnt[r] = (c + MGP_PAD - 1) / MGP_PAD * MGP_PAD;
  }
}

__global__ void fill_csr(const uint64_t* __restrict__ keys, const int32_t* __restrict__ eids,
                         const float* __restrict__ tri_val, const int32_t* __restrict__ rowstart,
                         const int32_t* __restrict__ rowptr, int64_t n, int32_t* __restrict__ col,
                         float* __restrict__ d2, int32_t* __restrict__ eid) {
  // one 16-lane group per row: copy the sorted entries, then write the padding (col = the row itself,
  // d2 = +inf -> weight 0, eid = -1)
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = g; r < n; r += ng) {
    const int s = rowstart[r], cnt = rowstart[r + 1] - s;
    const int o = rowptr[r], cap = rowptr[r + 1] - o;
    for (int i = lane; i < cap; i += 16) {
      if (i < cnt) {
        const uint64_t key = keys[s + i];
        const int32_t e = eids[s + i];
        col[o + i] = (int32_t)(key & 0xffffffffu);
        d2[o + i] = tri_val[e];
        eid[o + i] = e;
      } else {
        col[o + i] = (int32_t)r;
        d2[o + i] = INFINITY;
        eid[o + i] = -1;
      }
    }
  }
}

struct GraphWork {
  uint64_t *keys_a, *keys_b;
  float *vals_a, *vals_b;     // also reused as int32 payload
  int32_t *head, *uidx, *rowstart, *cnt;
  void* cub;
  size_t cub_bytes;
};

size_t cub_bytes_for(int64_t items) {
  size_t a = 0, b = 0, c = 0;
  hipcub::DoubleBuffer<uint64_t> k(nullptr, nullptr);
  hipcub::DoubleBuffer<float> v(nullptr, nullptr);
  hipcub::DoubleBuffer<int32_t> vi(nullptr, nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k, v, (int)items, 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, vi, (int)items, 0, 64, (hipStream_t)0);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, c, (int32_t*)nullptr, (int32_t*)nullptr, (int)items,
                                         (hipStream_t)0);
  size_t m = a > b ? a : b;
  m = m > c ? m : c;
  return mgp_align(m + 1024);
}

size_t graph_bytes(int64_t items, int64_t n) {
  size_t b = 0;
  b += 2 * mgp_align(items * sizeof(uint64_t));
  b += 2 * mgp_align(items * sizeof(float));
  b += 2 * mgp_align((items + 1) * sizeof(int32_t));
  b += 2 * mgp_align((n + 2) * sizeof(int32_t));
  b += cub_bytes_for(items > n + 2 ? items : n + 2);
  return b + 4096;
}

bool carve(GraphWork& w, void* work, size_t bytes, int64_t items, int64_t n) {
  MgpArena ar(work, bytes);
  w.keys_a = ar.take<uint64_t>(items);
  w.keys_b = ar.take<uint64_t>(items);
  w.vals_a = ar.take<float>(items);
  w.vals_b = ar.take<float>(items);
  w.head = ar.take<int32_t>(items + 1);
  w.uidx = ar.take<int32_t>(items + 1);
  w.rowstart = ar.take<int32_t>(n + 2);
  w.cnt = ar.take<int32_t>(n + 2);
  w.cub_bytes = cub_bytes_for(items > n + 2 ? items : n + 2);
  w.cub = ar.take<char>(w.cub_bytes);
  return ar.ok();
}

int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n) ++b;
  return b;
}

// tri_* (device, sorted unique) -> padded CSR (every row padded to 4 entries).  keys_a/keys_b/vals_* are
// scratch of >= 2M items.
int csr_from_tri(GraphWork& w, const int32_t* tri_row, const int32_t* tri_col, const float* tri_val,
                 int64_t M, int64_t n, int32_t* rowptr, int32_t* col, float* d2, int32_t* eid, int64_t* nnz,
                 hipStream_t st) {
  const int64_t items = 2 * M;
  if (items > 0) {
    hipLaunchKernelGGL(expand_both_directions, dim3(grid_for(M)), dim3(kBlock), 0, st, tri_row, tri_col, M,
                       w.keys_a, reinterpret_cast<int32_t*>(w.vals_a));
    MGP_LAUNCH_CHECK();
    hipcub::DoubleBuffer<uint64_t> kb(w.keys_a, w.keys_b);
    hipcub::DoubleBuffer<int32_t> vb(reinterpret_cast<int32_t*>(w.vals_a), reinterpret_cast<int32_t*>(w.vals_b));
    size_t tb = w.cub_bytes;
    MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, kb, vb, (int)items, 0, 32 + bits_for(n), st));
    const uint64_t* keys = kb.Current();
    const int32_t* eids = vb.Current();
    hipLaunchKernelGGL(row_search, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, keys, items, n, w.rowstart);
    MGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(padded_counts, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, w.rowstart, n, w.cnt);
    MGP_LAUNCH_CHECK();
    tb = w.cub_bytes;
    MGP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, tb, w.cnt, rowptr, (int)(n + 1), st));
    hipLaunchKernelGGL(fill_csr, dim3(grid_for(n * 16)), dim3(kBlock), 0, st, keys, eids, tri_val, w.rowstart, rowptr, n,
                       col, d2, eid);
    MGP_LAUNCH_CHECK();
  } else {
    MGP_HIP_TRY(hipMemsetAsync(rowptr, 0, (n + 1) * sizeof(int32_t), st));
  }
  int32_t last = 0;
  MGP_HIP_TRY(hipMemcpyAsync(&last, rowptr + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  *nnz = last;
  return MGP_OK;
}

}  // namespace

extern "C" size_t mgp_graph_workspace_bytes(int64_t n, int k) {
  if (n <= 0 || k < 2) return 0;
  return graph_bytes(2 * n * (int64_t)(k - 1), n);
}

extern "C" int mgp_graph_build(const float* D, const int32_t* I, int64_t n, int k, int32_t* tri_row, int32_t* tri_col,
                               float* tri_val, int64_t* M, int32_t* rowptr, int32_t* col, float* d2, int32_t* eid,
                               int64_t* nnz, void* work, size_t work_bytes, void* stream) {
  if (!D || !I || !tri_row || !tri_col || !tri_val || !M || !rowptr || !col || !d2 || !eid || !nnz || !work)
    return MGP_ERR_ARG;
  if (n <= 0 || k < 2 || n * (int64_t)(k - 1) * 2 > 0x7fffffff) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  const int64_t total = n * (int64_t)(k - 1);
  GraphWork w;
  if (!carve(w, work, work_bytes, 2 * total, n)) return MGP_ERR_WORKSPACE;

  hipLaunchKernelGGL(make_directed_keys, dim3(grid_for(total)), dim3(kBlock), 0, st, D, I, n, k, 1, w.keys_a, w.vals_a);
  MGP_LAUNCH_CHECK();
  hipcub::DoubleBuffer<uint64_t> kb(w.keys_a, w.keys_b);
  hipcub::DoubleBuffer<float> vb(w.vals_a, w.vals_b);
  size_t tb = w.cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, kb, vb, (int)total, 0, 32 + bits_for(n), st));
  const uint64_t* keys = kb.Current();
  const float* vals = vb.Current();
  hipLaunchKernelGGL(mark_heads, dim3(grid_for(total)), dim3(kBlock), 0, st, keys, total, w.head);
  MGP_LAUNCH_CHECK();
  tb = w.cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, tb, w.head, w.uidx, (int)total, st));
  hipLaunchKernelGGL(segment_mean, dim3(grid_for(total)), dim3(kBlock), 0, st, keys, vals, w.head, w.uidx, total,
                     tri_row, tri_col, tri_val);
  MGP_LAUNCH_CHECK();
  int32_t last_idx = 0, last_head = 0;
  MGP_HIP_TRY(hipMemcpyAsync(&last_idx, w.uidx + (total - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipMemcpyAsync(&last_head, w.head + (total - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  *M = (int64_t)last_idx + last_head;
  return csr_from_tri(w, tri_row, tri_col, tri_val, *M, n, rowptr, col, d2, eid, nnz, st);
}

namespace {
// symmetric = False (nearest_neighbors.py:45-46, 53): every (query, neighbour) pair as it stands, row-major
__global__ void directed_edges(const float* __restrict__ D, const int32_t* __restrict__ I, int64_t n, int k, int first_col,
                               int32_t* __restrict__ row, int32_t* __restrict__ col, float* __restrict__ val) {
  const int km1 = k - first_col;
  const int64_t total = n * km1;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = t / km1;
    const int c = (int)(t % km1) + first_col;
    row[t] = (int32_t)i;
    col[t] = I[i * k + c];
    val[t] = D[i * k + c];
  }
}
}  // namespace

// The edge list of NearestNeighbors.graph for ANY flag combination (nearest_neighbors.py:39-55), without the CSR that
// mgp_graph_build adds for the default one: self_loop keeps column 0 of the lists (the point itself at distance 0: an (i, i)
// entry), symmetric orients every pair as (min, max), sorts and merges duplicates with the fp32 mean in sorted order
// (torch_sparse.coalesce(op='mean')); not symmetric: the n (k - first) directed pairs in row-major order.
// out_* hold n (k - first) entries at most; *M = the number written.  Workspace: mgp_graph_workspace_bytes(n, k + 1).
extern "C" int mgp_graph_edges(const float* D, const int32_t* I, int64_t n, int k, int symmetric, int self_loop,
                               int32_t* out_row, int32_t* out_col, float* out_val, int64_t* M, void* work, size_t work_bytes,
                               void* stream) {
  if (!D || !I || !out_row || !out_col || !out_val || !M) return MGP_ERR_ARG;
  const int first = self_loop ? 0 : 1;
  if (n <= 0 || k <= first || n * (int64_t)(k - first) * 2 > 0x7fffffff) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  const int64_t total = n * (int64_t)(k - first);
  if (!symmetric) {
    hipLaunchKernelGGL(directed_edges, dim3(grid_for(total)), dim3(kBlock), 0, st, D, I, n, k, first, out_row, out_col, out_val);
    MGP_LAUNCH_CHECK();
    *M = total;
    return MGP_OK;
  }
  if (!work) return MGP_ERR_ARG;
  GraphWork w;
  if (!carve(w, work, work_bytes, 2 * total, n)) return MGP_ERR_WORKSPACE;
  hipLaunchKernelGGL(make_directed_keys, dim3(grid_for(total)), dim3(kBlock), 0, st, D, I, n, k, first, w.keys_a, w.vals_a);
  MGP_LAUNCH_CHECK();
  hipcub::DoubleBuffer<uint64_t> kb(w.keys_a, w.keys_b);
  hipcub::DoubleBuffer<float> vb(w.vals_a, w.vals_b);
  size_t tb = w.cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.cub, tb, kb, vb, (int)total, 0, 32 + bits_for(n), st));
  hipLaunchKernelGGL(mark_heads, dim3(grid_for(total)), dim3(kBlock), 0, st, kb.Current(), total, w.head);
  MGP_LAUNCH_CHECK();
  tb = w.cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, tb, w.head, w.uidx, (int)total, st));
  hipLaunchKernelGGL(segment_mean, dim3(grid_for(total)), dim3(kBlock), 0, st, kb.Current(), vb.Current(), w.head, w.uidx, total,
                     out_row, out_col, out_val);
  MGP_LAUNCH_CHECK();
  int32_t last_idx = 0, last_head = 0;
  MGP_HIP_TRY(hipMemcpyAsync(&last_idx, w.uidx + (total - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipMemcpyAsync(&last_head, w.head + (total - 1), sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  *M = (int64_t)last_idx + last_head;
  return MGP_OK;
}

extern "C" int mgp_graph_from_coo(const int32_t* tri_row, const int32_t* tri_col, const float* tri_val,
                                  int64_t M, int64_t n, int32_t* rowptr, int32_t* col, float* d2, int32_t* eid,
                                  int64_t* nnz, void* work, size_t work_bytes, void* stream) {
  if (!tri_row || !tri_col || !tri_val || !rowptr || !col || !d2 || !eid || !nnz || !work) return MGP_ERR_ARG;
  if (n <= 0 || M < 0 || 2 * M > 0x7fffffff) return MGP_ERR_ARG;
  GraphWork w;
  if (!carve(w, work, work_bytes, 2 * M > 0 ? 2 * M : 1, n)) return MGP_ERR_WORKSPACE;
  return csr_from_tri(w, tri_row, tri_col, tri_val, M, n, rowptr, col, d2, eid, nnz, mgp_stream(stream));
}

extern "C" size_t mgp_graph_coo_workspace_bytes(int64_t n, int64_t M) {
  if (n <= 0) return 0;
  return graph_bytes(2 * M > 0 ? 2 * M : 1, n);
}

// ------------------------------------------------------------------------------------------------
// Row-tile column dictionaries for the C == 1 SpMV (spmm.hip: spmv_tile_kernel).
// A tile = `tile_rows` consecutive rows.  For every tile: the ascending list of the DISTINCT columns
// its entries reference (tile_cols[tile_ptr[t] .. tile_ptr[t+1])) and, per entry, the 16-bit position
// of its column in that list (lid).  The SpMV stages x[tile_cols] in LDS once per tile and every
// per-entry gather becomes a ds_read: on a k-NN graph a tile's rows share most of their neighbours,
// so the texture-path accesses drop from one per entry to one per distinct column.
namespace {

// ordered tiles: position p of the tile order holds original row row_order[p]
__global__ void order_lengths(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ row_order, int64_t n,
                              int32_t* __restrict__ len) {
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p <= n; p += (int64_t)gridDim.x * blockDim.x)
    len[p] = p < n ? rowptr[row_order[p] + 1] - rowptr[row_order[p]] : 0;
}

// keys / entry map in tile order: entry i of position p <- entry rowptr[row_order[p]] + i of the CSR
__global__ void tile_keys_ordered(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                  const int32_t* __restrict__ row_order, const int32_t* __restrict__ tile_rowptr,
                                  int64_t n, int tile_rows, uint64_t* __restrict__ keys, int32_t* __restrict__ pos,
                                  int32_t* __restrict__ emap) {
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t p = g; p < n; p += ng) {
    const uint64_t tile = (uint64_t)(p / tile_rows);
    const int src = rowptr[row_order[p]];
    const int dst = tile_rowptr[p], cnt = tile_rowptr[p + 1] - dst;
    for (int i = lane; i < cnt; i += 16) {
      keys[dst + i] = (tile << 32) | (uint32_t)col[src + i];
      pos[dst + i] = dst + i;
      emap[dst + i] = src + i;
    }
  }
}

__global__ void tile_keys(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t n,
                          int tile_rows, uint64_t* __restrict__ keys, int32_t* __restrict__ pos) {
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = g; r < n; r += ng) {
    const uint64_t tile = (uint64_t)(r / tile_rows);
    const int s = rowptr[r], e = rowptr[r + 1];
    for (int i = s + lane; i < e; i += 16) {
      keys[i] = (tile << 32) | (uint32_t)col[i];
      pos[i] = i;
    }
  }
}

// tile_ptr[t] = number of distinct (tile, column) pairs before tile t
__global__ void tile_starts(const uint64_t* __restrict__ keys, const int32_t* __restrict__ uidx,
                            const int32_t* __restrict__ head, int64_t total, int64_t ntiles,
                            int32_t* __restrict__ tile_ptr) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= ntiles;
       t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t target = (uint64_t)t << 32;
    int64_t lo = 0, hi = total;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (keys[mid] < target) lo = mid + 1; else hi = mid;
    }
    tile_ptr[t] = lo < total ? uidx[lo] : uidx[total - 1] + head[total - 1];
  }
}

__global__ void tile_fill(const uint64_t* __restrict__ keys, const int32_t* __restrict__ pos,
                          const int32_t* __restrict__ head, const int32_t* __restrict__ uidx,
                          const int32_t* __restrict__ tile_ptr, int64_t total, int32_t* __restrict__ tile_cols,
                          uint16_t* __restrict__ lid) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = keys[i];
    const int32_t d = uidx[i] + head[i] - 1;          // index of this (tile, column) pair
    if (head[i]) tile_cols[d] = (int32_t)(key & 0xffffffffu);
    const int32_t local = d - tile_ptr[key >> 32];
    lid[pos[i]] = (uint16_t)(local > 65535 ? 65535 : local);
  }
}

__global__ void tile_extents(const int32_t* __restrict__ rowptr /* in tile order */, const int32_t* __restrict__ tile_ptr, int64_t n,
                             int tile_rows, int64_t ntiles, int32_t* __restrict__ maxima) {
  int mc = 0, me = 0;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ntiles;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r0 = t * tile_rows;
    const int64_t r1 = r0 + tile_rows < n ? r0 + tile_rows : n;
    const int c = tile_ptr[t + 1] - tile_ptr[t];
    const int e = rowptr[r1] - rowptr[r0];
    mc = c > mc ? c : mc;
    me = e > me ? e : me;
  }
  atomicMax(&maxima[0], mc);
  atomicMax(&maxima[1], me);
}

size_t tile_bytes(int64_t n, int64_t nnz) {
  const int64_t items = nnz > n + 2 ? nnz : n + 2;      // some scratch doubles as per-row storage
  size_t b = 2 * mgp_align(items * sizeof(uint64_t)) + 4 * mgp_align((items + 1) * sizeof(int32_t));
  return b + cub_bytes_for(items) + 4096;
}

}  // namespace

extern "C" size_t mgp_graph_tiles_workspace_bytes(int64_t n, int64_t nnz) { return (n > 0 && nnz > 0) ? tile_bytes(n, nnz) : 0; }

extern "C" int mgp_graph_tiles(int64_t n, const int32_t* rowptr, const int32_t* col, int64_t nnz, int tile_rows,
                               const int32_t* row_order, int32_t* tile_rowptr, int32_t* emap, int32_t* tile_ptr,
                               int32_t* tile_cols, uint16_t* lid, int64_t* total_cols, int32_t* max_cols,
                               int32_t* max_entries, void* work, size_t work_bytes, void* stream) {
  if (!rowptr || !col || !tile_ptr || !tile_cols || !lid || !total_cols || !max_cols || !max_entries || !work)
    return MGP_ERR_ARG;
  if (row_order && (!tile_rowptr || !emap)) return MGP_ERR_ARG;
  if (n <= 0 || nnz <= 0 || nnz > 0x7fffffff || tile_rows < 1) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  const int64_t ntiles = mgp_cdiv(n, tile_rows);
  MgpArena ar(work, work_bytes);
  const int64_t items = nnz > n + 2 ? nnz : n + 2;
  uint64_t* keys_a = ar.take<uint64_t>(items);
  uint64_t* keys_b = ar.take<uint64_t>(items);
  int32_t* pos_a = ar.take<int32_t>(items + 1);
  int32_t* pos_b = ar.take<int32_t>(items + 1);
  int32_t* head = ar.take<int32_t>(items + 1);
  int32_t* uidx = ar.take<int32_t>(items + 1);
  size_t cub_bytes = cub_bytes_for(items);
  void* cub = ar.take<char>(cub_bytes);
  int32_t* maxima = ar.take<int32_t>(2);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;

  const int32_t* rp_tile = rowptr;          // row pointers in tile order
  if (row_order) {
    // lengths of the rows in tile order -> tile_rowptr (head is free scratch of >= n + 1 ints here)
    hipLaunchKernelGGL(order_lengths, dim3(grid_for(n + 1)), dim3(kBlock), 0, st, rowptr, row_order, n, head);
    MGP_LAUNCH_CHECK();
    size_t tb0 = cub_bytes;
    MGP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, tb0, head, tile_rowptr, (int)(n + 1), st));
    hipLaunchKernelGGL(tile_keys_ordered, dim3(grid_for(n * 16)), dim3(kBlock), 0, st, rowptr, col, row_order, tile_rowptr,
                       n, tile_rows, keys_a, pos_a, emap);
    MGP_LAUNCH_CHECK();
    rp_tile = tile_rowptr;
  } else {
    hipLaunchKernelGGL(tile_keys, dim3(grid_for(n * 16)), dim3(kBlock), 0, st, rowptr, col, n, tile_rows, keys_a, pos_a);
    MGP_LAUNCH_CHECK();
  }
  hipcub::DoubleBuffer<uint64_t> kb(keys_a, keys_b);
  hipcub::DoubleBuffer<int32_t> vb(pos_a, pos_b);
  size_t tb = cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tb, kb, vb, (int)nnz, 0, 32 + bits_for(ntiles + 1), st));
  const uint64_t* keys = kb.Current();
  const int32_t* pos = vb.Current();
  hipLaunchKernelGGL(mark_heads, dim3(grid_for(nnz)), dim3(kBlock), 0, st, keys, nnz, head);
  MGP_LAUNCH_CHECK();
  tb = cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(cub, tb, head, uidx, (int)nnz, st));
  hipLaunchKernelGGL(tile_starts, dim3(grid_for(ntiles + 1)), dim3(kBlock), 0, st, keys, uidx, head, nnz, ntiles,
                     tile_ptr);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(tile_fill, dim3(grid_for(nnz)), dim3(kBlock), 0, st, keys, pos, head, uidx, tile_ptr, nnz,
                     tile_cols, lid);
  MGP_LAUNCH_CHECK();
  MGP_HIP_TRY(hipMemsetAsync(maxima, 0, 2 * sizeof(int32_t), st));
  hipLaunchKernelGGL(tile_extents, dim3(grid_for(ntiles)), dim3(kBlock), 0, st, rp_tile, tile_ptr, n, tile_rows, ntiles,
                     maxima);
  MGP_LAUNCH_CHECK();
  int32_t h[2] = {0, 0}, last = 0;
  MGP_HIP_TRY(hipMemcpyAsync(h, maxima, sizeof(h), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipMemcpyAsync(&last, tile_ptr + ntiles, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  *total_cols = last;
  *max_cols = h[0];
  *max_entries = h[1];
  return h[0] > 65536 ? MGP_ERR_UNSUPPORTED : MGP_OK;
}

// ------------------------------------------------------------------------------------------------
// Locality order for inputs that arrive unordered (random point order: a 64-row tile then references
// ~64 x row-length distinct columns and the dictionary SpMV degenerates to one gather per entry).
// Breadth-first (Cuthill-McKee style) numbering of the k-NN graph: level by level, the nodes of a level
// sorted by (position of their first-numbered parent, node id).  Works for any ambient dimension, needs
// only the CSR.  Deterministic: the parent of a node is the MINIMUM position among its numbered
// neighbours (atomicMin), ties broken by node id in the per-level radix sort.
namespace {

__global__ void bfs_expand(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                           const int32_t* __restrict__ frontier, int fcount, int base,
                           const int32_t* __restrict__ pos, uint32_t* __restrict__ pkey,
                           int32_t* __restrict__ cand, int32_t* __restrict__ ccount) {
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t f = g; f < fcount; f += ng) {
    const int v = frontier[f];
    const uint32_t mypos = (uint32_t)(base + f);
    for (int i = rowptr[v] + lane, e = rowptr[v + 1]; i < e; i += 16) {
      const int c = col[i];
      if (pos[c] >= 0) continue;                       // numbered already (incl. padding entries: c == v)
      const uint32_t old = atomicMin(&pkey[c], mypos);
      if (old == 0xffffffffu) cand[atomicAdd(ccount, 1)] = c;   // first touch: c joins the next level once
    }
  }
}

__global__ void bfs_keys(const int32_t* __restrict__ cand, int m, const uint32_t* __restrict__ pkey,
                         uint64_t* __restrict__ keys) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
    keys[i] = ((uint64_t)pkey[cand[i]] << 32) | (uint32_t)cand[i];
}

__global__ void bfs_number(const uint64_t* __restrict__ keys, int m, int base, int32_t* __restrict__ pos,
                           int32_t* __restrict__ frontier, int32_t* __restrict__ order) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const int v = (int)(keys[i] & 0xffffffffu);
    pos[v] = base + i;
    frontier[i] = v;
    order[base + i] = v;
  }
}

__global__ void bfs_first_unvisited(const int32_t* __restrict__ pos, int64_t n, int from, int32_t* __restrict__ out) {
  for (int64_t i = from + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (pos[i] < 0) { atomicMin(out, (int32_t)i); return; }
}

__global__ void bfs_seed(int32_t v, int base, int32_t* __restrict__ pos, int32_t* __restrict__ frontier,
                         int32_t* __restrict__ order) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { pos[v] = base; frontier[0] = v; order[base] = v; }
}

__global__ void bfs_append_rest(int32_t* __restrict__ pos, int64_t n, int32_t* __restrict__ order, int32_t* __restrict__ counter) {
  // level cap reached (pathological diameter): the remaining nodes are appended behind the numbered ones
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (pos[i] < 0) { const int p = atomicAdd(counter, 1); pos[i] = p; order[p] = (int32_t)i; }
}

}  // namespace

extern "C" size_t mgp_graph_bfs_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  size_t b = 3 * mgp_align((size_t)n * sizeof(int32_t)) + mgp_align((size_t)n * sizeof(uint32_t));
  b += 2 * mgp_align((size_t)n * sizeof(uint64_t)) + cub_bytes_for(n) + mgp_align(64);
  return b + 4096;
}

extern "C" int mgp_graph_bfs_order(int64_t n, const int32_t* rowptr, const int32_t* col, int32_t* order, void* work,
                                   size_t work_bytes, void* stream) {
  if (!rowptr || !col || !order || !work || n <= 0 || n > 0x7fffffff) return MGP_ERR_ARG;
  if (work_bytes < mgp_graph_bfs_workspace_bytes(n)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  int32_t* pos = ar.take<int32_t>(n);
  int32_t* frontier = ar.take<int32_t>(n);
  int32_t* cand = ar.take<int32_t>(n);
  uint32_t* pkey = ar.take<uint32_t>(n);
  uint64_t* keys_a = ar.take<uint64_t>(n);
  uint64_t* keys_b = ar.take<uint64_t>(n);
  const size_t cub_bytes = cub_bytes_for(n);
  void* cub = ar.take<char>(cub_bytes);
  int32_t* counters = ar.take<int32_t>(16);      // [0] candidates of the level, [1] first unvisited
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  MGP_HIP_TRY(hipMemsetAsync(pos, 0xff, (size_t)n * sizeof(int32_t), st));
  MGP_HIP_TRY(hipMemsetAsync(pkey, 0xff, (size_t)n * sizeof(uint32_t), st));
  int numbered = 0, fcount = 0, fbase = 0, scan_from = 0;
  int64_t levels = 0;
  const int64_t level_cap = 20000;   // one host round trip per level: bounds the setup cost on chain-like graphs (<~ 1 s)
  while (numbered < n) {
    if (fcount == 0) {                               // next component: seed = smallest unnumbered node
      int32_t big = 0x7fffffff;
      MGP_HIP_TRY(hipMemcpyAsync(counters + 1, &big, sizeof(int32_t), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(bfs_first_unvisited, dim3(grid_for(n - scan_from)), dim3(kBlock), 0, st, pos, n, scan_from, counters + 1);
      MGP_LAUNCH_CHECK();
      int32_t seed = 0;
      MGP_HIP_TRY(hipMemcpyAsync(&seed, counters + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      if (seed >= n) break;
      hipLaunchKernelGGL(bfs_seed, dim3(1), dim3(64), 0, st, seed, numbered, pos, frontier, order);
      MGP_LAUNCH_CHECK();
      scan_from = seed + 1;
      fbase = numbered;
      fcount = 1;
      numbered += 1;
      continue;
    }
    if (++levels > level_cap) break;
    MGP_HIP_TRY(hipMemsetAsync(counters, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(bfs_expand, dim3(grid_for((int64_t)fcount * 16)), dim3(kBlock), 0, st, rowptr, col, frontier, fcount,
                       fbase, pos, pkey, cand, counters);
    MGP_LAUNCH_CHECK();
    int32_t m = 0;
    MGP_HIP_TRY(hipMemcpyAsync(&m, counters, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    if (m > 0) {
      hipLaunchKernelGGL(bfs_keys, dim3(grid_for(m)), dim3(kBlock), 0, st, cand, m, pkey, keys_a);
      MGP_LAUNCH_CHECK();
      hipcub::DoubleBuffer<uint64_t> kb(keys_a, keys_b);
      size_t tb = cub_bytes;
      MGP_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(cub, tb, kb, m, 0, 64, st));
      hipLaunchKernelGGL(bfs_number, dim3(grid_for(m)), dim3(kBlock), 0, st, kb.Current(), m, numbered, pos, frontier, order);
      MGP_LAUNCH_CHECK();
    }
    fbase = numbered;
    fcount = m;
    numbered += m;
  }
  if (numbered < n) {
    MGP_HIP_TRY(hipMemcpyAsync(counters, &numbered, sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(bfs_append_rest, dim3(grid_for(n)), dim3(kBlock), 0, st, pos, n, order, counters);
    MGP_LAUNCH_CHECK();
  }
  MGP_HIP_TRY(hipStreamSynchronize(st));
  return MGP_OK;
}

// ---- row permutation of an [n, C] block: dst[p, :] = src[order[p], :].  The iterative solvers and the eigensolver run on
// P A P^T (graph.RelabelledGraph): right-hand sides are permuted in and solutions out around every solve.  torch.index_select
// takes 12 us for a 60 000 x 12 block (48-byte rows through a generic gather); here a lane moves a float4 of a row (C % 4 == 0)
// or one element, destination coalesced.
namespace {

__global__ __launch_bounds__(kBlock) void permute_rows_q_kernel(const float4* __restrict__ src, const int32_t* __restrict__ order,
                                                                 int64_t n, int CQ, float4* __restrict__ dst) {
  const int64_t total = n * CQ;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t p = e / CQ;
    const int q = (int)(e - p * CQ);
    dst[e] = src[(int64_t)order[p] * CQ + q];
  }
}

__global__ __launch_bounds__(kBlock) void permute_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ order,
                                                               int64_t n, int C, float* __restrict__ dst) {
  const int64_t total = n * C;
  for (int64_t e = blockIdx.x * (int64_t)kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t p = e / C;
    const int c = (int)(e - p * C);
    dst[e] = src[(int64_t)order[p] * C + c];
  }
}

}  // namespace

extern "C" int mgp_permute_rows(const float* src, const int32_t* order, int64_t n, int C, float* dst, void* stream) {
  if (!src || !order || !dst || n < 0 || C <= 0 || src == dst) return MGP_ERR_ARG;
  if (n == 0) return MGP_OK;
  hipStream_t st = mgp_stream(stream);
  const bool quads = (C & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
  const int64_t total = quads ? n * (C >> 2) : n * (int64_t)C;
  const unsigned grid = (unsigned)std::min<int64_t>(mgp_cdiv(total, kBlock), 8192);
  if (quads)
    hipLaunchKernelGGL(permute_rows_q_kernel, dim3(grid), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(src), order, n, C >> 2,
                       reinterpret_cast<float4*>(dst));
  else
    hipLaunchKernelGGL(permute_rows_kernel, dim3(grid), dim3(kBlock), 0, st, src, order, n, C, dst);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}


// ---------------------------------------------------------------- nearest-neighbour chain order (round 5)
// A locality order for k-NN graphs whose node order carries the CLUSTERS but not the order inside them (the RMNIST-like set: an
// orbit's 100 rotations arrive in random angle order): walk the graph, always stepping to the nearest not-yet-numbered neighbour
// of the node in hand (by the edge's squared distance); when it has none, to the nearest such neighbour of one of the last 64
// nodes of the chain; else to the unnumbered node of smallest index.  Consecutive positions are then near neighbours -- on a
// rotation orbit the chain runs along the angle -- and the rows of a tile share most of their columns: 16-row tiles of the 60k
// RMNIST-like graph name 291 distinct columns in the given order, 233 in breadth-first order, ~115 in this one, which is what
// the matrix-core SpMM's work and gathers are proportional to (docs/kernels/spmm.md, round 5).  No reference counterpart.
// The walk is sequential: it runs on the HOST, once per graph.  Round 5 (second half): the device first orders every row's
// neighbours by (d2, column) (chain_rank_kernel: one wave per row, every entry counts the entries ahead of it), so that the host
// receives ONE int32 array through a pinned staging buffer (60k nodes, k = 50: 15 MB instead of 30 MB into pageable vectors) and
// "the nearest unvisited neighbour" is the first unvisited entry of the row instead of a scan of all of it: 19.5 -> ~7 ms, the
// same order entry for entry (same tie rule: smaller column).  Deterministic.  Synchronises `stream`.
namespace {
__global__ __launch_bounds__(256) void chain_rank_kernel(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         const float* __restrict__ d2, int32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= n) return;
  const int32_t b = rowptr[v], e = rowptr[v + 1];
  for (int32_t i = b + lane; i < e; i += 64) {
    const float di = d2[i];
    const int32_t ci = col[i];
    int32_t ahead = 0;
    for (int32_t j = b; j < e; ++j) {
      const float dj = d2[j];
      const int32_t cj = col[j];
      ahead += (dj < di || (dj == di && (cj < ci || (cj == ci && j < i)))) ? 1 : 0;
    }
    out[b + ahead] = ci;
  }
}

// pinned staging for the sorted columns (kept for the life of the process; grown when a larger graph comes)
struct ChainStage {
  int32_t* p = nullptr;
  size_t cap = 0;
  std::mutex mu;
};
ChainStage g_chain_stage;
}  // namespace

extern "C" int mgp_graph_chain_order(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2, int32_t* order,
                                     void* stream) {
  if (!rowptr || !col || !d2 || !order || n <= 0 || n > 0x7fffffff) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  std::vector<int32_t> rp((size_t)n + 1);
  MGP_HIP_TRY(hipMemcpyAsync(rp.data(), rowptr, ((size_t)n + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  const size_t nnz = (size_t)rp[n];
  std::lock_guard<std::mutex> lock(g_chain_stage.mu);
  if (g_chain_stage.cap < nnz + 1) {
    if (g_chain_stage.p) (void)hipHostFree(g_chain_stage.p);
    g_chain_stage.p = nullptr;
    g_chain_stage.cap = 0;
    MGP_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&g_chain_stage.p), (nnz + 1) * sizeof(int32_t), hipHostMallocDefault));
    g_chain_stage.cap = nnz + 1;
  }
  const int32_t* cj = g_chain_stage.p;
  if (nnz) {
    int32_t* sorted = nullptr;
    MGP_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sorted), nnz * sizeof(int32_t)));
    hipLaunchKernelGGL(chain_rank_kernel, dim3((unsigned)mgp_cdiv(n, 4)), dim3(256), 0, st, n, rowptr, col, d2, sorted);
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipMemcpyAsync(g_chain_stage.p, sorted, nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (le == hipSuccess) le = hipStreamSynchronize(st);
    (void)hipFree(sorted);
    if (le != hipSuccess) return (int)le;
  }
  std::vector<uint8_t> seen((size_t)n, 0);
  std::vector<int32_t> ord((size_t)n);
  auto nearest_free = [&](int32_t v) -> int32_t {      // the row is in ascending (d2, column) order: the first unvisited entry
    for (int32_t e = rp[v]; e < rp[v + 1]; ++e) {
      const int32_t c = cj[e];
      if (c < 0 || c >= n || c == v || seen[c]) continue;        // (padding entries name the row itself)
      return c;
    }
    return -1;
  };
  int64_t scan = 0, cnt = 0;
  int32_t cur = 0;
  while (cnt < n) {
    ord[cnt++] = cur;
    seen[cur] = 1;
    if (cnt == n) break;
    int32_t nxt = nearest_free(cur);
    for (int64_t back = cnt - 2; nxt < 0 && back >= 0 && back >= cnt - 64; --back) nxt = nearest_free(ord[back]);
    if (nxt < 0) {
      while (scan < n && seen[scan]) ++scan;
      nxt = (int32_t)scan;
    }
    cur = nxt;
  }
  MGP_HIP_TRY(hipMemcpyAsync(order, ord.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  return MGP_OK;
}
