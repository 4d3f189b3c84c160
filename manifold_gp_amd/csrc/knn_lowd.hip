// Exact k-NN for low-dimensional points (d <= 3): no N x n distance slab.
//
// The slab pipeline of knn.hip writes and re-reads 4 N n bytes of fp32 distances; at d = 3 that traffic
// (20 TB at N = n = 1M) is ~100x the arithmetic.  For d <= 3 a cheap, PROVABLE upper bound on every
// query's k-th neighbour distance is available from a space-filling-curve order, and with it one fused
// streaming pass keeps only a few hundred candidates per query:
//
//   0. Morton-sort the indexed points (21 bits per axis, hipcub radix sort), keep the permutation;
//   A. per query: locate its Morton code in the sorted codes (binary search) and take the k-th smallest
//      fp32 distance T32 among the W points around that position (W = pow2 >= 4k, LDS bitonic sort).
//      The window is a SUBSET of the points, so its k-th smallest fp64 distance bounds the true k-th from
//      above; with the fp32 error bound gamma every true neighbour p satisfies
//          d32(p) <= T32 (1 + gamma) / (1 - gamma)  <=  T' := T32 (1 + 8 gamma)
//   B. fused filter: a workgroup owns 256 queries (one per lane), all points stream through LDS in
//      1024-point chunks (broadcast ds_read), every point with d32 <= T' is appended to the lane's own
//      candidate list (private counter, no atomics) -- ~8 VALU instructions per pair, no HBM traffic
//      beyond the 16 N bytes of points per workgroup;
//   C. per query: fp64 distances of the candidates in the oracle's operation order, bitonic sort by
//      (d64, ORIGINAL index), first k written.  All points with d64 <= the true k-th distance are among
//      the candidates, so this is the oracle's answer bit for bit, ties included.
//   Queries whose list overflows (`kCap` slots; Morton discontinuities, heavy duplicates) are redone by
//   the slab pipeline (mgp_knn_bruteforce) -- rare by construction, counted in stats[0].
#include <hipcub/hipcub.hpp>
#include <limits.h>
#include <math.h>
#include <vector>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kCap = 512;      // candidate slots per query
constexpr int kChunk = 1024;   // points staged in LDS per step of the filter
constexpr int kMaxW = 1024;    // window of phase A (W = pow2 >= 4k)

inline int grid_for(int64_t n, int per = kBlock) {
  int64_t g = mgp_cdiv(n, per);
  return (int)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

struct Box {
  float lo[3];
  float inv[3];   // 2^21 / extent (0 for unused axes)
};

__device__ __forceinline__ uint64_t spread3(uint32_t v) {   // 21 bits -> every third bit
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

__device__ __forceinline__ uint64_t morton_code(const float* __restrict__ p, int d, const Box& b) {
  uint64_t code = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (a < d) {
      float t = (p[a] - b.lo[a]) * b.inv[a];
      t = t < 0.f ? 0.f : (t > 2097151.f ? 2097151.f : t);
      code |= spread3((uint32_t)t) << a;
    }
  }
  return code;
}

__global__ void minmax_kernel(const float* __restrict__ x, int64_t n, int d, float* __restrict__ part /*[grid][6]*/) {
  __shared__ float sh[6][kBlock];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    for (int a = 0; a < d; ++a) {
      const float v = x[i * d + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  for (int a = 0; a < 3; ++a) { sh[a][threadIdx.x] = lo[a]; sh[3 + a][threadIdx.x] = hi[a]; }
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int a = 0; a < 3; ++a) {
        sh[a][threadIdx.x] = fminf(sh[a][threadIdx.x], sh[a][threadIdx.x + s]);
        sh[3 + a][threadIdx.x] = fmaxf(sh[3 + a][threadIdx.x], sh[3 + a][threadIdx.x + s]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void codes_kernel(const float* __restrict__ x, int64_t n, int d, Box b, uint64_t* __restrict__ codes,
                             int32_t* __restrict__ idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    codes[i] = morton_code(x + i * d, d, b);
    if (idx) idx[i] = (int32_t)i;
  }
}

// sorted points as float4 {x, y, z (0 for missing axes), unused}: one 16-byte LDS broadcast per point
__global__ void gather_points_kernel(const float* __restrict__ x, const int32_t* __restrict__ perm, int64_t n, int d,
                                     float4* __restrict__ xs) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = x + (int64_t)perm[i] * d;
    xs[i] = make_float4(p[0], d > 1 ? p[1] : 0.f, d > 2 ? p[2] : 0.f, 0.f);
  }
}

__global__ void query_pos_kernel(const float* __restrict__ q, int64_t n, int d, Box b,
                                 const uint64_t* __restrict__ sorted_codes, int64_t N, int32_t* __restrict__ pos) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t c = morton_code(q + i * d, d, b);
    int64_t lo = 0, hi = N;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_codes[mid] < c) lo = mid + 1; else hi = mid;
    }
    pos[i] = (int32_t)(lo < N ? lo : N - 1);
  }
}

__device__ __forceinline__ float d2_f32(float qx, float qy, float qz, const float4& p) {
  const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
  return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---- phase A: one wave per query, W window distances sorted in LDS, threshold = the k-th smallest
template <int W>
__global__ __launch_bounds__(kBlock) void window_threshold_kernel(const float4* __restrict__ xs, int64_t N,
                                                                  const float* __restrict__ q, int64_t n, int d,
                                                                  const int32_t* __restrict__ pos, int k, float inflate,
                                                                  float* __restrict__ T) {
  __shared__ float arr[kBlock / 64][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t qi_raw = (int64_t)blockIdx.x * (kBlock / 64) + wave;
  const int64_t qi = qi_raw < n ? qi_raw : n - 1;
  const float qx = q[qi * d], qy = d > 1 ? q[qi * d + 1] : 0.f, qz = d > 2 ? q[qi * d + 2] : 0.f;
  int64_t w0 = (int64_t)pos[qi] - W / 2;
  if (w0 > N - W) w0 = N - W;
  if (w0 < 0) w0 = 0;
  float* a = arr[wave];
#pragma unroll
  for (int t = 0; t < W / 64; ++t) {
    const int64_t j = w0 + lane + 64 * t;
    a[lane + 64 * t] = j < N ? d2_f32(qx, qy, qz, xs[j]) : INFINITY;
  }
  __syncthreads();
  for (int size = 2; size <= W; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int t = 0; t < W / 128; ++t) {
        const int e = lane + 64 * t;
        const int lo = (e / stride) * stride * 2 + (e % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const float vl = a[lo], vh = a[hi];
        if (up ? (vh < vl) : (vl < vh)) { a[lo] = vh; a[hi] = vl; }
      }
      __syncthreads();
    }
  }
  if (lane == 0 && qi_raw < n) T[qi] = a[k - 1] * inflate;
}

// ---- retry of overflowing queries: the kCap candidates a query DID store are real points under its old bound,
// so the k-th smallest distance among them is a valid -- and much tighter -- bound (the overflowing balls come
// from windows that straddle a jump of the curve).  One wave per listed query; resets its candidate counter.
__global__ __launch_bounds__(kBlock) void tighten_kernel(const float4* __restrict__ xs, const float* __restrict__ q, int d,
                                                         const int32_t* __restrict__ qlist, int nlist,
                                                         const int32_t* __restrict__ cand, int k, float inflate,
                                                         float* __restrict__ T, int32_t* __restrict__ cnt) {
  __shared__ float arr[kBlock / 64][kCap];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = blockIdx.x * (kBlock / 64) + wave;
  const int64_t qi = qlist[w < nlist ? w : nlist - 1];
  const float qx = q[qi * d], qy = d > 1 ? q[qi * d + 1] : 0.f, qz = d > 2 ? q[qi * d + 2] : 0.f;
  float* a = arr[wave];
#pragma unroll
  for (int t = 0; t < kCap / 64; ++t) a[lane + 64 * t] = d2_f32(qx, qy, qz, xs[cand[qi * kCap + lane + 64 * t]]);
  __syncthreads();
  for (int size = 2; size <= kCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int t = 0; t < kCap / 128; ++t) {
        const int e = lane + 64 * t;
        const int lo = (e / stride) * stride * 2 + (e % stride);
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const float vl = a[lo], vh = a[hi];
        if (up ? (vh < vl) : (vl < vh)) { a[lo] = vh; a[hi] = vl; }
      }
      __syncthreads();
    }
  }
  if (lane == 0 && w < nlist) {
    T[qi] = a[k - 1] * inflate;
    cnt[qi] = 0;
  }
}

// ---- phase B: fused distance + filter, one query per lane, points broadcast from LDS.
// gridDim.y > 1 (few queries against many points, e.g. out-of-sample features of a small test batch):
// the point range is split over blockIdx.y so that the pass still fills the chip; the lanes of the
// segments then share a query's candidate list through an atomic slot counter (the list order is
// irrelevant, phase C sorts by (d64, index)).  gridDim.y == 1: private counters, no atomics.
// bounding box of every chunk of kChunk Morton-consecutive points: [lo.xyz, hi.xyz]
__global__ __launch_bounds__(kBlock) void chunk_box_kernel(const float4* __restrict__ xs, int64_t N, float* __restrict__ boxes) {
  __shared__ float sh[kBlock / 64][6];
  const int64_t c0 = (int64_t)blockIdx.x * kChunk;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int t = threadIdx.x; t < kChunk; t += kBlock) {
    const int64_t j = c0 + t;
    if (j < N) {
      const float4 p = xs[j];
      lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
      hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if ((threadIdx.x & 63) == 0)
    for (int a = 0; a < 3; ++a) { sh[threadIdx.x >> 6][a] = lo[a]; sh[threadIdx.x >> 6][3 + a] = hi[a]; }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = sh[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fminf(v, sh[w][threadIdx.x]) : fmaxf(v, sh[w][threadIdx.x]);
    boxes[(int64_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ void iota_kernel(int32_t* __restrict__ v, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) v[i] = (int32_t)i;
}

// ---- phase B: one query per lane, the 256 queries of a workgroup CONSECUTIVE ON THE CURVE (qorder), so that
// their search balls -- centre q, radius sqrt(threshold) -- fill a small box.  A chunk of 1024 Morton-consecutive
// points whose bounding box misses that box (in any axis) holds no point under any of the 256 thresholds and is
// skipped without being staged: 256 chunk boxes are tested per pass (one per lane), the survivors compacted in
// order and streamed through LDS as before.  The candidate lists are exactly those of the unpruned sweep (the
// test is conservative: the box is widened by 1e-6 relative + the fp32 spacing of the coordinates);
// at N = 1M on a 2-D surface ~2 % of the chunks survive (113 -> 4 ms).
__global__ __launch_bounds__(kBlock) void filter_kernel(const float4* __restrict__ xs, int64_t N,
                                                        const float* __restrict__ q, int64_t n, int d,
                                                        const float* __restrict__ T, int32_t* __restrict__ cand,
                                                        int32_t* __restrict__ cnt, int64_t seg,
                                                        const int32_t* __restrict__ qorder, const float* __restrict__ boxes) {
  __shared__ float4 pts[kChunk];
  __shared__ float sh_box[kBlock / 64][6];
  __shared__ int sh_wcnt[kBlock / 64];
  __shared__ int sh_list[kBlock];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t slot_raw = (int64_t)blockIdx.x * kBlock + tid;
  const bool live = slot_raw < n;
  const int64_t qi = qorder[live ? slot_raw : n - 1];
  const float qx = q[qi * d], qy = d > 1 ? q[qi * d + 1] : 0.f, qz = d > 2 ? q[qi * d + 2] : 0.f;
  const float thr = live ? T[qi] : -1.f;
  int32_t* __restrict__ mine = cand + qi * kCap;
  const bool shared_list = gridDim.y > 1;
  const int64_t p_begin = (int64_t)blockIdx.y * seg;
  const int64_t p_end = p_begin + seg < N ? p_begin + seg : N;
  // the workgroup's search box
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (live) {
    const float qa[3] = {qx, qy, qz};
    const float rad = sqrtf(fmaxf(thr, 0.f));
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float r = rad * (1.f + 1e-6f) + 4e-7f * (fabsf(qa[a]) + rad) + 1e-37f;
      lo[a] = qa[a] - r; hi[a] = qa[a] + r;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if (lane == 0)
    for (int a = 0; a < 3; ++a) { sh_box[wave][a] = lo[a]; sh_box[wave][3 + a] = hi[a]; }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = fminf(fminf(sh_box[0][a], sh_box[1][a]), fminf(sh_box[2][a], sh_box[3][a]));
    hi[a] = fmaxf(fmaxf(sh_box[0][3 + a], sh_box[1][3 + a]), fmaxf(sh_box[2][3 + a], sh_box[3][3 + a]));
  }
  int c = 0;
  const int64_t ch_begin = p_begin / kChunk, ch_end = (p_end + kChunk - 1) / kChunk;
  for (int64_t cb = ch_begin; cb < ch_end; cb += kBlock) {
    // one chunk box per lane; survivors compacted in ascending order
    const int64_t ch = cb + tid;
    bool ov = false;
    if (ch < ch_end) {
      const float* bx = boxes + ch * 6;
      ov = bx[0] <= hi[0] && bx[3] >= lo[0] && bx[1] <= hi[1] && bx[4] >= lo[1] && bx[2] <= hi[2] && bx[5] >= lo[2];
    }
    const unsigned long long m = __ballot(ov);
    if (lane == 0) sh_wcnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { if (w < wave) base += sh_wcnt[w]; total += sh_wcnt[w]; }
    if (ov) sh_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)(ch - cb);
    __syncthreads();
    for (int s = 0; s < total; ++s) {
      const int64_t c0 = (cb + sh_list[s]) * kChunk;
      __syncthreads();
#pragma unroll
      for (int t = 0; t < kChunk / kBlock; ++t) {
        const int64_t j = c0 + tid + t * kBlock;
        // points past the end sit infinitely far away
        pts[tid + t * kBlock] = j < p_end ? xs[j] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
      }
      __syncthreads();
#pragma unroll 8
      for (int p = 0; p < kChunk; ++p) {
        const float dd = d2_f32(qx, qy, qz, pts[p]);
        if (dd <= thr) {
          const int slot = shared_list ? atomicAdd(&cnt[qi], 1) : c;
          if (slot < kCap) mine[slot] = (int32_t)(c0 + p);
          ++c;
        }
      }
    }
    __syncthreads();                                   // sh_list / sh_wcnt are rewritten by the next pass
  }
  if (live && !shared_list) cnt[qi] = c;
}

__device__ __forceinline__ bool pair_less(double d1, int i1, double d2, int i2) {
  return (d1 < d2) || (d1 == d2 && i1 < i2);
}

// ---- phase C: one wave per query: exact fp64 distances (oracle order), sort by (d64, original index)
__global__ __launch_bounds__(kBlock) void rerank_kernel(const float4* __restrict__ xs, const int32_t* __restrict__ perm,
                                                        const float* __restrict__ q, int64_t n, int d, int k,
                                                        const int32_t* __restrict__ cand, const int32_t* __restrict__ cnt,
                                                        float* __restrict__ D, int32_t* __restrict__ I,
                                                        int32_t* __restrict__ over_list, int32_t* __restrict__ over_count,
                                                        const int32_t* __restrict__ qlist) {
  __shared__ double sd[kBlock / 64][kCap];
  __shared__ int si[kBlock / 64][kCap];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // qlist == nullptr: queries 0 .. n-1; else the n listed queries
  const int64_t qi_raw = (int64_t)blockIdx.x * (kBlock / 64) + wave;
  const bool live = qi_raw < n;
  const int64_t qsel = live ? qi_raw : n - 1;
  const int64_t qi = qlist ? (int64_t)qlist[qsel] : qsel;
  const int c = cnt[qi];
  const bool over = c > kCap;
  const int m = over ? 0 : c;
  int P = 64;
  while (P < m) P <<= 1;                       // pow2 >= m, wave-uniform
  double* dd = sd[wave];
  int* ii = si[wave];
  const double qd[3] = {(double)q[qi * d], d > 1 ? (double)q[qi * d + 1] : 0.0, d > 2 ? (double)q[qi * d + 2] : 0.0};
  for (int e = lane; e < P; e += 64) {
    double dist = INFINITY;
    int orig = INT_MAX;
    if (e < m) {
      const int j = cand[qi * kCap + e];
      const float4 p = xs[j];
      const float pv[3] = {p.x, p.y, p.z};
      double acc = 0.0;
      for (int a = 0; a < d; ++a) {              // oracle/knn_oracle.c: ascending features, no contraction
        const double df = __dsub_rn(qd[a], (double)pv[a]);
        acc = __dadd_rn(acc, __dmul_rn(df, df));
      }
      dist = acc;
      orig = perm[j];
    }
    dd[e] = dist;
    ii[e] = orig;
  }
  __syncthreads();
  // kCap / 2 compare-exchanges per step at most; every wave runs the same (maximal) number of barriers
  for (int size = 2; size <= kCap; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (size <= P) {
        for (int e = lane; e < P / 2; e += 64) {
          const int lo = (e / stride) * stride * 2 + (e % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const double dl = dd[lo], dh = dd[hi];
          const int il = ii[lo], ih = ii[hi];
          const bool swap = up ? pair_less(dh, ih, dl, il) : pair_less(dl, il, dh, ih);
          if (swap) { dd[lo] = dh; dd[hi] = dl; ii[lo] = ih; ii[hi] = il; }
        }
      }
      __syncthreads();
    }
  }
  if (!live) return;
  if (over) {
    if (lane == 0) over_list[atomicAdd(over_count, 1)] = (int32_t)qi;
    return;
  }
  for (int t = lane; t < k; t += 64) {
    D[qi * k + t] = (float)dd[t];
    I[qi * k + t] = ii[t];
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows, int64_t m, int w,
                                   float* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m * w; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[(int64_t)rows[i / w] * w + (i % w)];
}

__global__ void scatter_rows_kernel(const float* __restrict__ Ds, const int32_t* __restrict__ Is,
                                    const int32_t* __restrict__ rows, int64_t m, int k, float* __restrict__ D,
                                    int32_t* __restrict__ I) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < m * k; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[i / k];
    D[r * k + (i % k)] = Ds[i];
    I[r * k + (i % k)] = Is[i];
  }
}

int window_for(int k) {
  int w = 256;
  while (w < 4 * k) w <<= 1;
  return w;
}

size_t cub_sort_bytes32(int64_t items) {
  size_t a = 0;
  hipcub::DoubleBuffer<int32_t> kb(nullptr, nullptr);
  hipcub::DoubleBuffer<int32_t> vb(nullptr, nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, kb, vb, (int)items, 0, 31, (hipStream_t)0);
  return mgp_align(a + 1024);
}

size_t cub_sort_bytes(int64_t items) {
  size_t a = 0;
  hipcub::DoubleBuffer<uint64_t> kb(nullptr, nullptr);
  hipcub::DoubleBuffer<int32_t> vb(nullptr, nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, kb, vb, (int)items, 0, 63, (hipStream_t)0);
  return mgp_align(a + 1024);
}

}  // namespace

int mgp_knn_lowd_eligible(int64_t N, int64_t n, int d, int k) {
  return d >= 1 && d <= 3 && N >= 4096 && k <= N && window_for(k) <= kMaxW && window_for(k) <= N && n >= 1;
}

size_t mgp_knn_lowd_workspace_bytes(int64_t N, int64_t n, int d, int k) {
  if (!mgp_knn_lowd_eligible(N, n, d, k)) return 0;
  size_t b = 0;
  b += 2 * mgp_align((size_t)N * sizeof(uint64_t));      // codes (double buffer)
  b += 2 * mgp_align((size_t)N * sizeof(int32_t));       // permutation (double buffer)
  b += cub_sort_bytes(N);
  b += mgp_align((size_t)N * sizeof(float4));            // sorted points
  b += mgp_align((size_t)1024 * 6 * sizeof(float));      // bounding-box partials
  b += 3 * mgp_align((size_t)n * sizeof(int32_t));       // pos, cnt, overflow list
  b += mgp_align((size_t)n * sizeof(float));             // thresholds
  b += mgp_align((size_t)n * kCap * sizeof(int32_t));    // candidate lists
  b += mgp_align(64);
  b += mgp_align((size_t)mgp_cdiv(N, kChunk) * 6 * sizeof(float));          // chunk boxes
  b += 4 * mgp_align((size_t)n * sizeof(int32_t)) + cub_sort_bytes32(n);  // queries in curve order
  return b + 4096;
}

// D / I rows of the queries; returns the number of overflowed queries through *n_over (their rows are
// listed in over_rows_host order is irrelevant: the caller redoes them with the slab pipeline)
int mgp_knn_lowd(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D, int32_t* I, void* work,
                 size_t work_bytes, std::vector<int32_t>* over_rows, void* stream) {
  if (!mgp_knn_lowd_eligible(N, n, d, k)) return MGP_ERR_UNSUPPORTED;
  if (work_bytes < mgp_knn_lowd_workspace_bytes(N, n, d, k)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  uint64_t* codes_a = ar.take<uint64_t>(N);
  uint64_t* codes_b = ar.take<uint64_t>(N);
  int32_t* perm_a = ar.take<int32_t>(N);
  int32_t* perm_b = ar.take<int32_t>(N);
  const size_t cub_bytes = cub_sort_bytes(N);
  void* cub = ar.take<char>(cub_bytes);
  float4* xs = ar.take<float4>(N);
  float* boxpart = ar.take<float>(1024 * 6);
  int32_t* pos = ar.take<int32_t>(n);
  int32_t* cnt = ar.take<int32_t>(n);
  int32_t* over_list = ar.take<int32_t>(n);
  float* T = ar.take<float>(n);
  int32_t* cand = ar.take<int32_t>((size_t)n * kCap);
  int32_t* over_count = ar.take<int32_t>(16);
  float* chunk_boxes = ar.take<float>((size_t)mgp_cdiv(N, kChunk) * 6);
  int32_t* qk_a = ar.take<int32_t>(n);
  int32_t* qk_b = ar.take<int32_t>(n);
  int32_t* qv_a = ar.take<int32_t>(n);
  int32_t* qv_b = ar.take<int32_t>(n);
  const size_t cubq_bytes = cub_sort_bytes32(n);
  void* cubq = ar.take<char>(cubq_bytes);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;

  // 0. bounding box (of the indexed points; queries outside are clamped onto it), Morton sort
  const int bgrid = (int)std::min<int64_t>(1024, mgp_cdiv(N, kBlock));
  hipLaunchKernelGGL(minmax_kernel, dim3(bgrid), dim3(kBlock), 0, st, db, N, d, boxpart);
  MGP_LAUNCH_CHECK();
  std::vector<float> hp((size_t)bgrid * 6);
  MGP_HIP_TRY(hipMemcpyAsync(hp.data(), boxpart, hp.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  Box box;
  for (int a = 0; a < 3; ++a) {
    float lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < bgrid; ++b) { lo = std::min(lo, hp[(size_t)b * 6 + a]); hi = std::max(hi, hp[(size_t)b * 6 + 3 + a]); }
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) { box.lo[a] = std::isfinite(lo) ? lo : 0.f; box.inv[a] = 0.f; }
    else { box.lo[a] = lo; box.inv[a] = 2097151.0f / (hi - lo); }
  }
  hipLaunchKernelGGL(codes_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, db, N, d, box, codes_a, perm_a);
  MGP_LAUNCH_CHECK();
  hipcub::DoubleBuffer<uint64_t> kb(codes_a, codes_b);
  hipcub::DoubleBuffer<int32_t> vb(perm_a, perm_b);
  size_t tb = cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tb, kb, vb, (int)N, 0, 63, st));
  const uint64_t* codes = kb.Current();
  const int32_t* perm = vb.Current();
  hipLaunchKernelGGL(gather_points_kernel, dim3(grid_for(N)), dim3(kBlock), 0, st, db, perm, N, d, xs);
  MGP_LAUNCH_CHECK();

  // A. thresholds
  hipLaunchKernelGGL(query_pos_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, q, n, d, box, codes, N, pos);
  MGP_LAUNCH_CHECK();
  const double gamma = (double)(d + 4) * 1.1920928955078125e-07;     // fp32 relative error of d2_f32
  const float inflate = (float)(1.0 + 8.0 * gamma);
  const int W = window_for(k);
  const int qgrid = (int)mgp_cdiv(n, kBlock / 64);
  if (W == 256) hipLaunchKernelGGL(window_threshold_kernel<256>, dim3(qgrid), dim3(kBlock), 0, st, xs, N, q, n, d, pos, k, inflate, T);
  else if (W == 512) hipLaunchKernelGGL(window_threshold_kernel<512>, dim3(qgrid), dim3(kBlock), 0, st, xs, N, q, n, d, pos, k, inflate, T);
  else hipLaunchKernelGGL(window_threshold_kernel<1024>, dim3(qgrid), dim3(kBlock), 0, st, xs, N, q, n, d, pos, k, inflate, T);
  MGP_LAUNCH_CHECK();

  // queries in curve order (their position among the sorted points), chunk bounding boxes
  MGP_HIP_TRY(hipMemcpyAsync(qk_a, pos, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, qv_a, n);
  MGP_LAUNCH_CHECK();
  hipcub::DoubleBuffer<int32_t> qkb(qk_a, qk_b);
  hipcub::DoubleBuffer<int32_t> qvb(qv_a, qv_b);
  size_t tq = cubq_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cubq, tq, qkb, qvb, (int)n, 0, 31, st));
  const int32_t* qorder = qvb.Current();
  hipLaunchKernelGGL(chunk_box_kernel, dim3((unsigned)mgp_cdiv(N, kChunk)), dim3(kBlock), 0, st, xs, N, chunk_boxes);
  MGP_LAUNCH_CHECK();

  // B. fused filter, C. exact re-rank
  // enough workgroups to fill the chip: split the point range when there are few query blocks
  const int64_t qblocks = mgp_cdiv(n, kBlock);
  int64_t segs = 1;
  if (qblocks < 1024) segs = std::min<int64_t>(mgp_cdiv(1024, qblocks), std::max<int64_t>(1, N / (8 * kChunk)));
  if (segs > 1) MGP_HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)n * sizeof(int32_t), st));
  const int64_t seg_len = mgp_cdiv(mgp_cdiv(N, segs), (int64_t)kChunk) * kChunk;     // whole LDS chunks per segment
  segs = mgp_cdiv(N, seg_len);
  hipLaunchKernelGGL(filter_kernel, dim3((unsigned)qblocks, (unsigned)segs), dim3(kBlock), 0, st, xs, N, q, n, d, T, cand, cnt,
                     seg_len, qorder, chunk_boxes);
  MGP_LAUNCH_CHECK();
  MGP_HIP_TRY(hipMemsetAsync(over_count, 0, 2 * sizeof(int32_t), st));
  hipLaunchKernelGGL(rerank_kernel, dim3(qgrid), dim3(kBlock), 0, st, xs, perm, q, n, d, k, cand, cnt, D, I, over_list,
                     over_count, (const int32_t*)nullptr);
  MGP_LAUNCH_CHECK();
  int32_t nover = 0;
  MGP_HIP_TRY(hipMemcpyAsync(&nover, over_count, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  const int32_t* final_list = over_list;
  if (nover > 0) {
    // retry with the bound tightened from the candidates those queries did store (0.3 % of the rows of a randomly
    // ordered 1M swiss roll overflow; through the slab pipeline they cost 14 ms of a 58 ms search)
    int32_t* over_list2 = pos;                       // the curve positions are dead once the queries are ordered
    hipLaunchKernelGGL(tighten_kernel, dim3((unsigned)mgp_cdiv(nover, kBlock / 64)), dim3(kBlock), 0, st, xs, q, d, over_list,
                       nover, cand, k, inflate, T, cnt);
    MGP_LAUNCH_CHECK();
    const int64_t qb2 = mgp_cdiv(nover, kBlock);
    int64_t segs2 = std::min<int64_t>(mgp_cdiv(1024, qb2), std::max<int64_t>(1, N / (8 * kChunk)));
    const int64_t seg_len2 = mgp_cdiv(mgp_cdiv(N, segs2), (int64_t)kChunk) * kChunk;
    segs2 = mgp_cdiv(N, seg_len2);
    if (segs2 < 2) segs2 = 2;                         // shared-list mode: the counters were reset, not the lists' owners
    hipLaunchKernelGGL(filter_kernel, dim3((unsigned)qb2, (unsigned)segs2), dim3(kBlock), 0, st, xs, N, q, (int64_t)nover, d, T,
                       cand, cnt, seg_len2, over_list, chunk_boxes);
    MGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(rerank_kernel, dim3((unsigned)mgp_cdiv(nover, kBlock / 64)), dim3(kBlock), 0, st, xs, perm, q,
                       (int64_t)nover, d, k, cand, cnt, D, I, over_list2, over_count + 1, over_list);
    MGP_LAUNCH_CHECK();
    MGP_HIP_TRY(hipMemcpyAsync(&nover, over_count + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    final_list = over_list2;
  }
  over_rows->resize((size_t)nover);
  if (nover > 0) {
    MGP_HIP_TRY(hipMemcpyAsync(over_rows->data(), final_list, (size_t)nover * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
  }
  return MGP_OK;
}

// gather / scatter helpers for the overflow fallback (device row lists)
int mgp_knn_gather_rows(const float* src, const int32_t* rows_dev, int64_t m, int w, float* dst, void* stream) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(m * w)), dim3(kBlock), 0, mgp_stream(stream), src, rows_dev, m, w, dst);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int mgp_knn_scatter_rows(const float* Ds, const int32_t* Is, const int32_t* rows_dev, int64_t m, int k, float* D,
                         int32_t* I, void* stream) {
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(m * k)), dim3(kBlock), 0, mgp_stream(stream), Ds, Is, rows_dev, m, k,
                     D, I);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// Morton (Z-curve) order of points with d <= 3: the row order handed to mgp_graph_tiles when the input
// order carries no locality (graph.py); same codes as the k-NN path above.
extern "C" size_t mgp_morton_order_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return 2 * mgp_align((size_t)n * sizeof(uint64_t)) + mgp_align((size_t)n * sizeof(int32_t)) + cub_sort_bytes(n) +
         mgp_align((size_t)1024 * 6 * sizeof(float)) + 4096;
}

extern "C" int mgp_morton_order(const float* x, int64_t n, int d, int32_t* order, void* work, size_t work_bytes,
                                void* stream) {
  if (!x || !order || !work || n <= 0 || n > INT_MAX || d < 1 || d > 3) return MGP_ERR_ARG;
  if (work_bytes < mgp_morton_order_workspace_bytes(n)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  uint64_t* codes_a = ar.take<uint64_t>(n);
  uint64_t* codes_b = ar.take<uint64_t>(n);
  int32_t* idx_a = ar.take<int32_t>(n);
  const size_t cub_bytes = cub_sort_bytes(n);
  void* cub = ar.take<char>(cub_bytes);
  float* boxpart = ar.take<float>(1024 * 6);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  const int bgrid = (int)std::min<int64_t>(1024, mgp_cdiv(n, kBlock));
  hipLaunchKernelGGL(minmax_kernel, dim3(bgrid), dim3(kBlock), 0, st, x, n, d, boxpart);
  MGP_LAUNCH_CHECK();
  std::vector<float> hp((size_t)bgrid * 6);
  MGP_HIP_TRY(hipMemcpyAsync(hp.data(), boxpart, hp.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  Box box;
  for (int a = 0; a < 3; ++a) {
    float lo = INFINITY, hi = -INFINITY;
    for (int b = 0; b < bgrid; ++b) { lo = std::min(lo, hp[(size_t)b * 6 + a]); hi = std::max(hi, hp[(size_t)b * 6 + 3 + a]); }
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) { box.lo[a] = std::isfinite(lo) ? lo : 0.f; box.inv[a] = 0.f; }
    else { box.lo[a] = lo; box.inv[a] = 2097151.0f / (hi - lo); }
  }
  hipLaunchKernelGGL(codes_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, x, n, d, box, codes_a, idx_a);
  MGP_LAUNCH_CHECK();
  hipcub::DoubleBuffer<uint64_t> kb(codes_a, codes_b);
  hipcub::DoubleBuffer<int32_t> vb(idx_a, order);
  size_t tb = cub_bytes;
  MGP_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(cub, tb, kb, vb, (int)n, 0, 63, st));
  if (vb.Current() != order)
    MGP_HIP_TRY(hipMemcpyAsync(order, vb.Current(), (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  return MGP_OK;
}
