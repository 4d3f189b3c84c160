// Exact brute-force k-NN (squared L2, ascending (d2, index)), bit-exact against oracle/knn_oracle.c.
//
// Replaces faiss IndexFlatL2 / IndexIVFFlat(nlist=1) (+ Gpu variants) `search` as called from
// manifold_gp/utils/nearest_neighbors.py:35-37.
//
// Pipeline per chunk of query rows (fp32 distance slab of 1-8 GiB of HBM, see chunk_rows; as many queries as points and
// N <= 92k: ONE chunk holding the whole N x N key matrix, of which a self-search computes the upper triangle only):
//   1. keys: d >= 32: dist_mfma_kernel (knn_mfma.hip, centred bf16-split GEMM form on the matrix cores);
//      else, or for a chunk where many rows fail the absolute check:
//      dist_tile_kernel   fp32 direct-difference distances, 128x128 tile per workgroup,
//                         8x8 register micro-tile per lane, operands staged k-major in LDS so
//                         a lane reads its 8 queries / 8 points with two ds_read_b128 each
//   2. select_kernel      one workgroup per query row.  T = the K'-th smallest key of the row (K' = k + pad):
//                         rows of 8192+ keys take the K'-th smallest of a 1/S sample (S = 16 up to ~61k keys)
//                         as an upper bound, keep the keys under it in LDS in ONE pass over the row and
//                         radix-select (11/11/10 bits) inside that list; long rows (expected list > LDS) and
//                         overflowing lists radix-select over the row with the bound as a filter, short rows
//                         without it.  fp64 re-evaluation of the K' candidates in the oracle's exact
//                         operation order (the selection's dominant cost at large d: K' rows of d
//                         floats per query, a strictly sequential fp64 chain per candidate); order by
//                         (d64, index) -- by counting for K' <= 256, bitonic for retry widths;
//                         sufficiency check
//                            d64[k-1] < T / (1 + gamma)        direct-difference keys (gamma = fp32
//                                                               relative error bound), or
//                            d64[k-1] + 2 E(x) < T             MFMA keys (knn_mfma.hip, absolute bound)
//                         which proves no unselected point can enter the top k.
//   3. rows that fail the check are redone with K' x 4 (up to 2048), then by exact_row_kernel:
//      fp64 distances to every point + k rounds of (d, index) arg-min.
#include <math.h>
#include <limits.h>
#include <vector>
#include <atomic>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

constexpr int kTile = 128;   // queries x points per workgroup
constexpr int kDK = 16;      // feature chunk staged in LDS
constexpr int kBlock = 256;
constexpr int kMaxKp = 2048;
constexpr int kMaxDimLds = 4096;

// ------------------------------------------------------------------ 1. distance tiles
// VEC = true (d % 4 == 0): 16-byte global loads, the next feature chunk is fetched into registers while
// the current one is consumed (one exposed global round trip per tile instead of one per chunk).
// The 8x8 micro-tile is computed on float2 pairs so that hipcc emits v_pk_add_f32 / v_pk_fma_f32: two
// IEEE fp32 results per lane per instruction, i.e. the packed-math VALU rate -- same roundings, same
// order over the features as the scalar form, so the selection bound gamma is unchanged.
typedef float knn_v2f __attribute__((ext_vector_type(2)));

template <bool VEC>
__global__ __launch_bounds__(kBlock) void dist_tile_kernel(const float* __restrict__ db, int64_t N, int d,
                                                           const float* __restrict__ q, int64_t nq,
                                                           float* __restrict__ out, int64_t ld) {
  __shared__ __attribute__((aligned(16))) float Qs[kDK][kTile + 4];
  __shared__ __attribute__((aligned(16))) float Ps[kDK][kTile + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int64_t p0 = (int64_t)blockIdx.x * kTile, q0 = (int64_t)blockIdx.y * kTile;
  knn_v2f acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = knn_v2f{0.f, 0.f};

  // VEC staging: float4 f -> (row = f / 4, kq = f % 4): rows of 64 contiguous bytes, 2 float4 per lane
  // per matrix; rows past the end are clamped (their results are never stored)
  float4 rq[2], rp[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = tid + h * kBlock;
      const int row = f >> 2, kq = f & 3;
      const int64_t qr = (q0 + row < nq) ? q0 + row : nq - 1;
      const int64_t pr = (p0 + row < N) ? p0 + row : N - 1;
      const int kg = k0 + 4 * kq;
      if (kg < d) {
        rq[h] = *reinterpret_cast<const float4*>(q + qr * d + kg);
        rp[h] = *reinterpret_cast<const float4*>(db + pr * d + kg);
      } else {
        rq[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        rp[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (VEC) fetch(0);

  for (int k0 = 0; k0 < d; k0 += kDK) {
    if (VEC) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int f = tid + h * kBlock;
        const int row = f >> 2, kq = 4 * (f & 3);
        Qs[kq + 0][row] = rq[h].x; Qs[kq + 1][row] = rq[h].y; Qs[kq + 2][row] = rq[h].z; Qs[kq + 3][row] = rq[h].w;
        Ps[kq + 0][row] = rp[h].x; Ps[kq + 1][row] = rp[h].y; Ps[kq + 2][row] = rp[h].z; Ps[kq + 3][row] = rp[h].w;
      }
    } else {
      // element e -> (row = e / 16, kk = e % 16): 64 contiguous bytes per row
#pragma unroll
      for (int e = tid; e < kTile * kDK; e += kBlock) {
        const int row = e >> 4, kk = e & 15;
        const int kg = k0 + kk;
        float qv = 0.f, pv = 0.f;
        if (kg < d) {
          if (q0 + row < nq) qv = q[(q0 + row) * d + kg];
          if (p0 + row < N) pv = db[(p0 + row) * d + kg];
        }
        Qs[kk][row] = qv;
        Ps[kk][row] = pv;
      }
    }
    __syncthreads();
    if (VEC && k0 + kDK < d) fetch(k0 + kDK);      // in flight during the 16 feature steps below
#pragma unroll
    for (int kk = 0; kk < kDK; ++kk) {
      const float4 q_lo = *reinterpret_cast<const float4*>(&Qs[kk][ty * 8]);
      const float4 q_hi = *reinterpret_cast<const float4*>(&Qs[kk][ty * 8 + 4]);
      const float4 p_lo = *reinterpret_cast<const float4*>(&Ps[kk][tx * 8]);
      const float4 p_hi = *reinterpret_cast<const float4*>(&Ps[kk][tx * 8 + 4]);
      const float qa[8] = {q_lo.x, q_lo.y, q_lo.z, q_lo.w, q_hi.x, q_hi.y, q_hi.z, q_hi.w};
      const knn_v2f pa[4] = {knn_v2f{p_lo.x, p_lo.y}, knn_v2f{p_lo.z, p_lo.w}, knn_v2f{p_hi.x, p_hi.y},
                             knn_v2f{p_hi.z, p_hi.w}};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const knn_v2f qq = knn_v2f{qa[i], qa[i]};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const knn_v2f df = qq - pa[j];
          acc[i][j] = __builtin_elementwise_fma(df, df, acc[i][j]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t qr = q0 + ty * 8 + i;
    if (qr >= nq) continue;
    const int64_t pc = p0 + tx * 8;
    float* o = out + qr * ld + pc;
    if (pc + 8 <= N) {
      *reinterpret_cast<float4*>(o) = make_float4(acc[i][0].x, acc[i][0].y, acc[i][1].x, acc[i][1].y);
      *reinterpret_cast<float4*>(o + 4) = make_float4(acc[i][2].x, acc[i][2].y, acc[i][3].x, acc[i][3].y);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (pc + j < N) o[j] = (j & 1) ? acc[i][j >> 1].y : acc[i][j >> 1].x;
    }
  }
}

// ------------------------------------------------------------------ helpers
// exact oracle distance: fp64, ascending feature order, no contraction (oracle/knn_oracle.c)
__device__ __forceinline__ double oracle_d2(const float* __restrict__ qrow, const float* __restrict__ xrow, int d) {
  double acc = 0.0;
  int j = 0;
  if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(xrow) & 15) == 0)) {
    // 32 features per pass: the 8 x 16-byte loads are issued before the (strictly sequential, oracle-
    // ordered) accumulation consumes them, so the chain waits for memory once per pass, not per feature
    for (; j + 32 <= d; j += 32) {
      float4 x4[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x4[u] = *reinterpret_cast<const float4*>(xrow + j + 4 * u);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float xs[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double df = __dsub_rn((double)qrow[j + 4 * u + t], (double)xs[t]);
          acc = __dadd_rn(acc, __dmul_rn(df, df));
        }
      }
    }
  }
  for (; j < d; ++j) {
    const double df = __dsub_rn((double)qrow[j], (double)xrow[j]);
    acc = __dadd_rn(acc, __dmul_rn(df, df));
  }
  return acc;
}

__device__ __forceinline__ bool cand_less(double d1, int i1, double d2, int i2) {
  return (d1 < d2) || (d1 == d2 && i1 < i2);
}

// block-wide exclusive scan of one int per thread (256 threads); returns exclusive prefix, total in *total
__device__ int block_excl_scan(int v, int* sh_wave /*[4]*/, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) sh_wave[w] = inc;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += sh_wave[i];
  *total = sh_wave[0] + sh_wave[1] + sh_wave[2] + sh_wave[3];
  __syncthreads();
  return base + inc - v;
}

struct SelectArgs {
  const float* dist;     // [rows_in_chunk, ld] fp32 distances
  int64_t ld;
  int64_t N;
  const float* db;
  const float* q;        // queries of this chunk [rows_in_chunk, d]
  int d;
  int k;
  int Kp;                // sort width (pow2, <= kMaxKp)
  int cand;              // candidates re-evaluated in fp64 (<= Kp; 0: Kp)
  const int* rows;       // nullable: row list (local row ids)
  float* D;              // [rows_in_chunk, k] (chunk-local base)
  int32_t* I;
  int* fail_list;
  int* fail_count;
  double gamma;
  // absolute form of the sufficiency check (keys from knn_mfma.hip); qn2 == nullptr: relative form
  const float* qn2;          // |c_x|^2 per chunk row
  const unsigned* r2max;     // bits of R^2
  double alpha, beta;        // E = alpha |c_x| R + beta (|c_x| + R)^2
  // candidate lists of the filtered key pass (knn_mfma.hip) instead of a slab row: dist == nullptr
  const uint2* lists;        // [rows_in_chunk, list_cap] {key bits, point index}
  const int* counts;         // entries drawn per row (may exceed list_cap: such a row fails over to the slab pipeline)
  int list_cap;
};

// Visit every key of a slab row: 4 x 16-byte loads per lane are issued before the first key is used.
// With one 4-byte load per iteration the histogram's LDS atomic forces a wait per key and the row
// scan is a chain of ~230 dependent HBM round trips per pass (measured 3.5 ms per 3200-row chunk).
template <class F>
__device__ __forceinline__ void for_each_key(const uint32_t* __restrict__ keys, int64_t N, int tid, F f) {
  const int64_t nq = N >> 2;
  const uint4* __restrict__ k4 = reinterpret_cast<const uint4*>(keys);
  // software pipelined (round 3): the next batch of 4 x 16 bytes per lane is requested before the current one is
  // consumed -- the consumer's LDS atomics otherwise sit between two memory round trips with nothing in flight (a
  // 240 KB row was 15 dependent HBM round trips, 25 us; the slab does not fit the Infinity Cache)
  uint4 v[4], w[4];
  auto load = [&](uint4 (&dst)[4], int64_t b0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t qi = b0 + tid + u * kBlock;
      dst[u] = k4[qi < nq ? qi : (nq > 0 ? nq - 1 : 0)];
    }
  };
  auto use = [&](const uint4 (&src)[4], int64_t b0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t qi = b0 + tid + u * kBlock;
      if (qi < nq) {
        f(src[u].x, 4 * qi); f(src[u].y, 4 * qi + 1); f(src[u].z, 4 * qi + 2); f(src[u].w, 4 * qi + 3);
      }
    }
  };
  if (nq > 0) load(v, 0);
  for (int64_t b0 = 0; b0 < nq; b0 += 8 * kBlock) {
    const int64_t b1 = b0 + 4 * kBlock;
    if (b1 < nq) load(w, b1);
    use(v, b0);
    if (b1 < nq) {
      if (b1 + 4 * kBlock < nq) load(v, b1 + 4 * kBlock);
      use(w, b1);
    }
  }
  for (int64_t i = (nq << 2) + tid; i < N; i += kBlock) f(keys[i], i);
}

// ------------------------------------------------------------------ 2. select + re-rank
constexpr int kListCap = 3840;      // LDS list of the keys under the sampled threshold (with the 24 KB of candidate
                                    // arrays of the widest retry the kernel stays under 64 KB of LDS)
constexpr int kSampleBlock = 64;    // sampled keys come in runs of 64 (256-byte loads)
// fp64 re-rank with staged products (select_kernel): candidates x (F + 1) doubles share the dead key list's storage
// (30 720 bytes) with the query row (d <= kMaxDimLds floats): up to kRerankWide candidates at 32 features per chunk, up
// to kRerankMax at 16; kRerankItems = float4 loads per thread and chunk (kRerankMax * 4 / 256, rounded up)
constexpr int kRerankWide = 96;
constexpr int kRerankMax = 192;
constexpr int kRerankItems = 3;

// rank-th smallest (1-based) of the keys `scan` visits: 3-pass radix select (11 / 11 / 10 bits) with an LDS
// histogram.  Returns the key; *rank_eq = how many of the keys equal to it are among the `rank` smallest.
// passes = 2: only the top 22 bits are resolved and the largest key of that bucket is returned -- an upper
// bound of the rank-th smallest (*rank_eq is then meaningless).
template <class Scan>
__device__ uint32_t radix_kth(Scan scan, int rank, int* hist, int* sh_wave, int* sh_bin, int* sh_rank, int* rank_eq,
                              int passes = 3) {
  const int tid = threadIdx.x;
  uint32_t prefix = 0, pmask = 0;
  const int shifts[3] = {21, 10, 0};
  const int nbits[3] = {11, 11, 10};
  for (int ps = 0; ps < passes; ++ps) {
    const int nb = 1 << nbits[ps];
    for (int i = tid; i < nb; i += kBlock) hist[i] = 0;
    __syncthreads();
    const int sh = shifts[ps];
    scan([&](uint32_t key, int64_t) {
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> sh) & (nb - 1)], 1);
    });
    __syncthreads();
    // 256 threads x (nb/256) bins: find the bin holding the rank-th key
    const int per = nb / kBlock;
    int local = 0;
    for (int j = 0; j < per; ++j) local += hist[tid * per + j];
    int total;
    int excl = block_excl_scan(local, sh_wave, &total);
    if (rank > excl && rank <= excl + local) {
      int run = excl;
      for (int j = 0; j < per; ++j) {
        const int h = hist[tid * per + j];
        if (rank <= run + h) { *sh_bin = tid * per + j; *sh_rank = rank - run; break; }
        run += h;
      }
    }
    __syncthreads();
    prefix |= ((uint32_t)*sh_bin) << shifts[ps];
    pmask |= ((uint32_t)(nb - 1)) << shifts[ps];
    rank = *sh_rank;
    __syncthreads();
  }
  *rank_eq = rank;
  return passes == 3 ? prefix : (prefix | 0x3ffu);
}

#ifdef MGP_SEL_LAB
__device__ unsigned long long g_sel_stamp[8];
#define SEL_STAMP(K) do { __syncthreads(); const unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && blockIdx.x % 61 == 0) atomicAdd(&g_sel_stamp[K], t_ - t_prev); t_prev = __builtin_readcyclecounter(); } while (0)
#else
#define SEL_STAMP(K) do { } while (0)
#endif
// One workgroup per query row.  Rows of 8192+ keys are not radix-selected over the whole row: the K'-th
// smallest of a 1/S sample of the row (runs of 64 keys, S = 16 up to ~61k keys) bounds the K'-th smallest
// of the row from above, ONE pass over the row keeps the ~S K' keys under that bound in LDS, and the exact
// K'-th smallest (the same T as a full-row select) comes from that list.  A list that overflows falls back
// to the three full-row passes.
__global__ __launch_bounds__(kBlock) void select_kernel(SelectArgs a) {
  __shared__ int hist[2048];
  __shared__ int sh_wave[4];
  __shared__ int sh_bin, sh_rank, sh_cnt_lt, sh_cnt_eq, sh_cnt_list;
  __shared__ double sh_dk;
  // candidate arrays sized by the launch (Kp x 12 bytes): the first attempt's Kp <= 256 leaves room for three
  // workgroups per CU (fp64-chain-bound re-rank of one overlaps the memory-bound scans of the others)
  extern __shared__ __attribute__((aligned(16))) unsigned char cand_mem[];
  double* cand_d = reinterpret_cast<double*>(cand_mem);
  int* cand_idx = reinterpret_cast<int*>(cand_mem + (size_t)a.Kp * sizeof(double));
  // the key list is dead before the query row is staged: same storage
  __shared__ __attribute__((aligned(16))) uint32_t list_mem[2 * kListCap];
  uint32_t* list_key = list_mem;
  int* list_idx = reinterpret_cast<int*>(list_mem + kListCap);
  float* qrow_s = reinterpret_cast<float*>(list_mem);
  static_assert(kMaxDimLds * sizeof(float) <= sizeof(uint32_t) * 2 * kListCap, "query row staging");

  const int tid = threadIdx.x;
#ifdef MGP_SEL_LAB
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  // consecutive rows to one XCD (blocks are dealt round-robin over the XCDs): neighbouring queries re-rank largely the same
  // candidate rows, which then come from that XCD's L2 instead of being fetched into all eight
  const int row = a.rows ? a.rows[blockIdx.x] : mgp_xcd_block((int)blockIdx.x, (int)gridDim.x);
  const uint32_t* keys = a.lists ? nullptr : reinterpret_cast<const uint32_t*>(a.dist + (int64_t)row * a.ld);
  const int64_t N = a.N;
  const int Kp = a.Kp;
  const int cand = a.cand > 0 && a.cand < Kp ? a.cand : Kp;
  const int want = (int64_t)cand < N ? cand : (int)N;   // number of real candidates
  int rank = want;   // how many of the keys equal to T are wanted
  uint32_t T = 0xffffffffu;
  bool from_list = false;
  int n_list = 0;
  if (a.lists) {
    // the filtered key pass left every key of the row under the row's bound (>= its want-th smallest key: at least
    // `want` of them) in the row's list: the exact T comes from the list, as from the one-pass list below.  A list that
    // overflowed, or holds fewer than `want` keys (a retry wider than the filter was built for; keys of the sample and of
    // the full pass that differ in the last bit at the bound), fails the row: the caller redoes it on a slab.
    n_list = a.counts[row];
    if (n_list > a.list_cap || n_list > kListCap || n_list < want || want >= N) {
      if (tid == 0) {
        const int slot = atomicAdd(a.fail_count, 1);
        a.fail_list[slot] = row;
      }
      return;
    }
    const uint2* lrow = a.lists + (int64_t)row * a.list_cap;
    for (int j = tid; j < n_list; j += kBlock) {
      const uint2 e = lrow[j];
      list_key[j] = e.x; list_idx[j] = (int)e.y;
    }
    __syncthreads();
    SEL_STAMP(0);    // list loaded
    T = radix_kth([&](auto f) { for (int j = tid; j < n_list; j += kBlock) f(list_key[j], (int64_t)list_idx[j]); },
                  want, hist, sh_wave, &sh_bin, &sh_rank, &rank);
    from_list = true;
    SEL_STAMP(1);    // radix select
  } else if (want < N) {
    // sampling stride: the smallest power of two from 16 up whose sample (runs of 64 keys) fits the LDS list
    int S = 16;
    while (((N / kSampleBlock + S - 1) / S) * kSampleBlock > kListCap) S <<= 1;
    const int64_t n_runs = (N / kSampleBlock + S - 1) / S;            // sampled runs (all complete)
    const int64_t n_samp = n_runs * kSampleBlock;
    uint32_t tau = 0xffffffffu;                                       // upper bound of the want-th smallest key
    if (N >= 8192 && n_samp >= 2 * want) {
      // ---- sample -> LDS, its want-th smallest bounds the row's want-th smallest
      const int phase = row & (S - 1);
      for (int j = tid; j < (int)n_samp; j += kBlock) {
        int64_t run = (int64_t)(j / kSampleBlock) * S + phase;
        if ((run + 1) * kSampleBlock > N) run -= phase;                // last group: its first run is complete
        list_key[j] = keys[run * kSampleBlock + (j & (kSampleBlock - 1))];
      }
      if (tid == 0) sh_cnt_list = 0;
      __syncthreads();
      int dummy;
      tau = radix_kth([&](auto f) { for (int j = tid; j < (int)n_samp; j += kBlock) f(list_key[j], (int64_t)j); },
                      want, hist, sh_wave, &sh_bin, &sh_rank, &dummy, 2);
      __syncthreads();
      if ((int64_t)S * want <= kListCap / 2) {
        // ---- one pass over the row: keys <= tau into the list (at least `want` of them exist, ~S want expected)
        for_each_key(keys, N, tid, [&](uint32_t key, int64_t i) {
          if (key <= tau) {
            const int slot = atomicAdd(&sh_cnt_list, 1);
            if (slot < kListCap) { list_key[slot] = key; list_idx[slot] = (int)i; }
          }
        });
        __syncthreads();
        n_list = sh_cnt_list;
        from_list = n_list <= kListCap;
        if (from_list) {
          T = radix_kth([&](auto f) { for (int j = tid; j < n_list; j += kBlock) f(list_key[j], (int64_t)list_idx[j]); },
                        want, hist, sh_wave, &sh_bin, &sh_rank, &rank);
        }
      }
    }
    if (!from_list) {
      // radix select over the row; keys above the sampled bound cannot be among the `want` smallest and skip
      // the histogram (its LDS atomics are what a full-row pass costs): long rows and overflowing lists land here
      T = radix_kth([&](auto f) { for_each_key(keys, N, tid, [&](uint32_t key, int64_t i) { if (key <= tau) f(key, i); }); },
                    want, hist, sh_wave, &sh_bin, &sh_rank, &rank);
    }
    // T: exact K'-th smallest key; `rank` of the keys equal to T are still wanted
  }
  // ---- collect candidates: all keys < T, plus `rank` keys == T (any of them; see header)
  if (tid == 0) { sh_cnt_lt = 0; sh_cnt_eq = 0; }
  for (int i = tid; i < Kp; i += kBlock) { cand_idx[i] = INT_MAX; cand_d[i] = INFINITY; }
  __syncthreads();
  auto keep = [&](uint32_t key, int64_t i) {
    if (key < T) {
      const int slot = atomicAdd(&sh_cnt_lt, 1);
      cand_idx[slot] = (int)i;
    } else if (key == T) {
      const int e = atomicAdd(&sh_cnt_eq, 1);
      if (e < rank) cand_idx[want - 1 - e] = (int)i;
    }
  };
  if (want < N) {
    if (from_list) {
      for (int j = tid; j < n_list; j += kBlock) keep(list_key[j], (int64_t)list_idx[j]);
    } else {
      for_each_key(keys, N, tid, keep);
    }
  } else {
    for (int i = tid; i < want; i += kBlock) cand_idx[i] = i;
  }
  __syncthreads();   // the list is dead from here on
  SEL_STAMP(2);      // candidates collected
  // query row to LDS (falls back to global reads for very wide features)
  const float* qrow = a.q + (int64_t)row * a.d;
  const bool q_lds = a.d <= kMaxDimLds;
  if (q_lds) for (int j = tid; j < a.d; j += kBlock) qrow_s[j] = qrow[j];
  __syncthreads();

  SEL_STAMP(3);      // query row staged
  // ---- fp64 re-evaluation in the oracle's operation order
  // One thread per candidate running oracle_d2 (below, kept for wide retry sets and odd shapes) leaves 3/4 of the
  // workgroup idle and makes every load instruction touch 64 different rows.  (Measured, round 3: 22.3 -> 21.6 ms per
  // 60k x 784 search, no more -- with three rows per CU in flight the kernel is bound by its TRAFFIC, not by a row's
  // latency: per 7 500-row chunk it requests the 1.8 GB key slab once and 75 candidate rows of 3 136 bytes per query,
  // another 1.76 GB, in 0.77 ms = 4.6 TB/s through L2, 2.8 - 3.3 TB/s from HBM by the counters.)  The SUM of a
  // candidate must run in ascending feature order (the oracle's rounding), the PRODUCTS (q_j - x_j)^2 need not: per
  // chunk of F features all 256 threads form the products of all candidates -- 8 consecutive lanes read one
  // candidate's 128 bytes, coalesced -- and leave them in LDS as doubles; thread c then adds candidate c's F products in
  // order.  Same operations, same order, same bits; the next chunk's loads are in flight during the adds.
  const int q_floats = ((a.d + 3) & ~3) + 4;
  auto fits = [&](int f) { return (size_t)q_floats * 4 + (size_t)want * (f + 1) * 8 <= sizeof(list_mem) && want * (f >> 2) <= kRerankItems * kBlock; };
  const int F = (want <= kRerankWide && fits(32)) ? 32 : 16;   // products staged per candidate and chunk
  const bool staged = q_lds && want <= kRerankMax && fits(F) && (a.d & 3) == 0 && (reinterpret_cast<uintptr_t>(a.db) & 15) == 0;
  if (staged) {
    double* prod = reinterpret_cast<double*>(list_mem + q_floats);     // behind the query row; [want][F + 1]
    const int PS = F + 1, F4 = F >> 2;
    const int items = want * F4;                                // (candidate, float4) pairs per chunk
    double acc = 0.0;
    // two chunks of candidate-row pieces in flight: a chunk's loads are a round trip of 1 - 2 us, its ordered adds 0.15 us;
    // with one chunk ahead the 25 chunks of a 784-feature row were 25 exposed round trips of the row's ~83 us
    float4 xa[kRerankItems], xb[kRerankItems];
    auto fetch = [&](float4 (&xv)[kRerankItems], int j0) {
#pragma unroll
      for (int u = 0; u < kRerankItems; ++u) {
        const int it = tid + u * kBlock;
        const int c = it / F4, q4 = it - c * F4;
        const bool on = it < items && j0 + 4 * q4 < a.d;
        const int idx = cand_idx[on ? c : 0];
        xv[u] = *reinterpret_cast<const float4*>(a.db + (int64_t)idx * a.d + (on ? j0 + 4 * q4 : 0));
      }
    };
    auto chunk = [&](float4 (&xv)[kRerankItems], int j0) {
#pragma unroll
      for (int u = 0; u < kRerankItems; ++u) {
        const int it = tid + u * kBlock;
        const int c = it / F4, q4 = it - c * F4;
        if (it < items && j0 + 4 * q4 < a.d) {
          const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
          double* o = prod + c * PS + 4 * q4;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const double df = __dsub_rn((double)qrow_s[j0 + 4 * q4 + t], (double)xs[t]);
            o[t] = __dmul_rn(df, df);
          }
        }
      }
      __syncthreads();
      if (j0 + 2 * F < a.d) fetch(xv, j0 + 2 * F);             // this register set's next chunk: in flight during two chunks' adds
      if (tid < want) {
        const int nf = a.d - j0 < F ? a.d - j0 : F;
        const double* pc = prod + tid * PS;
        for (int j = 0; j < nf; ++j) acc = __dadd_rn(acc, pc[j]);
      }
      __syncthreads();
    };
    fetch(xa, 0);
    if (F < a.d) fetch(xb, F);
    for (int j0 = 0; j0 < a.d; j0 += 2 * F) {
      chunk(xa, j0);
      if (j0 + F < a.d) chunk(xb, j0 + F);
    }
    if (tid < want) cand_d[tid] = acc;
  } else {
    for (int c = tid; c < want; c += kBlock) {
      const int idx = cand_idx[c];
      cand_d[c] = oracle_d2(q_lds ? qrow_s : qrow, a.db + (int64_t)idx * a.d, a.d);
    }
  }
  __syncthreads();
  SEL_STAMP(4);      // re-rank
  // ---- order by (d64, index).  Up to 256 candidates: every candidate counts the candidates ahead of it
  // (all pairs are distinct, so the counts are the sorted positions; LDS reads are broadcasts) -- one
  // barrier instead of the 28+ of a bitonic network; wider retry sets: bitonic sort of Kp (pow2) entries.
  const bool by_rank = want <= kBlock;
  int my_pos = INT_MAX;
  double my_d = 0.0;
  int my_i = 0;
  if (by_rank) {
    // up to 64 candidates... 256: the counting is split over the workgroup -- `parts` threads per candidate, each over a
    // slice of the others -- and the partial counts meet in LDS (the histogram is dead): with one thread per candidate
    // a 75-candidate row kept two of the four waves busy for 75 dependent LDS round trips (18 % of the kernel, stamps)
    const int parts = want <= 64 ? 4 : want <= 128 ? 2 : 1;
    int* ahead_s = hist;
    if (parts > 1) {
      if (tid < want) ahead_s[tid] = 0;
      __syncthreads();
    }
    const int c = tid % (kBlock / parts), part = tid / (kBlock / parts);
    int ahead = 0;
    if (c < want) {
      my_d = cand_d[c];
      my_i = cand_idx[c];
      const int per = (want + parts - 1) / parts;
      const int j0 = part * per, j1 = j0 + per < want ? j0 + per : want;
#pragma unroll 8
      for (int j = j0; j < j1; ++j) ahead += cand_less(cand_d[j], cand_idx[j], my_d, my_i) ? 1 : 0;
      if (parts > 1) atomicAdd(&ahead_s[c], ahead);
    }
    if (parts > 1) {
      __syncthreads();
      if (c < want) ahead = ahead_s[c];
    }
    if (c < want && part == 0) {
      my_pos = ahead;
      if (ahead == a.k - 1) sh_dk = my_d;
    }
    __syncthreads();
  } else {
    for (int size = 2; size <= Kp; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int t = tid; t < Kp / 2; t += kBlock) {
          const int lo = (t / stride) * stride * 2 + (t % stride);
          const int hi = lo + stride;
          const bool up = ((lo & size) == 0);
          const double dl = cand_d[lo], dh = cand_d[hi];
          const int il = cand_idx[lo], ih = cand_idx[hi];
          const bool swap = up ? cand_less(dh, ih, dl, il) : cand_less(dl, il, dh, ih);
          if (swap) { cand_d[lo] = dh; cand_d[hi] = dl; cand_idx[lo] = ih; cand_idx[hi] = il; }
        }
        __syncthreads();
      }
    }
  }
  SEL_STAMP(5);      // ordered
  // ---- sufficiency check, then write or flag
  const double dk = by_rank ? sh_dk : cand_d[a.k - 1];
  bool ok = true;
  if (want < N) {
    const double t32 = (double)__uint_as_float(T);
    if (a.qn2) {
      const double nx = sqrt((double)a.qn2[row] * (1.0 + 1e-6)), R = sqrt((double)__uint_as_float(*a.r2max) * (1.0 + 1e-6));
      const double E = a.alpha * nx * R + a.beta * (nx + R) * (nx + R);
      ok = dk + 2.0 * E < t32 * (1.0 - 1e-12);
    } else {
      ok = dk < t32 / (1.0 + a.gamma) * (1.0 - 1e-12);
    }
    if (!(t32 < (double)INFINITY)) ok = false;    // overflowed fp32 keys order nothing: exact path
  }
  if (ok) {
    if (by_rank) {
      if (my_pos < a.k) {
        a.D[(int64_t)row * a.k + my_pos] = (float)my_d;
        a.I[(int64_t)row * a.k + my_pos] = my_i;
      }
    } else {
      for (int t = tid; t < a.k; t += kBlock) {
        a.D[(int64_t)row * a.k + t] = (float)cand_d[t];
        a.I[(int64_t)row * a.k + t] = cand_idx[t];
      }
    }
  } else if (tid == 0) {
    const int slot = atomicAdd(a.fail_count, 1);
    a.fail_list[slot] = row;
  }
}

// ------------------------------------------------------------------ 2b. per-row bounds of the filtered key pass
// One workgroup per query row: the want-th smallest of the row's keys to the S sampled points (a subset of the row:
// its want-th smallest bounds the row's from above; two radix passes = the top 22 bits, rounded up) -> bounds[row],
// widened by 2^-12 relative (the full pass recomputes these keys with the operands' roles swapped: last-bit
// differences) and kept finite (keys that overflowed pass no finite bound: such rows fail over to the slab pipeline).
__global__ __launch_bounds__(kBlock) void bound_kernel(const float* __restrict__ samp, int64_t ld, int S, int want,
                                                       float* __restrict__ bounds) {
  __shared__ int hist[2048];
  __shared__ int sh_wave[4];
  __shared__ int sh_bin, sh_rank;
  const int tid = threadIdx.x;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(samp + (int64_t)blockIdx.x * ld);
  int dummy;
  const uint32_t tau = radix_kth([&](auto f) { for (int j = tid; j < S; j += kBlock) f(keys[j], (int64_t)j); }, want, hist,
                                 sh_wave, &sh_bin, &sh_rank, &dummy, 2);
  if (tid == 0) {
    float b = __uint_as_float(tau);
    b = b < 3.0e38f ? b * (1.f + 0x1p-12f) : 3.0e38f;     // NaN bits (tau above the infinity pattern) land on the cap too
    if (!(b < 3.0e38f)) b = 3.0e38f;
    bounds[blockIdx.x] = b;
  }
}

// The same bound with one WAVE per row and the row's sampled keys in registers (S <= 4096: up to ~65k points at stride 16):
// the top 22 bits of the want-th smallest key bit by bit -- count the keys <= (prefix | ones below the bit), a DPP wave sum,
// keep the bit clear when `want` of them are -- instead of two histogram passes with a dozen workgroup barriers (0.52 -> 
// 0.2x ms at 60k rows of 3750 keys).  Same value as bound_kernel's.
constexpr int kBoundRegs = 64;
__global__ __launch_bounds__(kBlock) void bound_wave_kernel(const float* __restrict__ samp, int64_t ld, int S, int want,
                                                            float* __restrict__ bounds, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(samp + row * ld);
  uint32_t kreg[kBoundRegs];
#pragma unroll
  for (int u = 0; u < kBoundRegs; ++u) {
    const int i = u * 64 + lane;
    kreg[u] = i < S ? keys[i] : 0xffffffffu;
  }
  uint32_t prefix = 0u;
  for (int bit = 31; bit >= 10; --bit) {
    const uint32_t c = prefix | ((1u << bit) - 1u);
    int n = 0;
#pragma unroll
    for (int u = 0; u < kBoundRegs; ++u) n += kreg[u] <= c ? 1 : 0;
    n += __builtin_amdgcn_update_dpp(0, n, 0x111, 0xf, 0xf, true);
    n += __builtin_amdgcn_update_dpp(0, n, 0x112, 0xf, 0xf, true);
    n += __builtin_amdgcn_update_dpp(0, n, 0x114, 0xf, 0xf, true);
    n += __builtin_amdgcn_update_dpp(0, n, 0x118, 0xf, 0xf, true);
    n += __builtin_amdgcn_update_dpp(0, n, 0x142, 0xa, 0xf, true);
    n += __builtin_amdgcn_update_dpp(0, n, 0x143, 0xc, 0xf, true);
    if (__builtin_amdgcn_readlane(n, 63) < want) prefix |= 1u << bit;
  }
  if (lane == 0) {
    float b = __uint_as_float(prefix | 0x3ffu);
    b = b < 3.0e38f ? b * (1.f + 0x1p-12f) : 3.0e38f;
    if (!(b < 3.0e38f)) b = 3.0e38f;
    bounds[row] = b;
  }
}

// ------------------------------------------------------------------ 3. exact fallback
__global__ __launch_bounds__(kBlock) void exact_row_kernel(const float* __restrict__ db, int64_t N, int d,
                                                           const float* __restrict__ q, const int* __restrict__ rows,
                                                           int k, double* __restrict__ scratch, float* __restrict__ D,
                                                           int32_t* __restrict__ I) {
  __shared__ double sh_d[kBlock];
  __shared__ int sh_i[kBlock];
  __shared__ double last_d;
  __shared__ int last_i;
  const int tid = threadIdx.x;
  const int row = rows[blockIdx.x];
  double* dist = scratch + (int64_t)blockIdx.x * N;
  const float* qrow = q + (int64_t)row * d;
  for (int64_t i = tid; i < N; i += kBlock) dist[i] = oracle_d2(qrow, db + i * d, d);
  if (tid == 0) { last_d = -1.0; last_i = -1; }
  __syncthreads();
  for (int t = 0; t < k; ++t) {
    double bd = INFINITY;
    int bi = INT_MAX;
    const double ld_ = last_d;
    const int li = last_i;
    for (int64_t i = tid; i < N; i += kBlock) {
      const double v = dist[i];
      if (cand_less(ld_, li, v, (int)i) && cand_less(v, (int)i, bd, bi)) { bd = v; bi = (int)i; }
    }
    sh_d[tid] = bd; sh_i[tid] = bi;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
      if (tid < s && cand_less(sh_d[tid + s], sh_i[tid + s], sh_d[tid], sh_i[tid])) {
        sh_d[tid] = sh_d[tid + s]; sh_i[tid] = sh_i[tid + s];
      }
      __syncthreads();
    }
    if (tid == 0) {
      last_d = sh_d[0]; last_i = sh_i[0];
      D[(int64_t)row * k + t] = (float)sh_d[0];
      I[(int64_t)row * k + t] = sh_i[0];
    }
    __syncthreads();
  }
}

constexpr int kExactBatch = 16;

// candidate distances on the matrix cores (knn_mfma.hip) from 32 features up; mgp_knn_set_mfma(0) forces
// the direct-difference tiles
std::atomic<int> g_knn_mfma{1};
std::atomic<int> g_knn_sym{1};        // self-search: upper-triangle key tiles only (mgp_knn_set_symmetric(0): every tile)
std::atomic<int64_t> g_last_direct_chunks{0};
// (the centring + split of the points is a fixed ~1 ms at 60k x 784: it pays from ~1000 queries on)
bool use_mfma(int d, int64_t n) { return g_knn_mfma && d >= 32 && n >= 1024; }

// first attempt: k + pad candidates (each costs a d-float row read in the fp64 re-rank: the selection's dominant traffic
// at large d); MFMA keys carry an absolute error: a wider pad there keeps the retry launches rare
int first_candidates(int k, bool mfma) {
  int want = mfma ? k + (k / 2 > 24 ? k / 2 : 24) : k + (k / 4 > 16 ? k / 4 : 16);
  // whole waves of candidates: the re-rank's ordered fp64 sums run one thread per candidate and a wave instruction costs the
  // same with 11 active lanes as with 64 -- K' = 75 (k = 50) held two waves for 784 dependent adds, K' = 64 holds one and fills
  // the product rounds exactly (64 x 8 float4 items = 2 x 256 threads).  Down to k + max(k / 4, 12) only; the few rows whose
  // check then fails (RMNIST-like 60k: 148, Gaussian 60k x 784: 11) are redone with 4 K' candidates from the same lists
  // (tools/lab/knn_pad.py: 12.5 -> 12.0 ms, identical lists).
  if (mfma && want > 64) {
    const int lo = k + (k / 4 > 12 ? k / 4 : 12);
    const int w64 = want / 64 * 64;
    if (w64 >= lo) want = w64;
  }
#ifdef MGP_KNN_PAD_LAB
  if (const char* e = getenv("MGP_KNN_CAND")) return atoi(e);
#endif
  return want;
}

// Candidate filter (round 5): the matrix-core keys of a large search are not written to an N x n slab and read back by
// the select kernel (60k x 60k: 14.4 GB each way, the key kernel's epilogue and the select pass were both bound by it);
// the key pass keeps the ~stride K' keys per row that lie under a per-row bound from a 1/stride sample of the points.
// 0: off; 1: searches of >= 4096 queries against >= 16384 points; 2: every matrix-core search the lists can serve (tests).
std::atomic<int> g_knn_filter{1};
constexpr int kLogPerRow = 2048;      // log entries per query row of a chunk (~stride K' = 1200 expected at k = 50)
std::atomic<int64_t> g_last_filter_failover{-1};

struct FilterPlan {
  bool on;
  int stride;          // every stride-th point is sampled
  int64_t S, ldS;      // sampled points, row pitch of their key slab
  int64_t qc;          // query rows per chunk (the lists of a chunk: qc x kListCap x 8 bytes)
  int64_t fr;          // rows per fail-over batch (slab pipeline)
};

FilterPlan filter_plan(int64_t N, int64_t n, int d, int k, bool indexed) {
  FilterPlan f{};
  const int mode = g_knn_filter;
  if (!mode || !g_knn_mfma || d < 32) return f;
  if (mode == 1 && (N < 16384 || n < 4096)) return f;
  if (mode == 2 && !(indexed || use_mfma(d, n))) return f;
  if (N < 1024) return f;
  const int want = first_candidates(k, true);
  int stride = 16;
  while (stride > 4 && (int64_t)stride * want > kListCap * 2 / 5) stride >>= 1;   // expected list: stride x want keys
  if ((int64_t)stride * want > kListCap * 2 / 5) return f;
  const int64_t S = mgp_cdiv(N, stride);
  if (S < 2 * want) return f;
  f.on = true;
  f.stride = stride;
  f.S = S;
  f.ldS = mgp_cdiv(S, 4) * 4;
  const int64_t ncap = mgp_cdiv(n, kTile) * kTile;
  f.qc = ncap < 262144 ? ncap : 262144;
  // the keys to the sampled points (qc x S floats) within 4 GiB: whole rounds of 8 XCDs x 128-row tiles (1M points: 16 384 rows)
  const int64_t by_sample = (((int64_t)1 << 30) / f.ldS) / 1024 * 1024;
  if (by_sample >= 1024 && f.qc > by_sample) f.qc = by_sample;
  if (by_sample < 1024) { f.on = false; return f; }          // N beyond ~16 M points: the slab pipeline's chunks
  f.fr = n < 2048 ? n : 2048;
  return f;
}

int64_t chunk_rows(int64_t N, int64_t n, int d) {
  const int64_t ld = mgp_cdiv(N, 4) * 4;
  // fp32 distance slab: 2 GiB (8 192 query rows at N = 60k: 8 query tiles per XCD share a point tile in L2, and half
  // the chunks / host polls of the 1 GiB slab -- measured 3 % faster), up to 8 GiB for large N so that a chunk still
  // holds >= 2048 query rows and the per-chunk launch + host poll is amortised (288 GB of HBM)
  int64_t slab_floats = (int64_t)1 << 29;
  if (ld * 2048 > slab_floats) slab_floats = ld * 2048;
  if (slab_floats > ((int64_t)1 << 31)) slab_floats = (int64_t)1 << 31;
  int64_t qc = slab_floats / ld;
  qc = qc / kTile * kTile;
  // whole rounds of 8 XCDs x 128-row query tiles, the same number for every XCD (knn_mfma.hip deals query tiles to
  // XCDs: 34 tiles are 2 x 5 + 6 x 4 and the launch takes as long as the XCDs with 5)
  if (qc >= 2048) qc = qc / 2048 * 2048;
  if (qc < kTile) qc = kTile;
  const int64_t ncap = mgp_cdiv(n, kTile) * kTile;
  // As many queries as points and the WHOLE key matrix within 32 GiB (N <= 92k; the card has 288 GB): one chunk.  The
  // key kernel walks the query tiles in groups by itself (knn_mfma.hip), and when the queries ARE the points -- the graph
  // build -- it computes only the tile pairs on and above the diagonal, which needs every row of the matrix in place.
  // Only when that symmetric path can run at all (matrix-core keys on, d >= 32): a search with as many OTHER queries as
  // points also sizes for it (the workspace query does not see the pointers), any other search keeps the 2-8 GiB slab.
  if (n == N && g_knn_mfma && g_knn_sym && d >= 32 && (int64_t)ld * ncap <= ((int64_t)1 << 33)) return ncap;
  return qc < ncap ? qc : ncap;
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

}  // namespace

static size_t slab_bytes(int64_t N, int64_t n, int d, int k) {
  if (N <= 0 || n <= 0 || d <= 0 || k <= 0) return 0;
  const int64_t ld = mgp_cdiv(N, 4) * 4;
  const int64_t qc = chunk_rows(N, n, d);
  size_t b = mgp_align((size_t)qc * ld * sizeof(float));
  b += 2 * mgp_align((size_t)qc * sizeof(int));
  b += mgp_align(64);
  b += mgp_align((size_t)kExactBatch * N * sizeof(double));
  // matrix-core keys: the query side of a chunk whenever d allows them (a search through a prepared index takes them at
  // any number of queries), the point side only when this search has to prepare the points itself
  if (d >= 32) b += mgp_knn_mfma_query_bytes(qc, d);
  if (use_mfma(d, n)) b += mgp_knn_mfma_index_bytes(N, d);
  return b + 1024;
}

static size_t filter_bytes(int64_t N, int64_t n, int d, int k, const FilterPlan& f) {
  size_t b = mgp_align(slab_bytes(N, f.fr, d, k));                       // fail-over batches
  b += mgp_align(mgp_knn_mfma_index_bytes(N, d) + 256);                  // the points' split (a caller's index is used instead)
  b += mgp_knn_mfma_sample_bytes(f.S, d) + mgp_knn_mfma_query_bytes(f.qc, d);
  b += mgp_align((size_t)f.qc * f.ldS * sizeof(float));                  // keys to the sampled points
  b += 4 * mgp_align((size_t)f.qc * sizeof(int)) + mgp_align(64);        // bounds, counters, two fail lists
  b += mgp_align((size_t)f.qc * kListCap * sizeof(uint2));               // candidate lists
  b += mgp_align((size_t)f.qc * kLogPerRow * sizeof(uint2));             // the key pass's log
  b += mgp_align(mgp_knn_mfma_table_entries(f.qc, N) * 16) + mgp_align((size_t)mgp_knn_mfma_log_shards() * 64 + 64);
  b += mgp_align((size_t)f.fr * d * sizeof(float)) + 2 * mgp_align((size_t)f.fr * k * sizeof(float));
  return b + 1024;
}

static size_t bruteforce_bytes(int64_t N, int64_t n, int d, int k) {
  if (N <= 0 || n <= 0 || d <= 0 || k <= 0) return 0;
  // the workspace query does not see whether a prepared index comes with the search: a forced filter (mode 2) sizes for it
  const FilterPlan f = filter_plan(N, n, d, k, true);
  const size_t sb = slab_bytes(N, n, d, k);
  if (!f.on) return sb;
  const size_t fb = filter_bytes(N, n, d, k, f);
  // a search the plan turns out not to serve (no index, few queries) still runs on the slab
  return (g_knn_filter == 2 && !use_mfma(d, n)) ? (fb > sb ? fb : sb) : fb;
}

// the slab pipeline (any d): distance tiles -> radix select -> fp64 re-rank -> sufficiency check
static int filtered_search(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D, int32_t* I, void* work,
                           size_t work_bytes, int64_t* stats, void* stream, const void* index, size_t index_bytes,
                           const FilterPlan& fp);

int mgp_knn_bruteforce(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D,
                       int32_t* I, void* work, size_t work_bytes, int64_t* stats, void* stream, const void* index,
                       size_t index_bytes, bool allow_filter) {
  if (!db || !q || !D || !I || !work) return MGP_ERR_ARG;
  if (N <= 0 || n <= 0 || d <= 0 || k <= 0 || k > N || k > 1024 || N > INT_MAX) return MGP_ERR_ARG;
  if (allow_filter) {
    const bool idx_ok = index != nullptr && g_knn_mfma && d >= 32;
    const FilterPlan fp = filter_plan(N, n, d, k, idx_ok);
    if (fp.on) {
      if (work_bytes < filter_bytes(N, n, d, k, fp)) return MGP_ERR_WORKSPACE;
      return filtered_search(db, N, d, q, n, k, D, I, work, work_bytes, stats, stream, idx_ok ? index : nullptr, index_bytes, fp);
    }
    g_last_filter_failover = -1;
  }
  if (work_bytes < slab_bytes(N, n, d, k)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  const int64_t ld = mgp_cdiv(N, 4) * 4;
  const int64_t qc = chunk_rows(N, n, d);
  MgpArena ar(work, work_bytes);
  float* slab = ar.take<float>((size_t)qc * ld);
  int* list_a = ar.take<int>(qc);
  int* list_b = ar.take<int>(qc);
  int* counter = ar.take<int>(16);
  double* scratch = ar.take<double>((size_t)kExactBatch * N);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  // a prepared index (mgp_knn_index_build: the points' column means, bf16 split and norms) makes the matrix-core keys pay
  // at any number of queries; without one the ~1 ms of preparing 60k x 784 points pays from ~1000 queries on
  const bool indexed = index != nullptr && g_knn_mfma && d >= 32;
  if (indexed && index_bytes < mgp_knn_mfma_index_bytes(N, d)) return MGP_ERR_WORKSPACE;
  const bool mfma = indexed || use_mfma(d, n);
  // self-search with the whole key matrix in one chunk: key(x, y) = key(y, x), half the tiles are computed
  const bool sym = mfma && g_knn_sym && q == db && n == N && qc >= n;
  MgpKnnMfma mm{};
  double alpha = 0.0, beta = 0.0;
  if (d >= 32) MGP_TRY(mgp_knn_mfma_query_take(ar, qc, d, &mm));
  if (indexed) {
    MgpArena ia(const_cast<void*>(index), index_bytes);
    MGP_TRY(mgp_knn_mfma_index_take(ia, N, d, &mm));
    mgp_knn_mfma_bound(mm.dpad, &alpha, &beta);
  } else if (mfma) {
    MGP_TRY(mgp_knn_mfma_index_take(ar, N, d, &mm));
    MGP_TRY(mgp_knn_mfma_prepare_points(db, N, d, mm, st));
    mgp_knn_mfma_bound(mm.dpad, &alpha, &beta);
  }

  // first attempt: k + pad candidates (each costs a d-float row read in the fp64 re-rank: the selection's
  // dominant traffic at large d), sorted in the next power of two; retries use the full sort width
  // (MFMA keys carry an absolute error: a wider pad there keeps the retry launches rare)
  int cand0 = first_candidates(k, mfma);
  int Kp0 = next_pow2(cand0);
  if (Kp0 < 64) Kp0 = 64;
  if (Kp0 > kMaxKp) Kp0 = kMaxKp;
  if (cand0 > Kp0) cand0 = Kp0;
  // fp32 direct-difference distance: one rounding per subtraction, one per fused accumulate
  const double gamma = (double)(d + 4) * 1.1920928955078125e-07;   // (d+4) * 2^-23
  int64_t n_wide = 0, n_exact = 0, n_chunks = 0, n_direct = 0;
  int streak_direct = 0;

  for (int64_t q0 = 0; q0 < n; q0 += qc) {
    const int64_t rows = (n - q0) < qc ? (n - q0) : qc;
    ++n_chunks;
    dim3 grid((unsigned)mgp_cdiv(N, kTile), (unsigned)mgp_cdiv(rows, kTile));
    SelectArgs a{slab, ld, N, db, q + q0 * d, d, k, Kp0, cand0, nullptr, D + q0 * k, I + q0 * k, list_a, counter, gamma,
                 nullptr, nullptr, 0.0, 0.0, nullptr, nullptr, 0};
    int fails = 0;
    // pass 0: MFMA keys + absolute check; pass 1 (no MFMA, or too many rows of the chunk failed the
    // absolute check): exact fp32 direct-difference keys + relative check
    // (two chunks in a row redone: this data does not suit the absolute bound, stop paying for MFMA passes)
    for (int pass = (mfma && streak_direct < 2) ? 0 : 1; pass < 2; ++pass) {
      if (pass == 0) {
        if (!sym) MGP_TRY(mgp_knn_mfma_prepare_queries(q + q0 * d, rows, d, mm, st));
        MGP_TRY(mgp_knn_mfma_tiles(mm, rows, N, slab, ld, st, sym));
        a.qn2 = sym ? mm.pn2 : mm.qn2; a.r2max = mm.r2max; a.alpha = alpha; a.beta = beta;
      } else {
        if (d % 4 == 0) hipLaunchKernelGGL(dist_tile_kernel<true>, grid, dim3(kBlock), 0, st, db, N, d, q + q0 * d, rows, slab, ld);
        else hipLaunchKernelGGL(dist_tile_kernel<false>, grid, dim3(kBlock), 0, st, db, N, d, q + q0 * d, rows, slab, ld);
        MGP_LAUNCH_CHECK();
        a.qn2 = nullptr; a.r2max = nullptr;
      }
      MGP_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), st));
      hipLaunchKernelGGL(select_kernel, dim3((unsigned)rows), dim3(kBlock), (size_t)a.Kp * 12, st, a);
      MGP_LAUNCH_CHECK();
      MGP_HIP_TRY(hipMemcpyAsync(&fails, counter, sizeof(int), hipMemcpyDeviceToHost, st));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      if (pass == 0 && fails > rows / 16 + 8) { ++n_direct; ++streak_direct; continue; }
      if (pass == 0) streak_direct = 0;
      break;
    }
    int Kp = Kp0;
    int* cur = list_a;
    int* nxt = list_b;
    while (fails > 0 && Kp < kMaxKp && Kp < N) {
      Kp = Kp * 4 > kMaxKp ? kMaxKp : Kp * 4;
      n_wide += fails;
      SelectArgs b = a;
      b.Kp = Kp; b.cand = 0; b.rows = cur; b.fail_list = nxt;
      MGP_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), st));
      hipLaunchKernelGGL(select_kernel, dim3((unsigned)fails), dim3(kBlock), (size_t)b.Kp * 12, st, b);
      MGP_LAUNCH_CHECK();
      MGP_HIP_TRY(hipMemcpyAsync(&fails, counter, sizeof(int), hipMemcpyDeviceToHost, st));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      int* t = cur; cur = nxt; nxt = t;
    }
    if (fails > 0) {
      n_exact += fails;
      for (int f0 = 0; f0 < fails; f0 += kExactBatch) {
        const int nb = (fails - f0) < kExactBatch ? (fails - f0) : kExactBatch;
        hipLaunchKernelGGL(exact_row_kernel, dim3(nb), dim3(kBlock), 0, st, db, N, d, q + q0 * d, cur + f0, k,
                           scratch, D + q0 * k, I + q0 * k);
        MGP_LAUNCH_CHECK();
      }
    }
  }
  MGP_HIP_TRY(hipStreamSynchronize(st));
  if (stats) { stats[0] = n_wide; stats[1] = n_exact; stats[2] = n_chunks; stats[3] = cand0; }
  g_last_direct_chunks = n_direct;
  return MGP_OK;
}

// The filtered pipeline per chunk of query rows: keys to the sampled points -> per-row bounds -> ALL keys, the ones under
// the bounds appended to the rows' lists -> select + fp64 re-rank + sufficiency check from the lists (the same kernel, the
// same T: a list holds every key of its row up to a bound >= T) -> rows that fail (check, overflowing or short list) are
// gathered and redone by the slab pipeline above, which ends in the exact fp64 scan.  The RESULT is the oracle's.
static int filtered_search(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D, int32_t* I, void* work,
                           size_t work_bytes, int64_t* stats, void* stream, const void* index, size_t index_bytes,
                           const FilterPlan& fp) {
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  const size_t fb_bytes = slab_bytes(N, fp.fr, d, k);
  void* fb_work = ar.take<char>(fb_bytes);
  const size_t own_idx_bytes = mgp_knn_mfma_index_bytes(N, d) + 256;
  void* own_idx = ar.take<char>(own_idx_bytes);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  if (index && index_bytes < mgp_knn_mfma_index_bytes(N, d)) return MGP_ERR_WORKSPACE;
  const void* idx = index ? index : own_idx;
  const size_t idx_bytes = index ? index_bytes : own_idx_bytes;
  MgpKnnMfma mm{};
  {
    MgpArena ia(const_cast<void*>(idx), idx_bytes);
    MGP_TRY(mgp_knn_mfma_index_take(ia, N, d, &mm));
  }
  if (!index) MGP_TRY(mgp_knn_mfma_prepare_points(db, N, d, mm, st));
  MGP_TRY(mgp_knn_mfma_sample_take(ar, fp.S, d, &mm));
  MGP_TRY(mgp_knn_mfma_query_take(ar, fp.qc, d, &mm));
  float* samp = ar.take<float>((size_t)fp.qc * fp.ldS);
  float* bounds = ar.take<float>(fp.qc);
  int* cnt = ar.take<int>(fp.qc);
  int* list_a = ar.take<int>(fp.qc);
  int* list_b = ar.take<int>(fp.qc);
  int* counter = ar.take<int>(16);
  uint2* lists = ar.take<uint2>((size_t)fp.qc * kListCap);
  uint2* klog = ar.take<uint2>((size_t)fp.qc * kLogPerRow);
  uint4* table = ar.take<uint4>(mgp_knn_mfma_table_entries(fp.qc, N));
  const int shards = mgp_knn_mfma_log_shards();
  unsigned* cursor = ar.take<unsigned>((size_t)shards * 16 + 16);       // + the overflow flag
  int* overflow = reinterpret_cast<int*>(cursor + (size_t)shards * 16);
  float* qsub = ar.take<float>((size_t)fp.fr * d);
  float* Dsub = ar.take<float>((size_t)fp.fr * k);
  int32_t* Isub = ar.take<int32_t>((size_t)fp.fr * k);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  MGP_TRY(mgp_knn_mfma_prepare_sample(db, fp.S, fp.stride, d, mm, st));
  double alpha = 0.0, beta = 0.0;
  mgp_knn_mfma_bound(mm.dpad, &alpha, &beta);
  const bool sym = g_knn_sym && q == db && n == N && fp.qc >= n;
  int cand0 = first_candidates(k, true);
  int Kp0 = next_pow2(cand0);
  if (Kp0 < 64) Kp0 = 64;
  if (Kp0 > kMaxKp) Kp0 = kMaxKp;
  if (cand0 > Kp0) cand0 = Kp0;
  int64_t n_wide = 0, n_exact = 0, n_chunks = 0, n_failover = 0;
  for (int64_t q0 = 0; q0 < n; q0 += fp.qc) {
    const int64_t rows = (n - q0) < fp.qc ? (n - q0) : fp.qc;
    ++n_chunks;
    if (!sym) MGP_TRY(mgp_knn_mfma_prepare_queries(q + q0 * d, rows, d, mm, st));
    MGP_TRY(mgp_knn_mfma_sample_tiles(mm, rows, fp.S, samp, fp.ldS, st, sym));
    if (fp.S <= kBoundRegs * 64)
      hipLaunchKernelGGL(bound_wave_kernel, dim3((unsigned)mgp_cdiv(rows, kBlock / 64)), dim3(kBlock), 0, st, samp, fp.ldS, (int)fp.S,
                         cand0, bounds, rows);
    else
      hipLaunchKernelGGL(bound_kernel, dim3((unsigned)rows), dim3(kBlock), 0, st, samp, fp.ldS, (int)fp.S, cand0, bounds);
    MGP_LAUNCH_CHECK();
    MGP_HIP_TRY(hipMemsetAsync(cursor, 0, ((size_t)shards * 16 + 16) * sizeof(unsigned), st));
    const unsigned shard_cap = (unsigned)(((size_t)rows * kLogPerRow) / shards);
    MGP_TRY(mgp_knn_mfma_tiles_filtered(mm, rows, N, bounds, klog, cursor, table, shard_cap, overflow, st, sym));
    MGP_TRY(mgp_knn_mfma_regroup(table, klog, rows, N, lists, cnt, kListCap, st, sym));
    SelectArgs a{nullptr, 0, N, db, q + q0 * d, d, k, Kp0, cand0, nullptr, D + q0 * k, I + q0 * k, list_a, counter, 0.0,
                 sym ? mm.pn2 : mm.qn2, mm.r2max, alpha, beta, lists, cnt, kListCap};
    int fails = 0;
    MGP_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), st));
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)rows), dim3(kBlock), (size_t)a.Kp * 12, st, a);
    MGP_LAUNCH_CHECK();
    int over = 0;
    MGP_HIP_TRY(hipMemcpyAsync(&fails, counter, sizeof(int), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipMemcpyAsync(&over, overflow, sizeof(int), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    int Kp = Kp0;
    int* cur = list_a;
    int* nxt = list_b;
    if (over) {
      // the log filled up (far more keys under the bounds than stride x K' per row): entries were dropped, every row of
      // the chunk is redone on the slab
      std::vector<int> all((size_t)rows);
      for (int64_t i = 0; i < rows; ++i) all[(size_t)i] = (int)i;
      MGP_HIP_TRY(hipMemcpyAsync(cur, all.data(), (size_t)rows * sizeof(int), hipMemcpyHostToDevice, st));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      fails = (int)rows;
      Kp = kMaxKp;
    }
    // wider candidate sets from the same lists while they can hold them (a list has ~stride x K' keys)
    while (fails > 0 && Kp < kMaxKp && Kp * 4 <= fp.stride * cand0) {
      Kp = Kp * 4 > kMaxKp ? kMaxKp : Kp * 4;
      n_wide += fails;
      SelectArgs b = a;
      b.Kp = Kp; b.cand = 0; b.rows = cur; b.fail_list = nxt;
      MGP_HIP_TRY(hipMemsetAsync(counter, 0, sizeof(int), st));
      hipLaunchKernelGGL(select_kernel, dim3((unsigned)fails), dim3(kBlock), (size_t)b.Kp * 12, st, b);
      MGP_LAUNCH_CHECK();
      MGP_HIP_TRY(hipMemcpyAsync(&fails, counter, sizeof(int), hipMemcpyDeviceToHost, st));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      int* t = cur; cur = nxt; nxt = t;
    }
    n_failover += fails;
    for (int f0 = 0; f0 < fails; f0 += (int)fp.fr) {
      const int64_t m = (fails - f0) < fp.fr ? (fails - f0) : fp.fr;
      MGP_TRY(mgp_knn_gather_rows(q + q0 * d, cur + f0, m, d, qsub, stream));
      int64_t s4[4] = {0, 0, 0, 0};
      MGP_TRY(mgp_knn_bruteforce(db, N, d, qsub, m, k, Dsub, Isub, fb_work, fb_bytes, s4, stream, idx, idx_bytes, false));
      MGP_TRY(mgp_knn_scatter_rows(Dsub, Isub, cur + f0, m, k, D + q0 * k, I + q0 * k, stream));
      MGP_HIP_TRY(hipStreamSynchronize(st));
      n_wide += s4[0]; n_exact += s4[1];
    }
  }
  MGP_HIP_TRY(hipStreamSynchronize(st));
  if (stats) { stats[0] = n_wide + n_failover; stats[1] = n_exact; stats[2] = n_chunks; stats[3] = cand0; }
  g_last_direct_chunks = 0;
  g_last_filter_failover = n_failover;
  return MGP_OK;
}

// candidate filter of the matrix-core searches: 0 off (key slab), 1 default (large searches), 2 whenever it can serve
extern "C" int mgp_knn_set_filter(int mode) {
  if (mode < 0 || mode > 2) return MGP_ERR_ARG;
  g_knn_filter = mode;
  return MGP_OK;
}

// rows of the last search that the filtered pipeline handed to the slab pipeline; -1: the search ran on the slab
extern "C" int64_t mgp_knn_last_filter_failover(void) { return g_last_filter_failover; }

#ifdef MGP_SEL_LAB
extern "C" int mgp_sel_lab_stamps(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sel_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sel_stamp), z, sizeof(z)) != hipSuccess) return 1; }
  return MGP_OK;
}
#endif

extern "C" int mgp_knn_set_symmetric(int on) {
  g_knn_sym = on ? 1 : 0;
  return MGP_OK;
}

extern "C" int mgp_knn_set_mfma(int on) {
  g_knn_mfma = on ? 1 : 0;
  return MGP_OK;
}

// chunks of the last slab search that were redone with direct-difference tiles (diagnostic)
extern "C" int64_t mgp_knn_last_direct_chunks(void) { return g_last_direct_chunks; }

// ---------------------------------------------------------------------------------------------------
// Public entry: d <= 3 and N >= 4096 take the slab-free path of knn_lowd.hip; its (rare) overflow rows,
// and every other shape, go through the slab pipeline above.
namespace {

int64_t fallback_rows(int64_t n) {          // rows the low-d path may hand back per round
  // after the low-d path's own retry with tightened bounds a handful of rows is left (4 of 1M on the random
  // swiss roll): a small slab (512 rows x N keys) keeps the workspace -- and its first allocation -- small
  return n < 512 ? n : 512;
}

}  // namespace

// Prepared index: what every search against the same points would otherwise recompute (column means, centred bf16
// split in the key kernel's tile layout, norms, R^2).  A snapshot of db at build time, as faiss's add() copies.
extern "C" size_t mgp_knn_index_bytes(int64_t N, int d) {
  if (N <= 0 || d < 32) return 0;
  return mgp_knn_mfma_index_bytes(N, d) + 256;
}

extern "C" int mgp_knn_index_build(const float* db, int64_t N, int d, void* index, size_t index_bytes, void* stream) {
  if (!db || !index || N <= 0 || d < 32 || N > INT_MAX) return MGP_ERR_ARG;
  if (index_bytes < mgp_knn_index_bytes(N, d)) return MGP_ERR_WORKSPACE;
  MgpArena ia(index, index_bytes);
  MgpKnnMfma mm{};
  MGP_TRY(mgp_knn_mfma_index_take(ia, N, d, &mm));
  return mgp_knn_mfma_prepare_points(db, N, d, mm, mgp_stream(stream));
}

extern "C" int mgp_knn_search_indexed(const float* db, int64_t N, int d, const void* index, size_t index_bytes, const float* q,
                                      int64_t n, int k, float* D, int32_t* I, void* work, size_t work_bytes, int64_t* stats,
                                      void* stream) {
  if (!index) return mgp_knn_search(db, N, d, q, n, k, D, I, work, work_bytes, stats, stream);
  if (!db || !q || !D || !I || !work) return MGP_ERR_ARG;
  if (N <= 0 || n <= 0 || d < 32 || k <= 0 || k > N || k > 1024 || N > INT_MAX) return MGP_ERR_ARG;
  if (index_bytes < mgp_knn_index_bytes(N, d)) return MGP_ERR_WORKSPACE;
  if (work_bytes < mgp_knn_workspace_bytes(N, n, d, k)) return MGP_ERR_WORKSPACE;
  return mgp_knn_bruteforce(db, N, d, q, n, k, D, I, work, work_bytes, stats, stream, index, index_bytes);
}

extern "C" size_t mgp_knn_workspace_bytes(int64_t N, int64_t n, int d, int k) {
  if (N <= 0 || n <= 0 || d <= 0 || k <= 0) return 0;
  if (!mgp_knn_lowd_eligible(N, n, d, k)) return bruteforce_bytes(N, n, d, k);
  const int64_t fr = fallback_rows(n);
  size_t b = bruteforce_bytes(N, fr, d, k) + mgp_knn_lowd_workspace_bytes(N, n, d, k);
  b += mgp_align((size_t)fr * d * sizeof(float)) + 2 * mgp_align((size_t)fr * k * sizeof(float)) +
       mgp_align((size_t)fr * sizeof(int32_t));
  return b + 1024;
}

extern "C" int mgp_knn_search(const float* db, int64_t N, int d, const float* q, int64_t n, int k, float* D,
                              int32_t* I, void* work, size_t work_bytes, int64_t* stats, void* stream) {
  if (!db || !q || !D || !I || !work) return MGP_ERR_ARG;
  if (N <= 0 || n <= 0 || d <= 0 || k <= 0 || k > N || k > 1024 || N > INT_MAX) return MGP_ERR_ARG;
  if (work_bytes < mgp_knn_workspace_bytes(N, n, d, k)) return MGP_ERR_WORKSPACE;
  if (!mgp_knn_lowd_eligible(N, n, d, k)) return mgp_knn_bruteforce(db, N, d, q, n, k, D, I, work, work_bytes, stats, stream);

  hipStream_t st = mgp_stream(stream);
  const int64_t fr = fallback_rows(n);
  MgpArena ar(work, work_bytes);
  const size_t bf_bytes = bruteforce_bytes(N, fr, d, k);
  void* bf_work = ar.take<char>(bf_bytes);
  const size_t ld_bytes = mgp_knn_lowd_workspace_bytes(N, n, d, k);
  void* ld_work = ar.take<char>(ld_bytes);
  float* qsub = ar.take<float>((size_t)fr * d);
  float* Dsub = ar.take<float>((size_t)fr * k);
  int32_t* Isub = ar.take<int32_t>((size_t)fr * k);
  int32_t* rows_dev = ar.take<int32_t>(fr);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;

  std::vector<int32_t> over;
  MGP_TRY(mgp_knn_lowd(db, N, d, q, n, k, D, I, ld_work, ld_bytes, &over, stream));
  int64_t st_wide = 0, st_exact = 0, st_chunks = 0;
  for (size_t o0 = 0; o0 < over.size(); o0 += (size_t)fr) {
    const int64_t m = (int64_t)std::min<size_t>((size_t)fr, over.size() - o0);
    MGP_HIP_TRY(hipMemcpyAsync(rows_dev, over.data() + o0, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MGP_TRY(mgp_knn_gather_rows(q, rows_dev, m, d, qsub, stream));
    int64_t s4[4] = {0, 0, 0, 0};
    MGP_TRY(mgp_knn_bruteforce(db, N, d, qsub, m, k, Dsub, Isub, bf_work, bf_bytes, s4, stream));
    MGP_TRY(mgp_knn_scatter_rows(Dsub, Isub, rows_dev, m, k, D, I, stream));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    st_wide += s4[0]; st_exact += s4[1]; st_chunks += s4[2];
  }
  // stats: [0] rows redone by the slab pipeline (low-d overflow + its own wide retries), [1] exact rows,
  // [2] slab chunks, [3] -1 marks the low-d path
  if (stats) { stats[0] = (int64_t)over.size() + st_wide; stats[1] = st_exact; stats[2] = st_chunks; stats[3] = -1; }
  return MGP_OK;
}
