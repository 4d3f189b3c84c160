// Candidate distances for the exact k-NN on the matrix cores (d >= 32).
//
// The slab pipeline of knn.hip only needs the distance tile to RANK candidates: the K' smallest keys of a
// row are re-evaluated in fp64 in the oracle's operation order and a sufficiency check proves that no
// unselected point can enter the top k.  So the O(N^2 d) part does not have to be the exact fp32
// direct-difference form (2 VALU instructions per pair-feature, 86 TF): here it is the GEMM form
//     key(x, y) = max(|c_x|^2 + |c_y|^2 - 2 c_x . c_y, 0),      c = x - mean(db)   (centred),
// with the dot product on v_mfma_f32_16x16x32_bf16 through a two-term bf16 split c = h + l (+ eps):
// c_x . c_y ~ h_x.h_y + h_x.l_y + l_x.h_y (3 MFMAs, fp32 accumulate, products exact in fp32).
//
// Error bound used by the sufficiency check (select_kernel, absolute form), per query row x with
// R = max_y |c_y|:
//   split        |h - c| <= 2^-9 |c|, |eps| <= 2^-18 |c|  =>  dropped terms <= 3.01 * 2^-18 |c_x||c_y|
//   accumulate   3 dpad products, any summation order, unit roundoff taken as 2^-23 (twice RNE)
//                                                         =>  <= 3.012 dpad 2^-23 |c_x||c_y|
//   norms + the two fp32 operations of the key            =>  <= 2^-22 (|c_x|^2 + |c_y|^2)
//   centring     c = fl(x - mu) per coordinate            =>  <= 1.01 * 2^-23 (|c_x| + |c_y|)^2
//   E(x) = 1.5 * [ 2 (3.01 * 2^-18 + 3.012 dpad 2^-23) |c_x| R + 2^-21 (|c_x| + R)^2 ]
// A row passes when d64[k-1] + 2 E < T (T = K'-th smallest key); rows that do not are widened, and a chunk
// with many such rows (data whose spread is tiny against its distance from the mean of a few far outliers,
// ...) is redone with the exact direct-difference tile kernel.  The RESULT is the oracle's either way.
#include <math.h>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMT = 128;     // queries x points per workgroup
constexpr int kBK = 32;      // features per LDS stage

typedef __bf16 knn_bf16x8 __attribute__((ext_vector_type(8)));
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ column means (deterministic two-pass)
__global__ __launch_bounds__(kBlock) void colsum_partial_kernel(const float* __restrict__ x, int64_t n, int d,
                                                                int64_t rows_per_block, float* __restrict__ partial) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int j = threadIdx.x; j < d; j += kBlock) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += x[r * d + j];
    partial[(int64_t)blockIdx.x * d + j] = s;
  }
}

// 64 columns per workgroup, 4 groups of partial rows each (fixed combination order: deterministic)
__global__ __launch_bounds__(kBlock) void colmean_kernel(const float* __restrict__ partial, int nblk, int d, int64_t n,
                                                         float* __restrict__ mu, unsigned* __restrict__ r2max) {
  __shared__ double sh[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  if (blockIdx.x == 0 && threadIdx.x == 0) *r2max = 0u;
  double s = 0.0;
  if (j < d)
    for (int b = g; b < nblk; b += 4) s += (double)partial[(int64_t)b * d + j];
  sh[g][c] = s;
  __syncthreads();
  if (g == 0 && j < d) mu[j] = (float)(((sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c])) / (double)n);
}

// ------------------------------------------------------------------ centre + two-term bf16 split + norms
__device__ __forceinline__ unsigned bf16_rne(float v) {
  const unsigned b = __float_as_uint(v);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

// Operand layout (H and L alike): [row tile of 128][stage of 32 features][128 rows][32 bf16], i.e. the 8 KB
// one workgroup stages per operand and stage are contiguous (the kernel copies them global -> LDS with
// 16-byte global_load_lds, no registers, no ds_write), and the four 16-byte pieces of a row are stored
// XOR-swizzled so that an MFMA fragment read (lane (r, g) reads piece g of row r) is free of bank conflicts.  Round 5: the
// swizzle follows the lane groups a ds_read_b128 is really served in -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
// + 32 (MI355X_MICROARCH.md, LDS): a group holds all 16 rows, rows 0-3 and 12-15 with piece g and rows 4-11 with piece g ^ 1,
// and the four rows of a 64-byte bank quarter (r, r + 4, r + 8, r + 12) must land on four different pieces:
// piece ^ ((0 - row / 4) & 3) does that.  The swizzle of rounds 1-4, piece ^ ((row / 4) & 3), assumed 16 consecutive lanes
// per group: every fragment read took twice its cycles (SQ_LDS_BANK_CONFLICT 49 % of SQ_LDS_IDX_ACTIVE).
// Rows past n (up to the next multiple of 128) are zero.
__host__ __device__ __forceinline__ int knn_swz(int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ int64_t tiled_index(int64_t row, int k, int nst) {
  const int rl = (int)(row & (kMT - 1));
  const int kl = k & (kBK - 1);
  return (((row / kMT) * nst + k / kBK) * kMT + rl) * kBK + ((((kl >> 3) ^ knn_swz(rl)) << 3) | (kl & 7));
}

// One wave per row, 4 consecutive rows (same tile) per workgroup: the 4 x 64 bytes they write per stage are
// contiguous -- whole 128-byte lines (one row per workgroup wrote half lines: 0.69 ms for 60k x 784).
// stride > 1: output row i is source row i * stride (the sampled points of the candidate filter, knn.hip).
__global__ __launch_bounds__(kBlock) void split_kernel(const float* __restrict__ x, int64_t n, int d, int dpad,
                                                       const float* __restrict__ mu, uint16_t* __restrict__ H,
                                                       uint16_t* __restrict__ L, float* __restrict__ norm2, int64_t stride) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nst = dpad / kBK;
  if (row >= n) {                         // padding rows of the last tile
    for (int j = lane; j < dpad; j += 64) {
      const int64_t o = tiled_index(row, j, nst);
      H[o] = 0; L[o] = 0;
    }
    return;
  }
  const float* xr = x + row * stride * d;
  double s = 0.0;
  // a lane converts one 16-byte piece (8 features) at a time: two 16-byte stores instead of sixteen 2-byte ones
  const int rl = (int)(row & (kMT - 1));
  const int64_t tile_base = (row / kMT) * nst * (int64_t)(kMT * kBK) + (int64_t)rl * kBK;
  for (int p = lane; p < dpad / 8; p += 64) {
    unsigned hh[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 8 * p + e;
      unsigned h = 0, l = 0;
      if (j < d) {
        const float c = xr[j] - mu[j];
        h = bf16_rne(c);
        l = bf16_rne(c - __uint_as_float(h << 16));
        s += (double)c * (double)c;
      }
      hh[e] = h; ll[e] = l;
    }
    const int64_t o = tile_base + (int64_t)(p >> 2) * (kMT * kBK) + (((p & 3) ^ knn_swz(rl)) << 3);
    *reinterpret_cast<uint4*>(H + o) = make_uint4(hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16), hh[4] | (hh[5] << 16), hh[6] | (hh[7] << 16));
    *reinterpret_cast<uint4*>(L + o) = make_uint4(ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16), ll[4] | (ll[5] << 16), ll[6] | (ll[7] << 16));
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    const float nf = (float)s;
    norm2[row] = nf;
  }
}

// R^2 = max_y |c_y|^2 (one workgroup; 60k same-address atomics from the split kernel cost 0.6 ms)
__global__ __launch_bounds__(1024) void r2max_kernel(const float* __restrict__ norm2, int64_t n, unsigned* __restrict__ r2max) {
  __shared__ float sh[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, norm2[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
    *r2max = __float_as_uint(m);          // non-negative floats order as their bit patterns
  }
}

// ------------------------------------------------------------------ distance tiles on MFMA
// 128 x 128 tile per workgroup, 4 waves as 2 x 2, each wave 4 x 4 accumulators of 16 x 16 (K = 32 per
// instruction = one stage; measured 7 % faster here than 2 x 2 tiles of v_mfma_f32_32x32x16_bf16).  The four
// operand tiles (query h / l, point h / l; 8 KB each) of a 32-feature stage are copied global -> LDS by
// global_load_lds (16 bytes per lane, LDS image = memory image) into one of two LDS buffers while the
// other one is multiplied: one barrier per stage, no staging registers.  Fragment maps of
// v_mfma_f32_16x16x32_bf16: lane (r = l & 15, g = l >> 4) holds A[row r][k = 8 g + 0..7] and
// B[k = 8 g + 0..7][col r]; D[col = l & 15][row = 4 (l >> 4) + reg].
//
// FILTER (round 5): no key slab.  Every row x has a bound B(x) >= its K'-th smallest key (the K'-th smallest of its keys
// to a 1/stride sample of the points, knn.hip); the epilogue keeps the keys <= B(x) -- ~stride K' of a row's N -- as
// entries {key bits, (row in the wave's 64) << 6 | (column in the wave's 64)} of an append-only LOG: a workgroup draws
// ONE range of it for all its survivors (one returning atomic per workgroup on one of 64 cursors, each with its own
// region of the log) and every wave writes its entries side by side -- the DIRECT ones (key of query row x to point y,
// for x's list) and, sym and off the diagonal, the MIRRORED ones (the same key, for y's list) -- and records
// {offset, direct count, mirrored count} in a table indexed by (tile pair, wave).  regroup_kernel (below) then deals a
// half tile's entries to its 64 rows' candidate lists with LDS counters.
// (First version of this round: slots drawn per ROW from global counters, two 64-lane returning atomics per wave and tile
// pair: 8.5 ms per 60k x 784 pass against 3.8 ms for the MFMA loop alone -- the atomics execute at the memory side and
// 56 M of them per launch slowed every other request of the launch; tools/lab/knn_filter_bounds.sh.)
#ifdef MGP_KNN_LAB
__device__ unsigned long long g_knn_stamp[8];
#define KNN_STAMP(K) do { if (MGP_KNN_LAB & 8) { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0 && blockIdx.x % 61 == 0) atomicAdd(&g_knn_stamp[K], t_ - t_prev); t_prev = __builtin_readcyclecounter(); } } while (0)
#else
#define KNN_STAMP(K) do { } while (0)
#endif
constexpr int kLogShards = 64;
struct KnnFilterArgs {
  const float* bq;       // [nq] bounds of the query rows (compare v <= b)
  const float* bp;       // [N] bounds of the point rows (sym only: the same array)
  uint2* log;            // [kLogShards * shard_cap] entries
  unsigned* cursor;      // [kLogShards * 16] fill of each shard's region (one word per 64-byte line; zeroed by the caller)
  uint4* table;          // [query tiles * nx * 4 waves] {offset, direct count, mirrored count, 0}
  unsigned shard_cap;    // entries per shard region
  int* overflow;         // set when a region is full: the caller redoes the chunk on the slab
};

template <int CTRL>
__device__ __forceinline__ unsigned knn_dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}

// lane s of the caller's 16-lane row to all 16 (row_newbcast:s); s is a constant after unrolling, the switch folds
__device__ __forceinline__ int knn_row_bcast(int v, int s) {
  switch (s) {
#define KNN_BC(S) case S: return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xf, 0xf, false);
    KNN_BC(0) KNN_BC(1) KNN_BC(2) KNN_BC(3) KNN_BC(4) KNN_BC(5) KNN_BC(6) KNN_BC(7)
    KNN_BC(8) KNN_BC(9) KNN_BC(10) KNN_BC(11) KNN_BC(12) KNN_BC(13) KNN_BC(14)
#undef KNN_BC
    default: return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xf, 0xf, false);
  }
}

template <bool FILTER>
__global__ __launch_bounds__(kBlock) void dist_mfma_kernel(const uint16_t* __restrict__ Qh, const uint16_t* __restrict__ Ql,
                                                           const float* __restrict__ qn2, int64_t nq,
                                                           const uint16_t* __restrict__ Ph, const uint16_t* __restrict__ Pl,
                                                           const float* __restrict__ pn2, int64_t N, int dpad,
                                                           float* __restrict__ out, int64_t ld, int ny_per_xcd, int nx,
                                                           int sym, KnnFilterArgs fa) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int kTileE = kMT * kBK;                 // bf16 elements of one operand tile (8 KB)
  __shared__ __attribute__((aligned(16))) uint16_t sm[2][4][kTileE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#ifdef MGP_KNN_LAB
  unsigned long long t_prev = __builtin_readcyclecounter();
#endif
  // blocks are dealt round-robin over the 8 XCDs: XCD c takes the query tiles c, c + 8, ... and walks the
  // point tiles with its query tiles innermost, so that a point tile fetched into that XCD's L2 is used
  // by all its query tiles at about the same time and the few query tiles stay L2-resident.
  // A launch over more than 8 * ny_per_xcd query tiles walks them in GROUPS of that many, one group after the other
  // (blocks are dispatched in index order): what a chunk per launch did, so that a group's query tiles stay L2-resident.
  // sym (the queries ARE the points: graph build): key(x, y) = key(y, x) -- the same products, and the bound of the
  // header holds for any summation order -- so only the tile pairs on and above the diagonal are computed and every
  // off-diagonal tile is stored twice, as it is and transposed: half the MFMA work for the same slab.
  const int per_group = MGP_NXCD * ny_per_xcd * nx;
  const int grp = blockIdx.x / per_group;
  const int bl = blockIdx.x - grp * per_group;
  const int xcd = bl % MGP_NXCD;
  const int t = bl / MGP_NXCD;
  const int ty = t % ny_per_xcd, tx = t / ny_per_xcd;
  const int64_t qt = (int64_t)grp * MGP_NXCD * ny_per_xcd + xcd + MGP_NXCD * ty;
  const int64_t q0 = qt * kMT, p0 = (int64_t)tx * kMT;
  if (q0 >= nq) return;
  if (sym && tx < qt) return;
  knn_f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = knn_f32x4{0.f, 0.f, 0.f, 0.f};

  const int nst = dpad / kBK;
  const uint16_t* gq_h = Qh + qt * nst * kTileE;
  const uint16_t* gq_l = Ql + qt * nst * kTileE;
  const uint16_t* gp_h = Ph + (int64_t)tx * nst * kTileE;
  const uint16_t* gp_l = Pl + (int64_t)tx * nst * kTileE;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // one stage: 8 copies of 4 KB (256 lanes x 16 bytes); LDS destination = wave-uniform base + lane * 16.  piece p of a stage:
  // operand p & 3 (query h / l, point h / l), half p >> 2
  auto piece = [&](int st, int buf, int p) {
    const int64_t so = (int64_t)st * kTileE;
    const int eo = (p >> 2) * (kTileE / 2) + wave * 512;          // this wave's 1 KB slice (elements)
    const uint16_t* src = (p & 3) == 0 ? gq_h : (p & 3) == 1 ? gq_l : (p & 3) == 2 ? gp_h : gp_l;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + so + eo + lane * 8), (lptr_t)&sm[buf][p & 3][eo], 16, 0, 0);
  };
  auto issue = [&](int st, int buf) {
#pragma unroll
    for (int p = 0; p < 8; ++p) piece(st, buf, p);
  };
  const int r = lane & 15, g = lane >> 4;
  // fragment (row, piece g) of a tile: row * 32 + ((g ^ knn_swz(row)) * 8); row = 16-aligned base + r
  const int pc = (g ^ knn_swz(r)) * 8;
  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this stage's copies have landed
    __syncthreads();                          // ... for every wave; the other buffer is free again
    const int buf = st & 1;
    const bool more = st + 1 < nst;
    if (more) issue(st + 1, (st + 1) & 1);
    knn_bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][0][(wm * 64 + i * 16 + r) * kBK + pc]);
      al[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][1][(wm * 64 + i * 16 + r) * kBK + pc]);
      bh[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][2][(wn * 64 + i * 16 + r) * kBK + pc]);
      bl[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][3][(wn * 64 + i * 16 + r) * kBK + pc]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  __syncthreads();
  KNN_STAMP(0);      // main loop
  // key = max(|c_x|^2 + |c_y|^2 - 2 S, +0): 16 consecutive points per 16 lanes and register.  The
  // query norms go through LDS (a global load per element would be 64 dependent round trips).
  float* qn_s = reinterpret_cast<float*>(&sm[0][0][0]);
  if constexpr (FILTER) {
#ifdef MGP_KNN_LAB
    if (MGP_KNN_LAB & 4) return;
#endif
    // rows / columns past the edge get an infinite norm (their keys pass no bound) and the bound -1
    float* bq_s = qn_s + kMT;
    if (tid < kMT) {
      const bool in = q0 + tid < nq;
      qn_s[tid] = in ? qn2[q0 + tid] : INFINITY;
      bq_s[tid] = in ? fa.bq[q0 + tid] : -1.f;
    }
    __syncthreads();
    float pn[4], bp[4];
    knn_f32x4 qn[4], bq[4];
    const bool mirror = sym && tx != qt;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t pcol = p0 + wn * 64 + j * 16 + r;
      pn[j] = pcol < N ? pn2[pcol] : INFINITY;
      bp[j] = (mirror && pcol < N) ? fa.bp[pcol] : -1.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qn[i] = *reinterpret_cast<const knn_f32x4*>(&qn_s[wm * 64 + i * 16 + g * 4]);
      bq[i] = *reinterpret_cast<const knn_f32x4*>(&bq_s[wm * 64 + i * 16 + g * 4]);
    }
    // keys in place; predicate bits: direct and mirrored alike (i, e, j) -> bit 16 i + 4 e + j
    unsigned pd[2] = {0u, 0u}, pm[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = (qn[i][e] + pn[j]) - 2.f * acc[i][j][e];
          v = v > 0.f ? v : 0.f;
          acc[i][j][e] = v;
          pd[i >> 1] |= (v <= bq[i][e] ? 1u : 0u) << (16 * (i & 1) + 4 * e + j);
          pm[i >> 1] |= (v <= bp[j] ? 1u : 0u) << (16 * (i & 1) + 4 * e + j);
        }
    KNN_STAMP(1);    // bounds, keys, predicates
    // this lane's entries, prefix over the wave (DPP: 16-lane rows, then row_bcast15 / row_bcast31), totals over the workgroup
    const unsigned nd = __popc(pd[0]) + __popc(pd[1]), nm = __popc(pm[0]) + __popc(pm[1]);
    unsigned x = nd | (nm << 16);                  // a wave holds at most 4096 entries of each kind
    x += knn_dpp<0x111>(x);
    x += knn_dpp<0x112>(x);
    x += knn_dpp<0x114>(x);
    x += knn_dpp<0x118>(x);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    const unsigned wt = (unsigned)__builtin_amdgcn_readlane((int)x, 63);
    unsigned* ws = reinterpret_cast<unsigned*>(bq_s + kMT);      // [4] wave totals, [4] base
    if (lane == 0) ws[wave] = wt;
    __syncthreads();
    KNN_STAMP(2);    // scans
    if (tid == 0) {
      const unsigned t0 = ws[0], t1 = ws[1], t2 = ws[2], t3 = ws[3];
      const unsigned n0 = (t0 & 0xffffu) + (t0 >> 16), n1 = (t1 & 0xffffu) + (t1 >> 16), n2 = (t2 & 0xffffu) + (t2 >> 16),
                     n3 = (t3 & 0xffffu) + (t3 >> 16);
      const unsigned total = n0 + n1 + n2 + n3;
      // a multiplicative hash of the block index: blockIdx % 64 is the QUERY TILE within its group, whose tile pairs (and
      // entries) a self-search deals unevenly (tile 0 has every pair, the last one a single pair)
      const unsigned shard = (blockIdx.x * 2654435761u) >> 26;
      static_assert(kLogShards == 64, "top 6 bits of the hash");
      unsigned base = 0u;
      bool ok = true;
#ifdef MGP_KNN_LAB
      if (!(MGP_KNN_LAB & 1))
#endif
      if (total) {
        base = atomicAdd(fa.cursor + shard * 16, total);
        if (base + total > fa.shard_cap) { ok = false; *fa.overflow = 1; }
      }
      base += shard * fa.shard_cap;
      uint4* trow = fa.table + ((int64_t)qt * nx + tx) * 4;
      const unsigned m = ok ? 0xffffffffu : 0u;
      trow[0] = make_uint4(base, (t0 & 0xffffu) & m, (t0 >> 16) & m, 0u);
      trow[1] = make_uint4(base + n0, (t1 & 0xffffu) & m, (t1 >> 16) & m, 0u);
      trow[2] = make_uint4(base + n0 + n1, (t2 & 0xffffu) & m, (t2 >> 16) & m, 0u);
      trow[3] = make_uint4(base + n0 + n1 + n2, (t3 & 0xffffu) & m, (t3 >> 16) & m, 0u);
      ws[4] = base; ws[5] = base + n0; ws[6] = base + n0 + n1; ws[7] = base + n0 + n1 + n2;
      ws[8] = ok ? 1u : 0u;
    }
    __syncthreads();
    KNN_STAMP(3);    // the workgroup's range drawn
    if (!ws[8]) return;
#ifdef MGP_KNN_LAB
    if (MGP_KNN_LAB & 2) return;
#endif
    const unsigned ex = x - (nd | (nm << 16));
    uint2* od = fa.log + ws[4 + wave] + (ex & 0xffffu);
    uint2* om = fa.log + ws[4 + wave] + (wt & 0xffffu) + (ex >> 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned bd = (pd[i >> 1] >> (16 * (i & 1))) & 0xffffu, bm = (pm[i >> 1] >> (16 * (i & 1))) & 0xffffu;
      if (bd | bm) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint2 ent = make_uint2(__float_as_uint(acc[i][j][e]), (unsigned)(((i * 16 + g * 4 + e) << 6) | (j * 16 + r)));
            if (bd & (1u << (4 * e + j))) *od++ = ent;
            if (bm & (1u << (4 * e + j))) *om++ = ent;
          }
      }
    }
    KNN_STAMP(5);    // stores issued
#ifdef MGP_KNN_LAB
    if (MGP_KNN_LAB & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); KNN_STAMP(7); }   // stores acknowledged
#endif
    return;
  }
  if (tid < kMT) qn_s[tid] = qn2[q0 + tid < nq ? q0 + tid : nq - 1];
  __syncthreads();
  // every load first (4 point norms, 16 query norms as 4 x 16 bytes from LDS), then the 64 stores back to back: on
  // gfx9 stores count in vmcnt too, and a load between them made each group of stores wait for all earlier ones
  float pn[4];
  knn_f32x4 qn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t pcol = p0 + wn * 64 + j * 16 + r;
    pn[j] = pn2[pcol < N ? pcol : N - 1];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) qn[i] = *reinterpret_cast<const knn_f32x4*>(&qn_s[wm * 64 + i * 16 + g * 4]);
  // ... and waited for HERE: the stores sit under exec masks (tile edges), behind such a branch the compiler cannot
  // count how many stores follow a load and falls back to vmcnt(0) in front of every use
  __builtin_amdgcn_s_waitcnt(0x0070);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int rl = wm * 64 + i * 16 + g * 4 + e;
      float* orow = out + (q0 + rl) * ld + p0 + wn * 64 + r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (qn[i][e] + pn[j]) - 2.f * acc[i][j][e];
        if (q0 + rl < nq && p0 + wn * 64 + j * 16 + r < N) orow[j * 16] = v > 0.f ? v : 0.f;
      }
    }
  if (sym && tx != qt) {
    // the transposed tile: rows = this tile's points, columns = its queries; a lane's four consecutive rows (e) of
    // one column become four consecutive floats of one row: a 16-byte store (q0, the 16-row blocks and ld are multiples
    // of 4; the tile edge falls back to single floats)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t prow = p0 + wn * 64 + j * 16 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t qc0 = q0 + wm * 64 + i * 16 + g * 4;
        knn_f32x4 v4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = (qn[i][e] + pn[j]) - 2.f * acc[i][j][e];
          v4[e] = v > 0.f ? v : 0.f;
        }
        float* trow = out + prow * ld + qc0;
        if (prow < N) {
          if (qc0 + 3 < nq) {
            *reinterpret_cast<knn_f32x4*>(trow) = v4;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (qc0 + e < nq) trow[e] = v4[e];
          }
        }
      }
    }
  }
#endif
}

// The log's entries of one half tile (64 rows) dealt to the rows' candidate lists {key bits, point index}: its direct
// entries sit in the segments of the tile pairs (t, tx) (sym: tx >= t), waves (b, wn); its mirrored ones in those of
// (qt < t, t), waves (wm, b).  Segments are taken 1024 at a time: their {offset, count} to LDS, a prefix over the counts, then
// the batch's entries flat over the 1024 threads (binary search of the segment); slots from 64 LDS counters, which end
// up as the rows' entry counts (a count past `cap` marks an overflowed list: the select kernel fails that row over).
constexpr int kRgThreads = 1024;
__global__ __launch_bounds__(kRgThreads) void regroup_kernel(const uint4* __restrict__ table, const uint2* __restrict__ log, int nx,
                                                             int64_t nq, int sym, uint2* __restrict__ lists,
                                                             int* __restrict__ counts, int cap) {
  __shared__ unsigned seg_off[kRgThreads], seg_pre[kRgThreads + 1];
  __shared__ int seg_base[kRgThreads];
  __shared__ int cnt[64];
  __shared__ unsigned wsum[kRgThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = blockIdx.x >> 1, b = blockIdx.x & 1;
  if (tid < 64) cnt[tid] = 0;
  const int n_dir = (sym ? nx - t : nx) * 2, n_mir = (sym ? t : 0) * 2, nseg = n_dir + n_mir;
  for (int s0 = 0; s0 < nseg; s0 += kRgThreads) {
    __syncthreads();                       // the previous batch is consumed (first round: the counters are zero)
    const int sg = s0 + tid;
    unsigned c = 0, off = 0;
    int base = 0;
    if (sg < nseg) {
      if (sg < n_dir) {
        const int tx = (sym ? t : 0) + (sg >> 1), wn = sg & 1;
        const uint4 e = table[((int64_t)t * nx + tx) * 4 + b * 2 + wn];
        off = e.x; c = e.y; base = tx * kMT + wn * 64;
      } else {
        const int m = sg - n_dir, qt = m >> 1, wm = m & 1;
        const uint4 e = table[((int64_t)qt * nx + t) * 4 + wm * 2 + b];
        off = e.x + e.y; c = e.z; base = (qt * kMT + wm * 64) | (int)0x80000000;
      }
    }
    unsigned inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = (unsigned)__shfl_up((int)inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned wb = 0, total = 0;
    for (int w = 0; w < kRgThreads / 64; ++w) { const unsigned v = wsum[w]; if (w < wave) wb += v; total += v; }
    seg_off[tid] = off; seg_base[tid] = base; seg_pre[tid] = wb + inc - c;
    if (tid == 0) seg_pre[kRgThreads] = total;
    __syncthreads();
    for (unsigned e = tid; e < total; e += kRgThreads) {
      int lo = 0, hi = kRgThreads;         // seg_pre[lo] <= e < seg_pre[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg_pre[mid] <= e) lo = mid; else hi = mid;
      }
      const uint2 ent = log[(size_t)seg_off[lo] + (e - seg_pre[lo])];
      const int sb = seg_base[lo];
      const int rl = (int)(ent.y >> 6) & 63, cl = (int)ent.y & 63;
      const int rowl = sb < 0 ? cl : rl;
      const unsigned idx = (unsigned)(sb & 0x7fffffff) + (unsigned)(sb < 0 ? rl : cl);
      const int slot = atomicAdd(&cnt[rowl], 1);
      if (slot < cap) lists[((int64_t)t * kMT + b * 64 + rowl) * cap + slot] = make_uint2(ent.x, idx);
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int64_t row = (int64_t)t * kMT + b * 64 + tid;
    if (row < nq) counts[row] = cnt[tid];
  }
}

// Tried in round 2 and removed (docs/kernels/knn.md): a 256 x 128 tile shared by eight waves with a
// three-stage LDS ring (144 KB, one workgroup per CU; copies requested two stages ahead, the next stage's fragments read
// into a second register set during the MFMAs, three separate __shared__ arrays so that the compiler's LDS-DMA alias
// check lets copies stay in flight across a stage), also as a persistent kernel.  Bit-identical keys, 131 instead of 98
// flop per byte copied -- and 1.19 ms per 4 096-row chunk against 0.94 ms for the kernel above: with one workgroup per
// CU nothing covers a stage whose point tile misses L2 (half of those requests do: a point tile is shared by only
// ny_per_xcd workgroups of an XCD at a time), and two stages of prefetch are shorter than that miss.
// Also tried on THIS kernel: a second barrier right behind the fragment reads, which frees the stage's buffer early, so
// that the copies of stage s + 2 go out before the MFMAs of stage s (two stages ahead with the same 64 KB; two arrays and
// a 2x unrolled loop for the alias check; __launch_bounds__(256, 2), without which the accumulators moved between AGPRs
// and VGPRs every stage): identical keys, 1.86 ms against 1.83 ms per 8 192-row chunk -- the wait for the copies is not
// what the remaining 41 % of the MFMA pipes' time goes to.

constexpr int kMaxPartialBlocks = 1024;

}  // namespace

int mgp_knn_mfma_dpad(int d) { return (int)(mgp_cdiv(d, kBK) * kBK); }

// The POINT side of the operands (split tiles, norms, column means, R^2): what an index keeps between searches.
size_t mgp_knn_mfma_index_bytes(int64_t N, int d) {
  const int dpad = mgp_knn_mfma_dpad(d);
  size_t b = 0;
  b += 2 * mgp_align((size_t)(mgp_cdiv(N, kMT) * kMT) * dpad * sizeof(uint16_t));
  b += mgp_align((size_t)N * sizeof(float));
  b += mgp_align((size_t)d * sizeof(float)) + mgp_align((size_t)kMaxPartialBlocks * d * sizeof(float)) + mgp_align(64);
  return b;
}

int mgp_knn_mfma_index_take(MgpArena& ar, int64_t N, int d, MgpKnnMfma* m) {
  const int dpad = mgp_knn_mfma_dpad(d);
  m->dpad = dpad;
  m->Ph = ar.take<uint16_t>((size_t)(mgp_cdiv(N, kMT) * kMT) * dpad);
  m->Pl = ar.take<uint16_t>((size_t)(mgp_cdiv(N, kMT) * kMT) * dpad);
  m->pn2 = ar.take<float>(N);
  m->mu = ar.take<float>(d);
  m->partial = ar.take<float>((size_t)kMaxPartialBlocks * d);
  m->r2max = ar.take<unsigned>(16);
  return ar.ok() ? MGP_OK : MGP_ERR_WORKSPACE;
}

// ... and the QUERY side of one chunk
size_t mgp_knn_mfma_query_bytes(int64_t qc, int d) {
  const int dpad = mgp_knn_mfma_dpad(d);
  return 2 * mgp_align((size_t)(mgp_cdiv(qc, kMT) * kMT) * dpad * sizeof(uint16_t)) + mgp_align((size_t)qc * sizeof(float));
}

int mgp_knn_mfma_query_take(MgpArena& ar, int64_t qc, int d, MgpKnnMfma* m) {
  const int dpad = mgp_knn_mfma_dpad(d);
  m->Qh = ar.take<uint16_t>((size_t)(mgp_cdiv(qc, kMT) * kMT) * dpad);
  m->Ql = ar.take<uint16_t>((size_t)(mgp_cdiv(qc, kMT) * kMT) * dpad);
  m->qn2 = ar.take<float>(qc);
  return ar.ok() ? MGP_OK : MGP_ERR_WORKSPACE;
}

size_t mgp_knn_mfma_bytes(int64_t N, int64_t qc, int d) { return mgp_knn_mfma_index_bytes(N, d) + mgp_knn_mfma_query_bytes(qc, d); }

int mgp_knn_mfma_take(MgpArena& ar, int64_t N, int64_t qc, int d, MgpKnnMfma* m) {
  MGP_TRY(mgp_knn_mfma_index_take(ar, N, d, m));
  return mgp_knn_mfma_query_take(ar, qc, d, m);
}

// mean of the points, their split and norms, R^2 = max |c_y|^2
int mgp_knn_mfma_prepare_points(const float* db, int64_t N, int d, const MgpKnnMfma& m, hipStream_t st) {
  int nblk = (int)mgp_cdiv(N, 64);
  if (nblk > kMaxPartialBlocks) nblk = kMaxPartialBlocks;
  const int64_t rpb = mgp_cdiv(N, nblk);
  nblk = (int)mgp_cdiv(N, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(kBlock), 0, st, db, N, d, rpb, m.partial);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)mgp_cdiv(d, 64)), dim3(kBlock), 0, st, m.partial, nblk, d, N, m.mu,
                     m.r2max);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)(mgp_cdiv(N, kMT) * kMT / 4)), dim3(kBlock), 0, st, db, N, d, m.dpad, m.mu, m.Ph, m.Pl, m.pn2, (int64_t)1);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(r2max_kernel, dim3(1), dim3(1024), 0, st, m.pn2, N, m.r2max);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int mgp_knn_mfma_prepare_queries(const float* q, int64_t rows, int d, const MgpKnnMfma& m, hipStream_t st) {
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)(mgp_cdiv(rows, kMT) * kMT / 4)), dim3(kBlock), 0, st, q, rows, d, m.dpad, m.mu, m.Qh, m.Ql, m.qn2, (int64_t)1);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// the sampled points of the candidate filter: every stride-th point, split with the POINTS' column means
size_t mgp_knn_mfma_sample_bytes(int64_t S, int d) {
  const int dpad = mgp_knn_mfma_dpad(d);
  return 2 * mgp_align((size_t)(mgp_cdiv(S, kMT) * kMT) * dpad * sizeof(uint16_t)) + mgp_align((size_t)S * sizeof(float));
}

int mgp_knn_mfma_sample_take(MgpArena& ar, int64_t S, int d, MgpKnnMfma* m) {
  const int dpad = mgp_knn_mfma_dpad(d);
  m->Sh = ar.take<uint16_t>((size_t)(mgp_cdiv(S, kMT) * kMT) * dpad);
  m->Sl = ar.take<uint16_t>((size_t)(mgp_cdiv(S, kMT) * kMT) * dpad);
  m->sn2 = ar.take<float>(S);
  return ar.ok() ? MGP_OK : MGP_ERR_WORKSPACE;
}

int mgp_knn_mfma_prepare_sample(const float* db, int64_t S, int64_t stride, int d, const MgpKnnMfma& m, hipStream_t st) {
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)(mgp_cdiv(S, kMT) * kMT / 4)), dim3(kBlock), 0, st, db, S, d, m.dpad, m.mu, m.Sh, m.Sl, m.sn2, stride);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

namespace {
struct TileGrid { int ny_per_xcd; int64_t blocks; int nx; };
int tile_grid(int64_t rows, int64_t N, TileGrid* t) {
  const int64_t nx = mgp_cdiv(N, kMT), ny = mgp_cdiv(rows, kMT);
  t->ny_per_xcd = ny > 8 * MGP_NXCD ? 8 : (int)mgp_cdiv(ny, MGP_NXCD);
  const int64_t groups = mgp_cdiv(ny, (int64_t)MGP_NXCD * t->ny_per_xcd);
  t->blocks = groups * MGP_NXCD * t->ny_per_xcd * nx;
  if (t->blocks > 0x7fffffff || nx > 0x7fffffff) return MGP_ERR_UNSUPPORTED;
  t->nx = (int)nx;
  return MGP_OK;
}
}  // namespace

// keys of `rows` queries (sym: the points themselves) against the S sampled points into samp[rows, ld]
int mgp_knn_mfma_sample_tiles(const MgpKnnMfma& m, int64_t rows, int64_t S, float* samp, int64_t ld, hipStream_t st, bool sym) {
  TileGrid t;
  MGP_TRY(tile_grid(rows, S, &t));
  hipLaunchKernelGGL(dist_mfma_kernel<false>, dim3((unsigned)t.blocks), dim3(kBlock), 0, st, sym ? m.Ph : m.Qh, sym ? m.Pl : m.Ql,
                     sym ? m.pn2 : m.qn2, rows, m.Sh, m.Sl, m.sn2, S, m.dpad, samp, ld, t.ny_per_xcd, t.nx, 0, KnnFilterArgs{});
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// all keys, the ones under the rows' bounds appended to the log (bounds[rows]; sym: rows == N, the bounds serve both
// sides); cursor (kLogShards x 16 words) and *overflow zeroed by the caller
size_t mgp_knn_mfma_table_entries(int64_t rows, int64_t N) { return (size_t)mgp_cdiv(rows, kMT) * (size_t)mgp_cdiv(N, kMT) * 4; }
int mgp_knn_mfma_log_shards(void) { return kLogShards; }

int mgp_knn_mfma_tiles_filtered(const MgpKnnMfma& m, int64_t rows, int64_t N, const float* bounds, void* log, unsigned* cursor,
                                void* table, unsigned shard_cap, int* overflow, hipStream_t st, bool sym) {
  TileGrid t;
  MGP_TRY(tile_grid(rows, N, &t));
  if (sym && rows != N) return MGP_ERR_ARG;
  KnnFilterArgs fa{bounds, bounds, static_cast<uint2*>(log), cursor, static_cast<uint4*>(table), shard_cap, overflow};
  hipLaunchKernelGGL(dist_mfma_kernel<true>, dim3((unsigned)t.blocks), dim3(kBlock), 0, st, sym ? m.Ph : m.Qh, sym ? m.Pl : m.Ql,
                     sym ? m.pn2 : m.qn2, rows, m.Ph, m.Pl, m.pn2, N, m.dpad, (float*)nullptr, (int64_t)0, t.ny_per_xcd, t.nx,
                     sym ? 1 : 0, fa);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// the log's entries dealt to the rows' candidate lists [rows, cap] and counts[rows]
int mgp_knn_mfma_regroup(const void* table, const void* log, int64_t rows, int64_t N, void* lists, int* counts, int cap,
                         hipStream_t st, bool sym) {
  const int64_t nx = mgp_cdiv(N, kMT), ny = mgp_cdiv(rows, kMT);
  hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)(2 * ny)), dim3(kRgThreads), 0, st, static_cast<const uint4*>(table),
                     static_cast<const uint2*>(log), (int)nx, rows, sym ? 1 : 0, static_cast<uint2*>(lists), counts, cap);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int mgp_knn_mfma_tiles(const MgpKnnMfma& m, int64_t rows, int64_t N, float* slab, int64_t ld, hipStream_t st, bool sym) {
  const int64_t nx = mgp_cdiv(N, kMT), ny = mgp_cdiv(rows, kMT);
  // at most 8 query tiles per XCD at a time (their operand tiles stay in that XCD's L2 while the point tiles stream by:
  // the shape the 8 192-row chunks were tuned for); more query tiles than that are walked group by group in one launch
  const int ny_per_xcd = ny > 8 * MGP_NXCD ? 8 : (int)mgp_cdiv(ny, MGP_NXCD);
  const int64_t groups = mgp_cdiv(ny, (int64_t)MGP_NXCD * ny_per_xcd);
  const int64_t blocks = groups * MGP_NXCD * ny_per_xcd * nx;
  if (blocks > 0x7fffffff || nx > 0x7fffffff) return MGP_ERR_UNSUPPORTED;
  if (sym && rows != N) return MGP_ERR_ARG;
  // sym: the queries are the points -- their split and norms serve both sides
  hipLaunchKernelGGL(dist_mfma_kernel<false>, dim3((unsigned)blocks), dim3(kBlock), 0, st, sym ? m.Ph : m.Qh, sym ? m.Pl : m.Ql,
                     sym ? m.pn2 : m.qn2, rows, m.Ph, m.Pl, m.pn2, N, m.dpad, slab, ld, ny_per_xcd, (int)nx, sym ? 1 : 0,
                     KnnFilterArgs{});
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

#ifdef MGP_KNN_LAB
extern "C" int mgp_knn_lab_stamps(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_knn_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stamp), z, sizeof(z)) != hipSuccess) return 1; }
  return MGP_OK;
}
#endif

// coefficients of E(x) = alpha |c_x| R + beta (|c_x| + R)^2 (header)
void mgp_knn_mfma_bound(int dpad, double* alpha, double* beta) {
  const double a0 = 3.01 * ldexp(1.0, -18) + 3.012 * (double)dpad * ldexp(1.0, -23);
  *alpha = 1.5 * 2.0 * a0;
  *beta = 1.5 * ldexp(1.0, -21);
}
