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
// XOR-swizzled by (row / 4) % 4 so that the MFMA fragment reads (16 lanes, 16 rows, the same piece) fall
// on 16 different 16-byte bank groups.  Rows past n (up to the next multiple of 128) are zero.
__device__ __forceinline__ int64_t tiled_index(int64_t row, int k, int nst) {
  const int rl = (int)(row & (kMT - 1));
  const int kl = k & (kBK - 1);
  return (((row / kMT) * nst + k / kBK) * kMT + rl) * kBK + ((((kl >> 3) ^ ((rl >> 2) & 3)) << 3) | (kl & 7));
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
    const int64_t o = tile_base + (int64_t)(p >> 2) * (kMT * kBK) + (((p & 3) ^ ((rl >> 2) & 3)) << 3);
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
// to a 1/stride sample of the points, knn.hip); the epilogue appends the keys <= B(x) -- ~stride K' of a row's N -- to
// that row's candidate list {key bits, point index}, `cap` entries per row, slots drawn from a per-row counter.  Per
// wave and tile pair that is TWO atomic instructions whose 64 lanes address 64 consecutive counters (256 contiguous
// bytes: the shape global atomics run at full rate for, they execute at the memory side): one for the wave's 64 query
// rows, one (sym, off the diagonal) for its 64 point rows, whose mirrored keys go to THEIR lists.  A row's survivors in a
// wave are counted with byte-packed counters: in-lane over the 4 column blocks, DPP row_shr scan over the 16 lanes that
// hold the row's columns (direct), in-lane over the 16 row slots and a 4-lane exchange (mirrored).  A counter may run past
// `cap`: the entries beyond it are dropped and the select kernel sends such a row to the slab pipeline.
struct KnnFilterArgs {
  const float* bq;       // [nq] bounds of the query rows (compare v <= b)
  const float* bp;       // [N] bounds of the point rows (sym only: the same array)
  int* cnt;              // [nq] list fill counters (zeroed by the caller)
  uint2* lists;          // [nq, cap] {key bits, point index}
  int cap;
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
  // one stage: 8 copies of 4 KB (256 lanes x 16 bytes); LDS destination = wave-uniform base + lane * 16
  auto issue = [&](int st, int buf) {
    const int64_t so = (int64_t)st * kTileE;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int eo = h * (kTileE / 2) + wave * 512;          // this wave's 1 KB slice (elements)
      __builtin_amdgcn_global_load_lds((gptr_t)(gq_h + so + eo + lane * 8), (lptr_t)&sm[buf][0][eo], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gq_l + so + eo + lane * 8), (lptr_t)&sm[buf][1][eo], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp_h + so + eo + lane * 8), (lptr_t)&sm[buf][2][eo], 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(gp_l + so + eo + lane * 8), (lptr_t)&sm[buf][3][eo], 16, 0, 0);
    }
  };
  const int r = lane & 15, g = lane >> 4;
  // fragment (row, piece g) of a tile: row * 32 + ((g ^ ((row >> 2) & 3)) * 8); row = 16-aligned base + r
  const int pc = (g ^ ((r >> 2) & 3)) * 8;
  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this stage's copies have landed
    __syncthreads();                          // ... for every wave; the other buffer is free again
    if (st + 1 < nst) issue(st + 1, (st + 1) & 1);
    const int buf = st & 1;
    knn_bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][0][(wm * 64 + i * 16 + r) * kBK + pc]);
      al[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][1][(wm * 64 + i * 16 + r) * kBK + pc]);
      bh[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][2][(wn * 64 + i * 16 + r) * kBK + pc]);
      bl[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[buf][3][(wn * 64 + i * 16 + r) * kBK + pc]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
  }
  __syncthreads();
  // key = max(|c_x|^2 + |c_y|^2 - 2 S, +0): 16 consecutive points per 16 lanes and register.  The
  // query norms go through LDS (a global load per element would be 64 dependent round trips).
  float* qn_s = reinterpret_cast<float*>(&sm[0][0][0]);
  if constexpr (FILTER) {
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
    // keys in place; predicate bits: direct (i, e, j) -> bit 16 i + 4 e + j, mirrored (j, i, e) -> bit 16 j + 4 i + e
    unsigned pd[2] = {0u, 0u}, pm[2] = {0u, 0u};
    unsigned cd[4] = {0u, 0u, 0u, 0u};     // cd[i], byte e: this lane's survivors of row (i, g, e) over its 4 columns
    unsigned cm = 0u;                      // byte j: this lane's survivors of point row (j, r) over its 16 query rows
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = (qn[i][e] + pn[j]) - 2.f * acc[i][j][e];
          v = v > 0.f ? v : 0.f;
          acc[i][j][e] = v;
          const unsigned a = v <= bq[i][e] ? 1u : 0u, b = v <= bp[j] ? 1u : 0u;
          pd[i >> 1] |= a << (16 * (i & 1) + 4 * e + j);
          pm[j >> 1] |= b << (16 * (j & 1) + 4 * i + e);
          cd[i] += a << (8 * e);
          cm += b << (8 * j);
        }
    // direct: inclusive scan over the 16 lanes of a row group (bytes stay under 64), lane 15's total to all of them
    unsigned inc[4], tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned x = cd[i];
      x += knn_dpp<0x111>(x);
      x += knn_dpp<0x112>(x);
      x += knn_dpp<0x114>(x);
      x += knn_dpp<0x118>(x);
      inc[i] = x;
      tot[i] = (unsigned)knn_row_bcast((int)x, 15);
    }
    // lane (g, r) draws the slots of row slot s = r = 4 i + e of its group: rows wm 64 + 16 (r >> 2) + 4 g + (r & 3),
    // the wave's 64 lanes cover its 64 consecutive rows
    const unsigned ts = (r >> 2) == 0 ? tot[0] : (r >> 2) == 1 ? tot[1] : (r >> 2) == 2 ? tot[2] : tot[3];
    const int myd = (int)((ts >> (8 * (r & 3))) & 0xffu);
    const int64_t rowd = q0 + wm * 64 + (r >> 2) * 16 + g * 4 + (r & 3);
    int based = 0, basem = 0;
    if (myd > 0) based = atomicAdd(fa.cnt + rowd, myd);
    // mirrored: the 4 lanes r, r + 16, r + 32, r + 48 hold point row (j, r); lane (g, r) draws for j = g: row wn 64 + lane
    unsigned exm = 0u;
    if (mirror) {
      const unsigned c0 = (unsigned)__shfl((int)cm, r, 64), c1 = (unsigned)__shfl((int)cm, r + 16, 64),
                     c2 = (unsigned)__shfl((int)cm, r + 32, 64), c3 = (unsigned)__shfl((int)cm, r + 48, 64);
      exm = (g > 0 ? c0 : 0u) + (g > 1 ? c1 : 0u) + (g > 2 ? c2 : 0u);
      const unsigned tm = c0 + c1 + c2 + c3;
      const int mym = (int)((tm >> (8 * g)) & 0xffu);
      if (mym > 0) basem = atomicAdd(fa.cnt + p0 + wn * 64 + lane, mym);
    }
    // direct stores
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int bs = knn_row_bcast(based, 4 * i + e);   // the slot's base from the lane that drew it
        const unsigned bits = (pd[i >> 1] >> (16 * (i & 1) + 4 * e)) & 0xfu;
        if (bits) {
          int slot = bs + (int)((inc[i] >> (8 * e)) & 0xffu) - (int)((cd[i] >> (8 * e)) & 0xffu);
          const int64_t rowg = q0 + wm * 64 + i * 16 + g * 4 + e;
          uint2* lrow = fa.lists + rowg * fa.cap;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (bits & (1u << j)) {
              if (slot < fa.cap) lrow[slot] = make_uint2(__float_as_uint(acc[i][j][e]), (unsigned)(p0 + wn * 64 + j * 16 + r));
              ++slot;
            }
        }
      }
    if (mirror) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int bs = __shfl(basem, j * 16 + r, 64);
        const unsigned bits = (pm[j >> 1] >> (16 * (j & 1))) & 0xffffu;
        if (bits) {
          int slot = bs + (int)((exm >> (8 * j)) & 0xffu);
          uint2* lrow = fa.lists + (p0 + wn * 64 + j * 16 + r) * fa.cap;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (bits & (1u << (4 * i + e))) {
                if (slot < fa.cap) lrow[slot] = make_uint2(__float_as_uint(acc[i][j][e]), (unsigned)(q0 + wm * 64 + i * 16 + g * 4 + e));
                ++slot;
              }
        }
      }
    }
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

// all keys, filtered into the rows' candidate lists (bounds[rows]; sym: rows == N, the bounds serve both sides)
int mgp_knn_mfma_tiles_filtered(const MgpKnnMfma& m, int64_t rows, int64_t N, const float* bounds, int* cnt, void* lists, int cap,
                                hipStream_t st, bool sym) {
  TileGrid t;
  MGP_TRY(tile_grid(rows, N, &t));
  if (sym && rows != N) return MGP_ERR_ARG;
  KnnFilterArgs fa{bounds, bounds, cnt, static_cast<uint2*>(lists), cap};
  hipLaunchKernelGGL(dist_mfma_kernel<true>, dim3((unsigned)t.blocks), dim3(kBlock), 0, st, sym ? m.Ph : m.Qh, sym ? m.Pl : m.Ql,
                     sym ? m.pn2 : m.qn2, rows, m.Ph, m.Pl, m.pn2, N, m.dpad, (float*)nullptr, (int64_t)0, t.ny_per_xcd, t.nx,
                     sym ? 1 : 0, fa);
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

// coefficients of E(x) = alpha |c_x| R + beta (|c_x| + R)^2 (header)
void mgp_knn_mfma_bound(int dpad, double* alpha, double* beta) {
  const double a0 = 3.01 * ldexp(1.0, -18) + 3.012 * (double)dpad * ldexp(1.0, -23);
  *alpha = 1.5 * 2.0 * a0;
  *beta = 1.5 * ldexp(1.0, -21);
}
