// Candidate distances for the exact k-NN on the matrix cores (d >= 32).
//
// The slab pipeline of knn.hip only needs the distance tile to RANK candidates: the K' smallest keys of a
// row are re-evaluated in fp64 in the oracle's operation order and a sufficiency check proves that no
// unselected point can enter the top k.  So the O(N^2 d) part does not have to be the exact fp32
// direct-difference form (2 VALU instructions per pair-feature, 86 TF): here it is the GEMM form
//     key(x, y) = max(|c_x|^2 + |c_y|^2 - 2 c_x . c_y, 0),      c = x - mean(db)   (centred),
// with the dot product on v_mfma_f32_32x32x16_bf16 through a two-term bf16 split c = h + l (+ eps):
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
constexpr int kRowE = 40;    // LDS row pitch in bf16 elements (32 + 8 pad: 80 B, conflict-free ds_read_b128)

typedef __bf16 knn_bf16x8 __attribute__((ext_vector_type(8)));
typedef float knn_f32x16 __attribute__((ext_vector_type(16)));

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

__global__ __launch_bounds__(kBlock) void colmean_kernel(const float* __restrict__ partial, int nblk, int d, int64_t n,
                                                         float* __restrict__ mu, unsigned* __restrict__ r2max) {
  const int j = blockIdx.x * kBlock + threadIdx.x;
  if (j == 0) *r2max = 0u;
  if (j >= d) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partial[(int64_t)b * d + j];
  mu[j] = (float)(s / (double)n);
}

// ------------------------------------------------------------------ centre + two-term bf16 split + norms
__device__ __forceinline__ unsigned bf16_rne(float v) {
  const unsigned b = __float_as_uint(v);
  return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16;
}

__global__ __launch_bounds__(kBlock) void split_kernel(const float* __restrict__ x, int64_t n, int d, int dpad,
                                                       const float* __restrict__ mu, uint16_t* __restrict__ H,
                                                       uint16_t* __restrict__ L, float* __restrict__ norm2,
                                                       unsigned* __restrict__ r2max) {
  __shared__ double red[kBlock / MGP_WAVE];
  const int64_t row = blockIdx.x;
  const float* xr = x + row * d;
  double s = 0.0;
  for (int j = threadIdx.x; j < dpad; j += kBlock) {
    unsigned h = 0, l = 0;
    if (j < d) {
      const float c = xr[j] - mu[j];
      h = bf16_rne(c);
      l = bf16_rne(c - __uint_as_float(h << 16));
      s += (double)c * (double)c;
    }
    H[row * dpad + j] = (uint16_t)h;
    L[row * dpad + j] = (uint16_t)l;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kBlock / MGP_WAVE; ++w) t += red[w];
    const float nf = (float)t;
    norm2[row] = nf;
    if (r2max) atomicMax(r2max, __float_as_uint(nf));   // non-negative floats order as their bit patterns
  }
}

// ------------------------------------------------------------------ distance tiles on MFMA
// 128 x 128 tile per workgroup, 4 waves as 2 x 2, each wave 2 x 2 accumulators of 32 x 32.  The four
// operand tiles (query h / l, point h / l) of a 32-feature stage sit in LDS in rows of 80 bytes; the next
// stage is fetched into registers while the current one is multiplied.  Fragment maps of
// v_mfma_f32_32x32x16_bf16: lane (r = l & 31, g = l >> 5) holds A[row r][k = 8 g + 0..7] and
// B[k = 8 g + 0..7][col r]; D[col = l & 31][row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)].
__global__ __launch_bounds__(kBlock) void dist_mfma_kernel(const uint16_t* __restrict__ Qh, const uint16_t* __restrict__ Ql,
                                                           const float* __restrict__ qn2, int64_t nq,
                                                           const uint16_t* __restrict__ Ph, const uint16_t* __restrict__ Pl,
                                                           const float* __restrict__ pn2, int64_t N, int dpad,
                                                           float* __restrict__ out, int64_t ld, int ny_per_xcd) {
  __shared__ __attribute__((aligned(16))) uint16_t sm[4][kMT][kRowE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // blocks are dealt round-robin over the 8 XCDs: XCD c takes the query tiles c, c + 8, ... and walks the
  // point tiles with its query tiles innermost, so that a point tile fetched into that XCD's L2 is used
  // by all its query tiles at about the same time and the few query tiles stay L2-resident.
  const int xcd = blockIdx.x % MGP_NXCD;
  const int t = blockIdx.x / MGP_NXCD;
  const int ty = t % ny_per_xcd, tx = t / ny_per_xcd;
  const int64_t q0 = (int64_t)(xcd + MGP_NXCD * ty) * kMT, p0 = (int64_t)tx * kMT;
  if (q0 >= nq) return;
  knn_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // staging: 16-byte piece f -> (row = f / 4, c = f % 4); 2 pieces per lane per operand tile
  const int f0 = tid, f1 = tid + kBlock;
  const int64_t sq0 = (q0 + (f0 >> 2) < nq ? q0 + (f0 >> 2) : nq - 1) * dpad + 8 * (f0 & 3);
  const int64_t sq1 = (q0 + (f1 >> 2) < nq ? q0 + (f1 >> 2) : nq - 1) * dpad + 8 * (f1 & 3);
  const int64_t sp0 = (p0 + (f0 >> 2) < N ? p0 + (f0 >> 2) : N - 1) * dpad + 8 * (f0 & 3);
  const int64_t sp1 = (p0 + (f1 >> 2) < N ? p0 + (f1 >> 2) : N - 1) * dpad + 8 * (f1 & 3);
  uint4 qh0, qh1, ql0, ql1, ph0, ph1, pl0, pl1;
#define MGP_KNN_FETCH(k0)                                            \
  do {                                                               \
    qh0 = *reinterpret_cast<const uint4*>(Qh + sq0 + (k0));           \
    qh1 = *reinterpret_cast<const uint4*>(Qh + sq1 + (k0));           \
    ql0 = *reinterpret_cast<const uint4*>(Ql + sq0 + (k0));           \
    ql1 = *reinterpret_cast<const uint4*>(Ql + sq1 + (k0));           \
    ph0 = *reinterpret_cast<const uint4*>(Ph + sp0 + (k0));           \
    ph1 = *reinterpret_cast<const uint4*>(Ph + sp1 + (k0));           \
    pl0 = *reinterpret_cast<const uint4*>(Pl + sp0 + (k0));           \
    pl1 = *reinterpret_cast<const uint4*>(Pl + sp1 + (k0));           \
  } while (0)
  MGP_KNN_FETCH(0);
  const int r = lane & 31, g8 = (lane >> 5) * 8;
  for (int k0 = 0; k0 < dpad; k0 += kBK) {
    *reinterpret_cast<uint4*>(&sm[0][f0 >> 2][8 * (f0 & 3)]) = qh0;
    *reinterpret_cast<uint4*>(&sm[0][f1 >> 2][8 * (f1 & 3)]) = qh1;
    *reinterpret_cast<uint4*>(&sm[1][f0 >> 2][8 * (f0 & 3)]) = ql0;
    *reinterpret_cast<uint4*>(&sm[1][f1 >> 2][8 * (f1 & 3)]) = ql1;
    *reinterpret_cast<uint4*>(&sm[2][f0 >> 2][8 * (f0 & 3)]) = ph0;
    *reinterpret_cast<uint4*>(&sm[2][f1 >> 2][8 * (f1 & 3)]) = ph1;
    *reinterpret_cast<uint4*>(&sm[3][f0 >> 2][8 * (f0 & 3)]) = pl0;
    *reinterpret_cast<uint4*>(&sm[3][f1 >> 2][8 * (f1 & 3)]) = pl1;
    __syncthreads();
    if (k0 + kBK < dpad) MGP_KNN_FETCH(k0 + kBK);
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 16) {
      knn_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[0][wm * 64 + i * 32 + r][kk + g8]);
        al[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[1][wm * 64 + i * 32 + r][kk + g8]);
        bh[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[2][wn * 64 + i * 32 + r][kk + g8]);
        bl[i] = *reinterpret_cast<const knn_bf16x8*>(&sm[3][wn * 64 + i * 32 + r][kk + g8]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // key = max(|c_x|^2 + |c_y|^2 - 2 S, +0): 32 consecutive points per half wave and register.  The
  // query norms go through LDS (a global load per element would be 64 dependent round trips).
  float* qn_s = reinterpret_cast<float*>(&sm[0][0][0]);
  if (tid < kMT) qn_s[tid] = qn2[q0 + tid < nq ? q0 + tid : nq - 1];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t pc = p0 + wn * 64 + j * 32 + r;
    const float pn = pn2[pc < N ? pc : N - 1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const float v = (qn_s[rl] + pn) - 2.f * acc[i][j][e];
        if (q0 + rl < nq && pc < N) out[(q0 + rl) * ld + pc] = v > 0.f ? v : 0.f;
      }
  }
}

#undef MGP_KNN_FETCH

constexpr int kMaxPartialBlocks = 256;

}  // namespace

int mgp_knn_mfma_dpad(int d) { return (int)(mgp_cdiv(d, kBK) * kBK); }

size_t mgp_knn_mfma_bytes(int64_t N, int64_t qc, int d) {
  const int dpad = mgp_knn_mfma_dpad(d);
  size_t b = 0;
  b += 2 * mgp_align((size_t)N * dpad * sizeof(uint16_t));
  b += 2 * mgp_align((size_t)qc * dpad * sizeof(uint16_t));
  b += mgp_align((size_t)N * sizeof(float)) + mgp_align((size_t)qc * sizeof(float));
  b += mgp_align((size_t)d * sizeof(float)) + mgp_align((size_t)kMaxPartialBlocks * d * sizeof(float)) + mgp_align(64);
  return b;
}

int mgp_knn_mfma_take(MgpArena& ar, int64_t N, int64_t qc, int d, MgpKnnMfma* m) {
  const int dpad = mgp_knn_mfma_dpad(d);
  m->dpad = dpad;
  m->Ph = ar.take<uint16_t>((size_t)N * dpad);
  m->Pl = ar.take<uint16_t>((size_t)N * dpad);
  m->Qh = ar.take<uint16_t>((size_t)qc * dpad);
  m->Ql = ar.take<uint16_t>((size_t)qc * dpad);
  m->pn2 = ar.take<float>(N);
  m->qn2 = ar.take<float>(qc);
  m->mu = ar.take<float>(d);
  m->partial = ar.take<float>((size_t)kMaxPartialBlocks * d);
  m->r2max = ar.take<unsigned>(16);
  return ar.ok() ? MGP_OK : MGP_ERR_WORKSPACE;
}

// mean of the points, their split and norms, R^2 = max |c_y|^2
int mgp_knn_mfma_prepare_points(const float* db, int64_t N, int d, const MgpKnnMfma& m, hipStream_t st) {
  int nblk = (int)mgp_cdiv(N, 256);
  if (nblk > kMaxPartialBlocks) nblk = kMaxPartialBlocks;
  const int64_t rpb = mgp_cdiv(N, nblk);
  nblk = (int)mgp_cdiv(N, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(kBlock), 0, st, db, N, d, rpb, m.partial);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(colmean_kernel, dim3((unsigned)mgp_cdiv(d, kBlock)), dim3(kBlock), 0, st, m.partial, nblk, d, N, m.mu,
                     m.r2max);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)N), dim3(kBlock), 0, st, db, N, d, m.dpad, m.mu, m.Ph, m.Pl, m.pn2, m.r2max);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int mgp_knn_mfma_prepare_queries(const float* q, int64_t rows, int d, const MgpKnnMfma& m, hipStream_t st) {
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)rows), dim3(kBlock), 0, st, q, rows, d, m.dpad, m.mu, m.Qh, m.Ql, m.qn2,
                     (unsigned*)nullptr);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int mgp_knn_mfma_tiles(const MgpKnnMfma& m, int64_t rows, int64_t N, float* slab, int64_t ld, hipStream_t st) {
  const int64_t nx = mgp_cdiv(N, kMT), ny = mgp_cdiv(rows, kMT);
  const int ny_per_xcd = (int)mgp_cdiv(ny, MGP_NXCD);
  const int64_t blocks = (int64_t)MGP_NXCD * ny_per_xcd * nx;
  if (blocks > 0x7fffffff) return MGP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dist_mfma_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, m.Qh, m.Ql, m.qn2, rows, m.Ph, m.Pl, m.pn2,
                     N, m.dpad, slab, ld, ny_per_xcd);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// coefficients of E(x) = alpha |c_x| R + beta (|c_x| + R)^2 (header)
void mgp_knn_mfma_bound(int dpad, double* alpha, double* beta) {
  const double a0 = 3.01 * ldexp(1.0, -18) + 3.012 * (double)dpad * ldexp(1.0, -23);
  *alpha = 1.5 * 2.0 * a0;
  *beta = 1.5 * ldexp(1.0, -21);
}
