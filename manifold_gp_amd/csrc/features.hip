// Spectral side of the Riemann kernel: eigenvector post-processing, in-sample and out-of-sample
// features, the dense kernel block K = Z1 Z2^T on the fp32 MFMA, its diagonal, and the low-rank
// covariance apply  y = alpha Z (Z^T x) + beta x.
//
// Reference: manifold_gp/kernels/riemann_kernel.py:79-149, riemann_matern_kernel.py:21-22,
// manifold_gp/operators/graph_laplacian_operator.py:146-157, manifold_gp/utils/torch_utils.py:38-41.
#include <math.h>
#include <type_traits>
#include <atomic>
#include "mgp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxModes = 1024;
constexpr int kNormChunks = 256;

// ---------------------------------------------------------------- eval() post-processing
// riemann_kernel.py:127-128: eigvec *= D^-1/2 (row scaling), then F.normalize(p=2, dim=0)
__global__ void scale_rows_partial_norm(float* __restrict__ V, int64_t n, int m, const float* __restrict__ deg,
                                        float* __restrict__ partial, int64_t rows_per_block) {
  // thread layout: TC lanes over columns, TS slices over rows
  int TC = 1;
  while (TC < m && TC < kBlock) TC <<= 1;
  const int TS = kBlock / TC;
  const int cc = threadIdx.x % TC, sl = threadIdx.x / TC;
  __shared__ float sh[kBlock];
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  for (int c0 = 0; c0 < m; c0 += TC) {
    const int c = c0 + cc;
    float acc = 0.f;
    if (c < m) {
      for (int64_t r = r0 + sl; r < r1; r += TS) {
        const float v = V[r * m + c] * (1.0f / sqrtf(deg[r]));
        V[r * m + c] = v;
        acc = fmaf(v, v, acc);
      }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && c < m) {
      float t = 0.f;
      for (int s = 0; s < TS; ++s) t += sh[s * TC + cc];
      partial[(int64_t)blockIdx.x * m + c] = t;
    }
    __syncthreads();
  }
}

__global__ void finalize_colnorm(const float* __restrict__ partial, int nblk, int m, float* __restrict__ inv_norm) {
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < m; c += gridDim.x * blockDim.x) {
    float t = 0.f;
    for (int b = 0; b < nblk; ++b) t += partial[(int64_t)b * m + c];
    const float nrm = sqrtf(t);
    inv_norm[c] = 1.0f / fmaxf(nrm, 1e-12f);   // F.normalize eps
  }
}

__global__ void scale_cols(float* __restrict__ V, int64_t n, int m, const float* __restrict__ s) {
  const int64_t total = n * m;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    V[i] *= s[i % m];
}

// ---------------------------------------------------------------- spectral density in LDS
// s_j = (2 nu / kappa^2 + lambda_j)^-nu, optionally / (1 - eps^2 lambda_j)^2 (riemann_kernel.py:143),
// normalised to sum 1 and multiplied by n; out[j] = sqrt(s_j)
__device__ void sqrt_density(const float* __restrict__ evals, int m, int nu, float kappa, float eps_oos, float nscale,
                             float* sh_s /*[m]*/, float* sh_tmp /*[kBlock]*/) {
  const float tau = 2.0f * (float)nu / (kappa * kappa);
  float local = 0.f;
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    float s = powf(tau + evals[j], -(float)nu);
    if (eps_oos > 0.f) {
      const float t = 1.0f - eps_oos * eps_oos * evals[j];
      s = s / (t * t);
    }
    sh_s[j] = s;
    local += s;
  }
  sh_tmp[threadIdx.x] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)blockDim.x; ++i) t += sh_tmp[i];
    sh_tmp[0] = t;
  }
  __syncthreads();
  const float tot = sh_tmp[0];
  for (int j = threadIdx.x; j < m; j += blockDim.x) sh_s[j] = sqrtf(sh_s[j] / tot * nscale);
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void features_insample_kernel(const float* __restrict__ evals,
                                                                   const float* __restrict__ V, int64_t n, int m,
                                                                   int nu, float kappa, float* __restrict__ Z) {
  __shared__ float sh_s[kMaxModes];
  __shared__ float sh_tmp[kBlock];
  sqrt_density(evals, m, nu, kappa, 0.f, (float)n, sh_s, sh_tmp);
  const int64_t total = n * m;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    Z[i] = sh_s[i % m] * V[i];
}

// ---------------------------------------------------------------- fused out-of-sample features
// graph_laplacian_operator.py:146-157 + riemann_kernel.py:138-147 + torch_utils.py:38-41, one
// workgroup per test point, no [T,k,m] temporary.
__global__ __launch_bounds__(kBlock) void features_oos_kernel(
    const float* __restrict__ evals, const float* __restrict__ V, int64_t n, int m, int nu, float kappa, float eps,
    int normalization, const float* __restrict__ dtil, const float* __restrict__ deg,
    const float* __restrict__ knn_d2, const int32_t* __restrict__ knn_idx, int k, float bump_scale,
    float bump_decay, float* __restrict__ Z) {
  __shared__ float sh_s[kMaxModes];
  __shared__ float sh_tmp[kBlock];
  __shared__ float sh_w[1024];
  __shared__ int sh_j[1024];
  const int64_t t = blockIdx.x;
  const float d1 = sqrtf(knn_d2[t * k]);
  const float alpha = bump_scale * eps;
  if (!(d1 < alpha)) {   // outside the support: features stay zero (riemann_kernel.py:139-140)
    for (int j = threadIdx.x; j < m; j += blockDim.x) Z[t * m + j] = 0.f;
    return;
  }
  sqrt_density(evals, m, nu, kappa, eps, (float)n, sh_s, sh_tmp);
  // weights
  const float nq = -4.0f * eps * eps;
  for (int l = threadIdx.x; l < k; l += blockDim.x) {
    sh_w[l] = expf(knn_d2[t * k + l] / nq);
    sh_j[l] = knn_idx[t * k + l];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float dsum = 0.f;
    for (int l = 0; l < k; ++l) dsum += sh_w[l];              // degree_test
    float s2 = 0.f;
    for (int l = 0; l < k; ++l) { sh_w[l] = sh_w[l] / (dtil[sh_j[l]] * dsum); s2 += sh_w[l]; }
    if (normalization == 0) {
      const float sq = sqrtf(s2);
      for (int l = 0; l < k; ++l) sh_w[l] = sh_w[l] / (sqrtf(deg[sh_j[l]]) * sq);
    } else {
      for (int l = 0; l < k; ++l) sh_w[l] = sh_w[l] / s2;
    }
  }
  __syncthreads();
  // bump(x; alpha, beta) = exp(beta / (x^2 - alpha^2)) / exp(-beta / alpha^2)
  const float a2 = alpha * alpha;
  const float bump = expf(bump_decay / (d1 * d1 - a2)) / expf(-bump_decay / a2);
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    float acc = 0.f;
    for (int l = 0; l < k; ++l) acc = fmaf(sh_w[l], V[(int64_t)sh_j[l] * m + j], acc);
    Z[t * m + j] = sh_s[j] * acc * bump;
  }
}

// ---------------------------------------------------------------- K = scale * Z1 Z2^T  (fp32 MFMA)
// v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kKB = 128;   // workgroup tile (rows of Z1 x rows of Z2)
constexpr int kKC = 16;    // modes per LDS stage (32 is no faster and wastes the tail stage at m = 100)

// VEC (m % 4 == 0, 16-byte aligned rows): 16-byte global loads, and the next 16-mode stage is fetched
// into registers while the MFMAs of the current one run -- with scalar staging and no prefetch every
// stage exposed a full memory round trip (~1 us against ~0.85 us of MFMA work per stage).
// X4 (n2 % 4 == 0, ldk % 4 == 0, K 16-byte aligned): the MFMA operands are swapped, so a lane's four consecutive accumulator
// registers are four consecutive COLUMNS of one row of K and go out as one 16-byte store.  A vector store costs the CU about 64
// cycles of its memory pipe whatever its width: with one dword per lane the 256 store instructions of a tile were 16k cycles
// against 13k cycles of MFMA at m = 100, and the kernel ran at the rate of its stores (1.5 TB/s of K), not of the matrix pipe.
template <bool VEC, bool X4>
__global__ __launch_bounds__(kBlock) void kernel_block_mfma(const float* __restrict__ Z1, int64_t n1,
                                                            const float* __restrict__ Z2, int64_t n2, int m,
                                                            float scale, float* __restrict__ K, int64_t ldk) {
  __shared__ float As[kKB][kKC + 1];
  __shared__ float Bs[kKB][kKC + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;              // 2 x 2 waves, 64 x 64 each
  const int64_t row0 = (int64_t)blockIdx.y * kKB, col0 = (int64_t)blockIdx.x * kKB;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // VEC staging: float4 f -> (row = f / (kKC/4), kq = f % (kKC/4)); NF float4 per lane per matrix; rows
  // past the end are clamped (their products are never stored)
  constexpr int QPR = kKC / 4, NF = kKB * QPR / kBlock;
  float4 ra[NF], rb[NF];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      const int f = tid + h * kBlock;
      const int r = f / QPR, kq = f % QPR;
      const int64_t ar = (row0 + r < n1) ? row0 + r : n1 - 1;
      const int64_t br = (col0 + r < n2) ? col0 + r : n2 - 1;
      const int kg = k0 + 4 * kq;
      if (kg < m) {
        ra[h] = *reinterpret_cast<const float4*>(Z1 + ar * m + kg);
        rb[h] = *reinterpret_cast<const float4*>(Z2 + br * m + kg);
      } else {
        ra[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        rb[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  if (VEC) fetch(0);

  for (int k0 = 0; k0 < m; k0 += kKC) {
    if (VEC) {
#pragma unroll
      for (int h = 0; h < NF; ++h) {
        const int f = tid + h * kBlock;
        const int r = f / QPR, kq = 4 * (f % QPR);
        As[r][kq + 0] = ra[h].x; As[r][kq + 1] = ra[h].y; As[r][kq + 2] = ra[h].z; As[r][kq + 3] = ra[h].w;
        Bs[r][kq + 0] = rb[h].x; Bs[r][kq + 1] = rb[h].y; Bs[r][kq + 2] = rb[h].z; Bs[r][kq + 3] = rb[h].w;
      }
    } else {
      for (int e = tid; e < kKB * kKC; e += kBlock) {
        const int r = e / kKC, kk = e % kKC;
        const int kg = k0 + kk;
        float a = 0.f, b = 0.f;
        if (kg < m) {
          if (row0 + r < n1) a = Z1[(row0 + r) * m + kg];
          if (col0 + r < n2) b = Z2[(col0 + r) * m + kg];
        }
        As[r][kk] = a;
        Bs[r][kk] = b;
      }
    }
    __syncthreads();
    if (VEC && k0 + kKC < m) fetch(k0 + kKC);          // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < kKC; kk += 2) {
      const int ksel = kk + (lane >> 5);
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[wr * 64 + i * 32 + (lane & 31)][ksel];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bs[wc * 64 + j * 32 + (lane & 31)][ksel];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = X4 ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  if (X4) {
    // transposed accumulators: K row = the MFMA column (lane & 31), K columns = 8 (r >> 2) + 4 (lane >> 5) + (r & 3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t row = row0 + wr * 64 + i * 32 + (lane & 31);
          const int64_t col = col0 + wc * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = scale * acc[i][j][4 * g + e];
          // default cache policy: a non-temporal store of PART of a line (here 32 B of each of 32 rows) runs at a fifth of the rate
          if (row < n1 && col < n2) *reinterpret_cast<f32x4*>(K + row * ldk + col) = v;
        }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int64_t col = col0 + wc * 64 + j * 32 + (lane & 31);
        // write-once output, larger than L2 at the shapes that matter: streaming (non-temporal) stores
        if (row < n1 && col < n2) __builtin_nontemporal_store(scale * acc[i][j][r], K + row * ldk + col);
      }
}

// The same product with the matrix pipe handed back and forth between TWO HALVES of one 512-thread workgroup per CU (X4 inputs
// only).  What the stage stamps, the store-rate lab (tools/lab/storebw.hip) and a run with every store dropped showed about
// kernel_block_mfma at two or three independent waves per SIMD:
//  * K leaves the chip at 5-6 TB/s whatever a store instruction's width (144 MB: 24-28 us of a 103 us launch), as long as it is
//    not a non-temporal store of PART of a line (32 rows x 32 B per instruction with nt: 1.0 TB/s): the stores are not the bound;
//  * with no store at all the launch still takes 81 us for 54 us of MFMA: a wave's 32 MFMAs of a 16-mode stage are 2048 cycles
//    of the pipe, and the pipe idles whenever both waves of a SIMD are between MFMA blocks at once (waiting for staging loads,
//    writing LDS, at their workgroup's barrier, reading the first operands).  Independent workgroups drift into exactly that:
//    measured stage cadence 5500 cycles for 2 x 2048 of MFMA.
// So the alternation is made explicit.  Each half (4 waves, one per SIMD, 64 x 64 of a 128 x 128 tile each) walks its own tiles
// t = 2 blockIdx.x + half, + 2 gridDim.x, ... and lives in two kinds of phase, a workgroup barrier after each:
//     MFMA phase     issue the staging loads of the next stage (ONE instruction each: 32-bit lane offset fixed per tile + scalar
//                    base that advances with the stage), then the stage's MFMAs from the half's LDS image;
//     service phase  after a tile's last stage: scale the accumulators and store them (16 stores of 16 bytes per lane, see X4);
//                    wait for the staging loads, write them to LDS (single buffer: the half's MFMAs are not running).
// While one half is in its MFMA phase the other is in its service phase, so each SIMD always has exactly one wave feeding the
// pipe and one wave doing everything else in its shadow.  One accumulator set, one LDS image per half.
//  * no branch around a store or a load: the last row tile and the last column tile are moved back to end at n1 / n2, writing
//    what they share with their neighbours twice with the same values.  Buffer stores: one VGPR of lane offset, the rest of the
//    address scalar or constant (host side: 512 ldk < 2^31).
//  * the ragged last stage (m not a multiple of 16) runs TS = 2, 4 or 6 two-mode steps: a template parameter, and the last
//    stage is peeled out of the stage loop, so that no two alternative MFMA sequences meet in a join (where they did, the
//    compiler copied accumulators between MFMAs).
struct KbTile {
  __amdgpu_buffer_rsrc_t rs;
  int lane_off;
};
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// q = 0..15 -> accumulator (i, j) = (q >> 3, (q >> 2) & 1), registers 4 g .. 4 g + 3 with g = q & 3: with the operands swapped
// (see kernel_block_mfma X4) these are columns j 32 + 8 g + 4 (lane >> 5) + 0..3 of row i 32 + (lane & 31) of the wave's 64 x 64
// The data registers of these stores are the accumulators themselves and nothing writes them until kb_store_done: a 16-byte
// buffer store reads its data over several cycles after issue, and with a REGISTER in the scalar-offset field the compiler does
// not keep the next VALU write of those registers away from it (its hazard rule covers only the constant-offset form): with
// scaled temporaries reused from store to store, lanes 12-15 of every 16 of the stores with a scalar row offset got the NEXT
// store's first two values.
template <int Q>
__device__ __forceinline__ void kb_store_q(const f32x16 (&acc)[2][2], const KbTile& o, int ldk4) {
  constexpr int i = Q >> 3, j = (Q >> 2) & 1, g = Q & 3;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
  // the lane's part in the vector offset, the sub-tile's rows in the scalar offset, its columns as a constant
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o.rs, o.lane_off + (j * 32 + 8 * g) * 4, i * 32 * ldk4, 0);
}

template <int Q0, int N>
__device__ __forceinline__ void kb_store_range(const f32x16 (&acc)[2][2], const KbTile& o, int ldk4) {
  if constexpr (N > 0) {
    kb_store_q<Q0>(acc, o, ldk4);
    kb_store_range<Q0 + 1, N - 1>(acc, o, ldk4);
  }
}

// between the last store and the first write of its data registers.  Where the 5 wait states come from: the ISA's
// manually-inserted-wait table (and LLVM's GCNHazardRecognizer, "VMEM store of more than 8 bytes followed by a VALU write
// of its data VGPRs": VmemStoreHazardWaitStates = 1) asks for 1 wait state, and the compiler inserts it -- but only for
// stores WITHOUT an SGPR in the soffset field (its createsVALUHazard returns "no hazard" when soffset is a register,
// as the hardware documentation words the rule).  Round 2's parity test showed the rule does not hold on gfx950 for
// this form (lanes 12-15 of every 16 took the next store's first two values).  A 16-byte store reads its four data
// registers over at most four issue cycles after the address cycle; `s_nop 4` (5 wait states) covers that with one to
// spare, and it is needed once per tile: only the LAST store is followed by a write of accumulator registers (the
// earlier ones are followed by further stores, which read other registers).
__device__ __forceinline__ void kb_store_done() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 4");
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int kKS = kKC + 1;      // LDS row stride in floats: odd, so the 32 rows a half-wave reads fall in 32 banks

// NS two-mode steps of one stage, straight-line
template <int NS, int ST = 0>
__device__ __forceinline__ void kb_steps(const float (*__restrict__ A)[kKS], const float (*__restrict__ B)[kKS], int arow, int brow,
                                         int khalf, float a0, float a1, float b0, float b1, f32x16 (&acc)[2][2]) {
  if constexpr (ST < NS) {
    const float c0 = a0, c1 = a1, d0 = b0, d1 = b1;
    if constexpr (ST + 1 < NS) {
      const int ksel = 2 * (ST + 1) + khalf;
      a0 = A[arow][ksel], a1 = A[arow + 32][ksel], b0 = B[brow][ksel], b1 = B[brow + 32][ksel];
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, c0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, c0, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, c1, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, c1, acc[1][1], 0, 0, 0);
    kb_steps<NS, ST + 1>(A, B, arow, brow, khalf, a0, a1, b0, b1, acc);
  }
}

constexpr int kPpThreads = 2 * kBlock;

template <int TS>
__global__ __launch_bounds__(kPpThreads, 1) void kernel_block_pp(const float* __restrict__ Z1, int64_t n1,
                                                                 const float* __restrict__ Z2, int64_t n2, int m, float scale,
                                                                 float* __restrict__ K, int64_t ldk, int nrt, int ntiles,
                                                                 int records) {
  __shared__ float As[2][kKB][kKS];                     // [half]
  __shared__ float Bs[2][kKB][kKS];
  // wave-uniform by construction; said so, so that the tile walk and its branches are scalar
  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);
  const int tid = threadIdx.x & 255, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;              // 2 x 2 waves per half, 64 x 64 each
  const int arow = wr * 64 + (lane & 31), brow = wc * 64 + (lane & 31), khalf = lane >> 5;
  const int ldk4 = (int)ldk * 4;
  const int nst = (m + kKC - 1) / kKC;
  const float (*A)[kKS] = As[half];
  const float (*B)[kKS] = Bs[half];

  constexpr int QPR = kKC / 4, NF = kKB * QPR / kBlock;
  f32x4 ra[NF], rb[NF];
  const int kq4 = 4 * (tid % QPR);                      // this lane's quad inside a stage (kBlock is a multiple of QPR)
  int offa[NF], offb[NF];                               // byte offsets of this lane's rows in Z1 / Z2 (host side: both < 2^31)
  auto aim = [&](int64_t row0, int64_t col0) {
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      const int r = (tid + h * kBlock) / QPR;
      offa[h] = (int)(((row0 + r) * m + kq4) * 4);
      offb[h] = (int)(((col0 + r) * m + kq4) * 4);
    }
  };
  // The staging loads are inline asm and so is their wait.  As compiler-visible loads they were followed, a few instructions
  // later, by `s_waitcnt vmcnt(1); v_mov` of two of the loaded dwords (live-range splitting: the registers were wanted for LDS
  // addresses inside the MFMA block), a full memory round trip in front of every stage.  The compiler does not know these
  // registers are in flight, so nothing may read them before kb_wait: the "+v" operands of kb_wait tie every later use to it
  // (tests/test_host_cpu.py compiles this file and checks the instruction stream for exactly that).
  // k0 past m - 16 (the ragged last stage): quads past m are aimed at the row's last quad instead; they are staged like the
  // rest and never read (that stage runs TS steps).  One select, no second code path.
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z1), (short)0, (int)(n1 * m * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z2), (short)0, (int)(n2 * m * 4), 0x00020000);
  const int back_last = ((nst - 1) * kKC + kq4 < m) ? 0 : (m - 4 - (nst - 1) * kKC - kq4) * 4;
  // buffer loads through descriptors of EXACTLY the operands' extents (n1 m and n2 m floats; host side: both < 2^29):
  // a stray staging address returns zeros instead of faulting (a GPU memory fault can reset every GPU of the host).  The
  // lane's row / quad offset is the vector offset, the stage's k0 the scalar offset.
  auto fetch = [&](int k0) {
    const int so = k0 * 4;
    const int back = (k0 + kKC > m) ? back_last : 0;
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[h]) : "v"(offa[h] + back), "s"(rs1), "s"(so));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[h]) : "v"(offb[h] + back), "s"(rs2), "s"(so));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // vmcnt counts loads and stores in issue order; the wait comes before the service phase's stores, so nothing issued after the
  // staging loads may still be in flight: 0
  auto kb_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]));
#pragma unroll
    for (int h = 1; h < NF; ++h) asm volatile("" : "+v"(ra[h]), "+v"(rb[h]));
  };
  auto stash = [&]() {
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      const int r = (tid + h * kBlock) / QPR;
      float* a = &As[half][r][kq4];
      float* b = &Bs[half][r][kq4];
      a[0] = ra[h].x; a[1] = ra[h].y; a[2] = ra[h].z; a[3] = ra[h].w;
      b[0] = rb[h].x; b[1] = rb[h].y; b[2] = rb[h].z; b[3] = rb[h].w;
    }
  };
  auto describe = [&](int64_t row0, int64_t col0) {
    KbTile o;
    float* base = K + row0 * ldk + col0;
    const uint64_t bits = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bits), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bits >> 32));
    // num_records = the bytes from the tile's first element to the end of its LAST row inside K (the tile's 128 rows x the
    // row stride, cut at column n2): a store that strays past the tile's rows is dropped by the hardware's range check
    // instead of landing in another allocation (records = 0 drops every store: the timing knob)
    const int extent = records == 0 ? 0 : (int)(((int64_t)(kKB - 1) * ldk + (n2 - col0)) * 4);
    o.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, extent, 0x00020000);
    o.lane_off = ((wr * 64 + (lane & 31)) * (int)ldk + wc * 64 + 4 * (lane >> 5)) * 4;
    return o;
  };
  auto zero = [](f32x16 (&x)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[i][j][r] = 0.f;
  };
  // the last row tile and the last column tile are moved back to end at n1 / n2 (host side: both >= 128, n2 a multiple of 4)
  auto row_of = [&](int tt) { const int64_t r = (int64_t)(tt % nrt) * kKB; return r + kKB <= n1 ? r : n1 - kKB; };
  auto col_of = [&](int tt) { const int64_t c = (int64_t)(tt / nrt) * kKB; return c + kKB <= n2 ? c : n2 - kKB; };

  // tiles of this half, and of the half with more of them (half 0): every wave passes the same number of barriers
  const int first = 2 * blockIdx.x + half, stride = 2 * gridDim.x;
  const int mine = first < ntiles ? (ntiles - first + stride - 1) / stride : 0;
  const int most = (ntiles - 2 * (int)blockIdx.x + stride - 1) / stride;     // half 0's count (blockIdx.x < ntiles / 2 rounded up)

  f32x16 acc[2][2];
  zero(acc);
  if (mine > 0) {
    aim(row_of(first), col_of(first));
    fetch(0);
    kb_wait();
    stash();
  }
  __syncthreads();
  if (half == 1) __syncthreads();          // half 1 runs one phase behind half 0

  int t = first;
  for (int it = 0; it < mine; ++it, t += stride) {
    const KbTile out = describe(row_of(t), col_of(t));
    const bool next_tile = it + 1 < mine;
    // MFMA phases of the full stages, each followed by its service phase
    for (int s = 0; s + 1 < nst; ++s) {
      fetch((s + 1) * kKC);
      kb_steps<kKC / 2>(A, B, arow, brow, khalf, A[arow][khalf], A[arow + 32][khalf], B[brow][khalf], B[brow + 32][khalf], acc);
      __syncthreads();
      kb_wait();
      stash();
      __syncthreads();
    }
    // the last stage: the next tile's first stage is what travels meanwhile, and the tile is written in the service phase
    if (next_tile) {
      aim(row_of(t + stride), col_of(t + stride));
      fetch(0);
    }
    kb_steps<TS>(A, B, arow, brow, khalf, A[arow][khalf], A[arow + 32][khalf], B[brow][khalf], B[brow + 32][khalf], acc);
    __syncthreads();
    kb_wait();                     // unconditional: no branch between a staging load and its wait (tools/check_kblock_isa.py)
    if (next_tile) stash();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] *= scale;
    __builtin_amdgcn_sched_barrier(0);
    kb_store_range<0, 16>(acc, out, ldk4);
    kb_store_done();
    zero(acc);
    __syncthreads();
  }
  // the barriers of the phases this half does not have
  const int pad = 2 * nst * (most - mine) + (half == 0 ? 1 : 0);
  for (int i = 0; i < pad; ++i) __syncthreads();
}

// One tile per workgroup with the same lean stage as kernel_block_pp (single-instruction staging loads, exact ragged last stage,
// 16-byte stores, edge tiles moved back), LDS double buffered, one barrier per stage, 107 VGPRs = four workgroups per CU.
// The default kernel wherever its conditions hold: 4096 x 60000 x 100 628 -> 493 us (78 -> 100 TFLOP/s), 8192 x 8192 x 256
// 300 -> 280 (123 TFLOP/s = 78 % of peak) against kernel_block_mfma; the 600 x 60000 x 100 posterior block stays at 105 us:
// 2345 tiles are 2.3 rounds of the 1024 resident workgroups, and each tile's first staging round trip and its final stores
// are a third of its 5.3 us of MFMA.
template <int TS>
__global__ __launch_bounds__(kBlock, 4) void kernel_block_one(const float* __restrict__ Z1, int64_t n1,
                                                              const float* __restrict__ Z2, int64_t n2, int m, float scale,
                                                              float* __restrict__ K, int64_t ldk, int nrt, int records) {
  __shared__ float As[2][kKB][kKS];                     // [buffer]
  __shared__ float Bs[2][kKB][kKS];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;              // 2 x 2 waves, 64 x 64 each
  const int arow = wr * 64 + (lane & 31), brow = wc * 64 + (lane & 31), khalf = lane >> 5;
  const int ldk4 = (int)ldk * 4;
  const int nst = (m + kKC - 1) / kKC;
  const int t = blockIdx.x;
  int64_t row0 = (int64_t)(t % nrt) * kKB, col0 = (int64_t)(t / nrt) * kKB;
  if (row0 + kKB > n1) row0 = n1 - kKB;                 // edge tiles moved back to end at n1 / n2 (host side: both >= 128)
  if (col0 + kKB > n2) col0 = n2 - kKB;

  constexpr int QPR = kKC / 4, NF = kKB * QPR / kBlock;
  f32x4 ra[NF], rb[NF];
  const int kq4 = 4 * (tid % QPR);
  int offa[NF], offb[NF];
#pragma unroll
  for (int h = 0; h < NF; ++h) {
    const int r = (tid + h * kBlock) / QPR;
    offa[h] = (int)(((row0 + r) * m + kq4) * 4);
    offb[h] = (int)(((col0 + r) * m + kq4) * 4);
  }
  // inline-asm staging loads and their wait: see kernel_block_pp
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z1), (short)0, (int)(n1 * m * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z2), (short)0, (int)(n2 * m * 4), 0x00020000);
  const int back_last = ((nst - 1) * kKC + kq4 < m) ? 0 : (m - 4 - (nst - 1) * kKC - kq4) * 4;
  // buffer loads through descriptors of EXACTLY the operands' extents (n1 m and n2 m floats; host side: both < 2^29):
  // a stray staging address returns zeros instead of faulting (a GPU memory fault can reset every GPU of the host).  The
  // lane's row / quad offset is the vector offset, the stage's k0 the scalar offset.
  auto fetch = [&](int k0) {
    const int so = k0 * 4;
    const int back = (k0 + kKC > m) ? back_last : 0;
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[h]) : "v"(offa[h] + back), "s"(rs1), "s"(so));
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[h]) : "v"(offb[h] + back), "s"(rs2), "s"(so));
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto kb_wait = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(rb[0]));
#pragma unroll
    for (int h = 1; h < NF; ++h) asm volatile("" : "+v"(ra[h]), "+v"(rb[h]));
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      const int r = (tid + h * kBlock) / QPR;
      float* a = &As[buf][r][kq4];
      float* b = &Bs[buf][r][kq4];
      a[0] = ra[h].x; a[1] = ra[h].y; a[2] = ra[h].z; a[3] = ra[h].w;
      b[0] = rb[h].x; b[1] = rb[h].y; b[2] = rb[h].z; b[3] = rb[h].w;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fetch(0);
  kb_wait();
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int s = 0; s + 1 < nst; ++s) {
    fetch((s + 1) * kKC);
    kb_steps<kKC / 2>(As[buf], Bs[buf], arow, brow, khalf, As[buf][arow][khalf], As[buf][arow + 32][khalf], Bs[buf][brow][khalf],
                      Bs[buf][brow + 32][khalf], acc);
    kb_wait();
    stash(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  kb_steps<TS>(As[buf], Bs[buf], arow, brow, khalf, As[buf][arow][khalf], As[buf][arow + 32][khalf], Bs[buf][brow][khalf],
               Bs[buf][brow + 32][khalf], acc);
  KbTile out;
  {
    float* base = K + row0 * ldk + col0;
    const uint64_t bits = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)bits), hi = __builtin_amdgcn_readfirstlane((uint32_t)(bits >> 32));
    // exact extent of the tile inside K (see kernel_block_pp::describe)
    const int extent = records == 0 ? 0 : (int)(((int64_t)(kKB - 1) * ldk + (n2 - col0)) * 4);
    out.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0, extent, 0x00020000);
    out.lane_off = ((wr * 64 + (lane & 31)) * (int)ldk + wc * 64 + 4 * (lane >> 5)) * 4;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] *= scale;
  __builtin_amdgcn_sched_barrier(0);
  kb_store_range<0, 16>(acc, out, ldk4);
  kb_store_done();
}

constexpr int kKbResStores = 16;   // store instructions per 32-row block: one dword per accumulator register

// kernel_block_res: loads of quads J0 .. J1 - 1 of the streamed operand (inline asm: see the kernel), the wait in front of their use,
// and the groups of a block in their order (ragged last group first).  NRB = 32-row blocks of Z1 the wave holds (2, or 1 for
// the wave of a short last row group: ONE chain of MFMAs in the same order as a full group's, because the rows it shares with
// the group before it -- it is moved back to end at n1 -- are written by both waves and must get the same bits from both;
// split over two accumulators, even / odd steps, the shared rows came out different by a rounding from run to run).
template <int J, int J0, int J1>
__device__ __forceinline__ void kb_res_loads(f32x4 (&sb)[J], int off, __amdgpu_buffer_rsrc_t rs) {
  if constexpr (J0 < J1) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(sb[J0]) : "v"(off), "s"(rs), "n"(32 * J0));
    kb_res_loads<J, J0 + 1, J1>(sb, off, rs);
  }
}
template <int J0, int J1, int N, int J>
__device__ __forceinline__ void kb_res_wait(f32x4 (&sb)[J]) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(sb[J0]) : "n"(N));
#pragma unroll
  for (int j = J0 + 1; j < J1; ++j) asm volatile("" : "+v"(sb[j]));
}
template <int J, bool ODD, int NRB, int GI>
__device__ __forceinline__ void kb_res_groups(f32x4 (&sb)[J], const f32x4 (&ra)[NRB][J], f32x16& acc0, f32x16& acc1, int offn,
                                              __amdgpu_buffer_rsrc_t rs, int h, f32x4 zero4) {
  constexpr int NG = (J + 3) / 4;
  if constexpr (GI < NG) {
    constexpr int grp = (GI + NG - 1) % NG, j0 = 4 * grp, j1 = j0 + 4 < J ? j0 + 4 : J;
    kb_res_wait<j0, j1, J - (j1 - j0) + kKbResStores * NRB>(sb);
    if constexpr (ODD && j1 == J) sb[J - 1] = h ? zero4 : sb[J - 1];
#pragma unroll
    for (int j = j0; j < j1; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[0][j][e], sb[j][e], acc0, 0, 0, 0);
        if constexpr (NRB == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[1][j][e], sb[j][e], acc1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    kb_res_loads<J, j0, j1>(sb, offn, rs);
    __builtin_amdgcn_sched_barrier(0);
    kb_res_groups<J, ODD, NRB, GI + 1>(sb, ra, acc0, acc1, offn, rs, h, zero4);
  }
}

// one wave's walk over the column blocks [b0, b1) for the NRB row blocks that start at row0
template <int Q, int NRB>
__device__ __forceinline__ void kb_res_walk(const float* __restrict__ Z1, int64_t n1, const float* __restrict__ Z2, int64_t n2,
                                            float scale, float* __restrict__ K, int64_t ldk, int64_t row0, int b0, int b1,
                                            int records) {
  constexpr int J = (Q + 1) / 2, m = 4 * Q;
  constexpr bool ODD = (Q & 1) != 0;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z1), (short)0, (int)(n1 * m * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Z2), (short)0, (int)(n2 * m * 4), 0x00020000);
#ifdef MGP_KB_STORE_WINDOW
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(records == 0 ? K : K + row0 * ldk, (short)0,
                                                                       records == 0 ? 2048 * 8192 : (int)(((32 * NRB - 1) * ldk + n2) * 4), 0x00020000);
#else
  const int kext = records == 0 ? 0 : (int)(((32 * NRB - 1) * ldk + n2) * 4);
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(K + row0 * ldk, (short)0, kext, 0x00020000);
#endif

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[NRB][J];
  f32x4 sb[J];
  const int last_col0 = (int)n2 - 32;
  auto col_of = [&](int b) { return b * 32 > last_col0 ? last_col0 : b * 32; };
  auto z2_off = [&](int col0) { return ((col0 + c) * m + 4 * h) * 4; };         // host side: n2 m < 2^29
  // prologue: the resident operand and the first block's streamed operand in ONE round trip (all inline asm: compiler-visible
  // loads of Z1 were sunk below the streamed operand's wait -- two round trips in a row)
#pragma unroll
  for (int i = 0; i < NRB; ++i) kb_res_loads<J, 0, J>(ra[i], (int)(((row0 + 32 * i + c) * m + 4 * h) * 4), rs1);
  kb_res_loads<J, 0, J>(sb, z2_off(col_of(b0)), rs2);
  kb_res_wait<0, J, 0>(sb);
#pragma unroll
  for (int i = 0; i < NRB; ++i) {
    kb_res_wait<0, J, 0>(ra[i]);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (ODD && j == J - 1) ra[i][j] = h ? zero4 : ra[i][j];
      ra[i][j] *= scale;
    }
  }
  // The streamed operand's loads are inline asm with the kernel's own vmcnt waits (as in kernel_block_pp): as compiler-visible
  // loads every block began with `s_waitcnt vmcnt(1)` / `vmcnt(0)` -- the wait-count pass merges the loop's back edge with the
  // prologue's pending loads and takes the stricter count -- i.e. with a wait for the previous block's last load AND its
  // stores.  Quads go in groups of four (the loads of a group share cache lines and are issued together), the ragged last
  // group first: a block's memory operations are then  L(g_0) ... L(g_last) S x 16 NRB  (kKbResStores = 16 dword stores per
  // row block),  each L(g) right behind the MFMAs that read g's registers for the last time, and in front of group g's MFMAs of
  // the next block exactly J - |g| loads and 16 NRB stores are younger than L(g): `vmcnt(J - |g| + 16 NRB)` waits for L(g) and
  // for nothing issued after it (the vector memory operations of one wave complete in order on gfx9, loads and stores alike:
  // the compiler's own wait counts rely on it).  The largest count, J = 16 - 4 + 32 = 44, stays under vmcnt's 6-bit limit of 63.
#ifdef MGP_KB_STORE_WINDOW
  const int row_k = records == 0 ? 128 : (int)ldk * 4;
#else
  const int row_k = (int)ldk * 4;                       // bytes per row of K
#endif
  const int lane_k = 4 * h * row_k + c * 4;
  for (int b = b0; b < b1; ++b) {
    const int col0 = col_of(b);
    const int offn = z2_off(col_of(b + 1 < b1 ? b + 1 : b));
    f32x16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc0[e] = 0.f, acc1[e] = 0.f;
    kb_res_groups<J, ODD, NRB, 0>(sb, ra, acc0, acc1, offn, rs2, h, zero4);
    // register r of acc_i (Z1 is the MFMA's first operand): row row0 + 32 i + 8 (r / 4) + 4 h + r % 4, column col0 + c -- a dword
    // store per register writes two full 128-byte lines (h = 0, 1), non-temporal like the general kernel's.  The row part
    // goes into the vector offset (one VALU add per store, in the other wave's shadow), not into the scalar offset, which
    // the descriptor's range check does not see.
#ifdef MGP_KB_STORE_WINDOW
    // lab build (tools/lab/build_variant.sh window features.hip -DMGP_KB_STORE_WINDOW): with knob 6 the stores are not dropped but land in a
    // private 8 KB window per wave at the start of K, which stays in L2 -- the store instructions without their HBM traffic
    const int okd = records == 0 ? (int)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8192 + (4 * h * 32 + c) * 4 : lane_k + col0 * 4;
#else
    const int okd = lane_k + col0 * 4;
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ro = (8 * (r / 4) + r % 4) * row_k;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc0[r]), rsk, okd + ro, 0, 2);
      if constexpr (NRB == 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc1[r]), rsk, okd + ro + 32 * row_k, 0, 2);
    }
    __builtin_amdgcn_sched_barrier(0);     // the waits above count these stores: none may move into the next block
  }
}

// The kernel block at m <= 128 modes (C3's posterior block: 600 x 60000, m = 100) without LDS and without barriers.  The modes of
// a 64-row group of Z1 are m registers per lane: a wave keeps them RESIDENT (scaled once), streams 32-row blocks of Z2 straight
// from global memory into the MFMA operand layout and owns a 64 x 32 piece of K per block -- 2 x 4J MFMAs = 6656 cycles of the
// pipe at m = 100, in which it issues 13 loads and 32 stores and waits for nothing: the streamed operand of block b + 1 is
// loaded into the registers block b's MFMAs have just read for the last time (four quads at a time, so that the four loads
// that share cache lines are issued together), one and a half thousand cycles and more before its first use.  Waves are
// independent; two per SIMD cover each other's store phases.  Measured at 600 x 60000 x 100 (tools/lab/kblock_shapes.py,
// tools/lab/pmc_kbres.sh): 79 us against the lean LDS kernel's 100 on the same box, 68 with every store dropped, matrix pipe
// busy 48 us (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs at 2.4 GHz).
//  * Z1 is the MFMA's first operand, so accumulator register r of lane (c, h) is row 8 (r / 4) + 4 h + r % 4, column c of the
//    32 x 32 block: a dword store per register writes two full 128-byte lines.  (With Z2 first a lane holds four consecutive
//    columns of one row and a block leaves in 8 stores of 16 bytes per lane = 32 pieces of 32 bytes each: 4 % slower here.)
//  * operand layout: lane (c = lane & 31, h = lane >> 5) holds quads 2 j + h of "its" row, so MFMA step 4 j + e multiplies modes
//    (8 j + e, 8 j + 4 + e): a 16-byte load per lane per quad, no transposition.  Not the (2 s, 2 s + 1) pairing of the LDS kernels:
//    the sum over modes is taken in another order and `scale` multiplies Z1 instead of the sum -- same value to fp32 rounding,
//    not the same bits (the parity test holds it to the fp64 product like the others and to the lean kernel at the same bound).
//  * an odd quad count (m = 100: 25) leaves quad 2 J - 1 of the h = 1 lanes past the row: both operands are zeroed there (the
//    address itself is inside the operand or dropped by the descriptor).
//  * work: G = ceil(n1 / 64) row groups; when the last group has 32 rows or fewer it is ONE row block (600 rows: 9 groups + 24
//    rows -> 19 row blocks of MFMA work instead of 20), and its wave walks twice as many column blocks.  The column blocks are
//    cut into S super-ranges; in each, every full group has two waves (a half each) and a short last group one wave: wave
//    w -> (s = w / WT, k = w % WT), WT = waves per super-range.  The waves of a super-range run side by side (same workgroups,
//    same XCD: logical workgroup ids are dealt to XCDs in runs) and read Z2 once from HBM between them.  Edge groups / blocks
//    are moved back to end at n1 / n2 like the lean kernels' tiles.
//  * every address goes through a descriptor of the operand's exact extent (Z1, Z2) or of the row group's rows of K.
template <int Q>
__global__ __launch_bounds__(kBlock, 2) void kernel_block_res(const float* __restrict__ Z1, int64_t n1,
                                                              const float* __restrict__ Z2, int64_t n2, float scale,
                                                              float* __restrict__ K, int64_t ldk, int G, int S, int nblk,
                                                              int short_last, int records) {
  const int nwg = (int)gridDim.x;                       // host side: a multiple of 8
  const int lb = ((int)blockIdx.x & 7) * (nwg >> 3) + ((int)blockIdx.x >> 3);
  const int w = __builtin_amdgcn_readfirstlane(lb * (kBlock / 64) + (int)(threadIdx.x >> 6));
  const int WT = 2 * G - short_last;
  if (w >= WT * S) return;
  const int s = w / WT, k = w % WT;
  const int s0 = (int)((unsigned)s * (unsigned)nblk / (unsigned)S);            // host side: S nblk < 2^31
  const int s1 = (int)((unsigned)(s + 1) * (unsigned)nblk / (unsigned)S);
  const int mid = (s0 + s1 + 1) >> 1;
  if (short_last && k == WT - 1) {
    kb_res_walk<Q, 1>(Z1, n1, Z2, n2, scale, K, ldk, n1 - 32, s0, s1, records);
  } else {
    int64_t row0 = (int64_t)(k >> 1) * 64;
    if (row0 + 64 > n1) row0 = n1 - 64;
    kb_res_walk<Q, 2>(Z1, n1, Z2, n2, scale, K, ldk, row0, (k & 1) ? mid : s0, (k & 1) ? s1 : mid, records);
  }
}

__global__ void kernel_diag_kernel(const float* __restrict__ Z1, const float* __restrict__ Z2, int64_t n, int m,
                                   float scale, float* __restrict__ out) {
  // one 16-lane group per row
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  for (int64_t r = g; r < n; r += ng) {
    float acc = 0.f;
    for (int j = lane; j < m; j += 16) acc = fmaf(Z1[r * m + j], Z2[r * m + j], acc);
    acc = mgp_group_sum<16>(acc);
    if (lane == 0) out[r] = scale * acc;
  }
}

// ---------------------------------------------------------------- low-rank apply
// t = Z^T X  ([m, C]) via per-workgroup partials; Y = alpha Z t + beta X
__global__ __launch_bounds__(kBlock) void zt_x_partial(const float* __restrict__ Z, int64_t n, int m,
                                                       const float* __restrict__ X, int C,
                                                       float* __restrict__ partial, int64_t rows_per_block) {
  __shared__ float sh[kBlock];
  int TC = 1;
  while (TC < m && TC < kBlock) TC <<= 1;
  const int TS = kBlock / TC;
  const int cc = threadIdx.x % TC, sl = threadIdx.x / TC;
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  for (int c = 0; c < C; ++c)
    for (int j0 = 0; j0 < m; j0 += TC) {
      const int j = j0 + cc;
      float acc = 0.f;
      if (j < m)
        for (int64_t r = r0 + sl; r < r1; r += TS) acc = fmaf(Z[r * m + j], X[r * C + c], acc);
      sh[threadIdx.x] = acc;
      __syncthreads();
      if (sl == 0 && j < m) {
        float t = 0.f;
        for (int s = 0; s < TS; ++s) t += sh[s * TC + cc];
        partial[((int64_t)blockIdx.x * C + c) * m + j] = t;
      }
      __syncthreads();
    }
}

__global__ void zt_x_finalize(const float* __restrict__ partial, int nblk, int mc, float* __restrict__ t) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < mc; i += gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * mc + i];
    t[i] = s;
  }
}

__global__ void z_t_axpby(const float* __restrict__ Z, int64_t n, int m, const float* __restrict__ t,
                          const float* __restrict__ X, int C, float alpha, float beta, float* __restrict__ Y) {
  // one 16-lane group per (row, column)
  const int lane = threadIdx.x & 15;
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) >> 4;
  const int64_t total = n * C;
  for (int64_t e = g; e < total; e += ng) {
    const int64_t r = e / C;
    const int c = (int)(e % C);
    float acc = 0.f;
    for (int j = lane; j < m; j += 16) acc = fmaf(Z[r * m + j], t[c * m + j], acc);
    acc = mgp_group_sum<16>(acc);
    if (lane == 0) Y[e] = alpha * acc + beta * X[e];
  }
}

int lowrank_blocks(int64_t n) {
  int64_t b = mgp_cdiv(n, 256);
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mgp_eigvec_postprocess(float* evecs, int64_t n, int m, const float* degree, float* colnorm_work,
                                      void* stream) {
  if (!evecs || !degree || !colnorm_work || n <= 0 || m <= 0) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  int64_t nblk = mgp_cdiv(n, 1024);
  if (nblk > kNormChunks) nblk = kNormChunks;
  const int64_t rpb = mgp_cdiv(n, nblk);
  nblk = mgp_cdiv(n, rpb);
  float* partial = colnorm_work + m;     // [nblk][m]; colnorm_work[0..m) = 1/norm
  hipLaunchKernelGGL(scale_rows_partial_norm, dim3((int)nblk), dim3(kBlock), 0, st, evecs, n, m, degree, partial, rpb);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(finalize_colnorm, dim3((int)mgp_cdiv(m, kBlock)), dim3(kBlock), 0, st, partial, (int)nblk, m,
                     colnorm_work);
  MGP_LAUNCH_CHECK();
  int64_t grid = mgp_cdiv(n * m, kBlock);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(scale_cols, dim3((int)grid), dim3(kBlock), 0, st, evecs, n, m, colnorm_work);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" size_t mgp_eigvec_postprocess_work_floats(int m) { return (size_t)(kNormChunks + 1) * (size_t)m; }

extern "C" int mgp_features_insample(const float* evals_dev, const float* evecs, int64_t n, int m, int nu,
                                     float kappa, float* Z, void* stream) {
  if (!evals_dev || !evecs || !Z || n <= 0 || m <= 0 || m > kMaxModes || nu < 1) return MGP_ERR_ARG;
  int64_t grid = mgp_cdiv(n * m, kBlock);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(features_insample_kernel, dim3((int)grid), dim3(kBlock), 0, mgp_stream(stream), evals_dev, evecs,
                     n, m, nu, kappa, Z);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" int mgp_features_oos(const float* evals_dev, const float* evecs, int64_t n, int m, int nu, float kappa,
                                float eps, int normalization, const float* degree_unnorm, const float* degree,
                                const float* knn_d2, const int32_t* knn_idx, int64_t T, int k, float bump_scale,
                                float bump_decay, float* Z, void* stream) {
  if (!evals_dev || !evecs || !degree_unnorm || !degree || !knn_d2 || !knn_idx || !Z) return MGP_ERR_ARG;
  if (n <= 0 || m <= 0 || m > kMaxModes || T <= 0 || k <= 0 || k > 1024 || nu < 1) return MGP_ERR_ARG;
  if (normalization < 0 || normalization > 1) return MGP_ERR_ARG;
  hipLaunchKernelGGL(features_oos_kernel, dim3((unsigned)T), dim3(kBlock), 0, mgp_stream(stream), evals_dev, evecs, n,
                     m, nu, kappa, eps, normalization, degree_unnorm, degree, knn_d2, knn_idx, k, bump_scale,
                     bump_decay, Z);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

constexpr int kPpBlocks = 256;     // one 512-thread workgroup per CU, two tile walks each
// Measured (profiles/r02_kernel_block.txt; us: general kernel | lean one tile per workgroup | two-half walk):
//   600 x 60000 x 100 (2345 tiles)   106 | 105 | 107      1000 x 50000 x 64 (3128)      83 |  79 |  90
//   4096 x 60000 x 100 (15008)       628 | 493 | 544      8192 x 8192 x 256 (4096)     300 | 280 | 283
//   600 x 60000 x 128                120 | 123 | 119      32768 x 60000 x 128 (120064) 4829 | 4761 | 4609
// (clock 2.31 GHz on random operands, 2.39 on zeros: the chip does not hold its clock down here; with the stores dropped the
// walk reaches 127-129 TFLOP/s.)  Among the LDS kernels the lean one-tile kernel is taken wherever its conditions hold; the walk is
// a knob: it wins a few per cent on the largest blocks only, its static split loses to the hardware's dispatch of 2000-4000 tiles.
// Round 4, the resident-operand kernel (tools/lab/kblock_shapes.py res, us: lean | resident, same box, back to back):
//   600 x 60000 x 100     99.8 |  79.4      600 x 60000 x 52     72.5 |  49.6      1000 x 50000 x 64    84.9 |  63.7
//   600 x 60000 x 124    103.4 |  81.3      600 x 60000 x 128   104.9 |  83.7      4096 x 60000 x 128  612.2 | 507.8 (124 TFLOP/s)
//   8192 x 60000 x 100  1012.6 | 881.0      20000 x 20000 x 88  798.4 | 637.6      64 x 60000 x 100     21.1 |  13.1
// It is faster at every shape it admits (16 <= m <= 128, multiples of 4), so the default takes it wherever it applies.
// 0 = by shape (default): the resident-operand kernel where the operands allow it, else the lean one-tile kernel, else the
// general one; 1 = lean one tile per workgroup where the operands allow it, 2 = the two-half walk where they allow it,
// 3 = as 2 with every store dropped (timing), 4 = the general kernel always (kernel_block_mfma: any shape, any alignment)
// 5 = as 0 (the resident-operand kernel, kernel_block_res, where the operands allow it), 6 = as 5 with every store dropped (timing)
std::atomic<int> g_kblock_pipe{0};
constexpr int kResMinModes = 16, kResMaxModes = 128;
constexpr int kResWaves = 2048;    // 256 CUs x 4 SIMDs x 2 waves: every wave of kernel_block_res is resident from the start

// K row stride ldk >= n2 (internal: the eigensolver rotates blocks in place of wider buffers)
int mgp_kernel_block_ld(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale, float* K,
                        int64_t ldk, void* stream) {
  if (!Z1 || !Z2 || !K || n1 <= 0 || n2 <= 0 || m <= 0 || ldk < n2) return MGP_ERR_ARG;
  const int kb_pipe = g_kblock_pipe;       // (lab knob: read once per call)
  dim3 grid((unsigned)mgp_cdiv(n2, kKB), (unsigned)mgp_cdiv(n1, kKB));
  const bool vec = (m % 4 == 0) && ((reinterpret_cast<uintptr_t>(Z1) | reinterpret_cast<uintptr_t>(Z2)) & 15) == 0;
  const bool x4 = vec && n2 % 4 == 0 && ldk % 4 == 0 && (reinterpret_cast<uintptr_t>(K) & 15) == 0;
  const int64_t ntiles = (int64_t)grid.x * grid.y;
  // the lean kernels: 16-byte stores, edge tiles moved back, 32-bit staging offsets, a 31-bit tile extent
  const bool lean = x4 && kb_pipe != 4 && n1 >= kKB && n2 >= kKB && ldk * 512 < (int64_t(1) << 31) &&
                    ntiles < (int64_t(1) << 30) && n1 * m < (int64_t(1) << 29) && n2 * m < (int64_t(1) << 29);
  const bool walk = kb_pipe == 2 || kb_pipe == 3;
  // the resident-operand kernel: 16 <= m <= 128 (4..32 quads: Z1's 64-row group in 4 ceil(Q / 2) x 2 registers, 245 VGPRs at m = 128),
  // n1 >= 64, n2 >= 32
  const int64_t G = mgp_cdiv(n1, 64), nblk = mgp_cdiv(n2, 32);
  const bool res_ok = x4 && m >= kResMinModes && m <= kResMaxModes && n1 >= 64 && n2 >= 32 && 2 * G <= kResWaves && ldk * 512 < (int64_t(1) << 31) &&
                      n1 * m < (int64_t(1) << 29) && n2 * m < (int64_t(1) << 29);
  const bool res = res_ok && (kb_pipe == 0 || kb_pipe == 5 || kb_pipe == 6);
  if (res) {
    // a last group of 32 rows or fewer is one row block (n1 >= 64: it is moved back to end at n1)
    const int short_last = (n1 - (G - 1) * 64 <= 32 && G > 1) ? 1 : 0;
    const int64_t WT = 2 * G - short_last;
    // two waves per SIMD whatever the register count (m <= 48 would admit four: measured, no gain -- those blocks are bound
    // by their stores)
    const int64_t S = std::max<int64_t>(1, std::min<int64_t>((nblk + 1) / 2, kResWaves / WT));
    const int64_t nwg = (mgp_cdiv(WT * S, kBlock / 64) + 7) / 8 * 8;
    const int records = kb_pipe == 6 ? 0 : 1;
#define MGP_KB_LAUNCH(Q)                                                                                                    \
  case Q:                                                                                                                   \
    hipLaunchKernelGGL(kernel_block_res<Q>, dim3((unsigned)nwg), dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, scale, K, ldk, \
                       (int)G, (int)S, (int)nblk, short_last, records);                                                     \
    break
    switch (m / 4) {
      MGP_KB_LAUNCH(4); MGP_KB_LAUNCH(5); MGP_KB_LAUNCH(6); MGP_KB_LAUNCH(7); MGP_KB_LAUNCH(8); MGP_KB_LAUNCH(9);
      MGP_KB_LAUNCH(10); MGP_KB_LAUNCH(11); MGP_KB_LAUNCH(12); MGP_KB_LAUNCH(13); MGP_KB_LAUNCH(14); MGP_KB_LAUNCH(15);
      MGP_KB_LAUNCH(16); MGP_KB_LAUNCH(17); MGP_KB_LAUNCH(18); MGP_KB_LAUNCH(19); MGP_KB_LAUNCH(20); MGP_KB_LAUNCH(21);
      MGP_KB_LAUNCH(22); MGP_KB_LAUNCH(23); MGP_KB_LAUNCH(24); MGP_KB_LAUNCH(25); MGP_KB_LAUNCH(26); MGP_KB_LAUNCH(27);
      MGP_KB_LAUNCH(28); MGP_KB_LAUNCH(29); MGP_KB_LAUNCH(30); MGP_KB_LAUNCH(31); MGP_KB_LAUNCH(32);
      default: return MGP_ERR_ARG;
    }
#undef MGP_KB_LAUNCH
    MGP_LAUNCH_CHECK();
    return MGP_OK;
  }
  if (lean && walk) {
    // tile id = column tile * row tiles + row tile, so the tiles in flight together share their Z2 rows
    const int64_t nb = (ntiles + 1) / 2 < kPpBlocks ? (ntiles + 1) / 2 : kPpBlocks;
    const int records = kb_pipe == 3 ? 0 : 0x7fffffff;   // 3: a descriptor of zero bytes drops every store (timing only)
    const int ts = (m - (m - 1) / kKC * kKC) / 2;      // 2-mode steps of the last 16-mode stage: 2, 4, 6 or 8
#define MGP_KB_LAUNCH(TS)                                                                                                   \
  hipLaunchKernelGGL(kernel_block_pp<TS>, dim3((unsigned)nb), dim3(kPpThreads), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, \
                     ldk, (int)grid.y, (int)ntiles, records)
    if (ts == 8) MGP_KB_LAUNCH(8);
    else if (ts == 6) MGP_KB_LAUNCH(6);
    else if (ts == 4) MGP_KB_LAUNCH(4);
    else MGP_KB_LAUNCH(2);
#undef MGP_KB_LAUNCH
    MGP_LAUNCH_CHECK();
    return MGP_OK;
  }
  if (lean) {
    const int records = 0x7fffffff;
    const int ts = (m - (m - 1) / kKC * kKC) / 2;
#define MGP_KB_LAUNCH(TS)                                                                                                   \
  hipLaunchKernelGGL(kernel_block_one<TS>, dim3((unsigned)ntiles), dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, \
                     ldk, (int)grid.y, records)
    if (ts == 8) MGP_KB_LAUNCH(8);
    else if (ts == 6) MGP_KB_LAUNCH(6);
    else if (ts == 4) MGP_KB_LAUNCH(4);
    else MGP_KB_LAUNCH(2);
#undef MGP_KB_LAUNCH
    MGP_LAUNCH_CHECK();
    return MGP_OK;
  }
  if (x4) hipLaunchKernelGGL((kernel_block_mfma<true, true>), grid, dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, ldk);
  else if (vec) hipLaunchKernelGGL((kernel_block_mfma<true, false>), grid, dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, ldk);
  else hipLaunchKernelGGL((kernel_block_mfma<false, false>), grid, dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, ldk);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" int mgp_kernel_block(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale,
                                float* K, void* stream) {
  return mgp_kernel_block_ld(Z1, n1, Z2, n2, m, scale, K, n2, stream);
}

extern "C" int mgp_kernel_block_set_pipe(int mode) {
  if (mode < 0 || mode > 6) return MGP_ERR_ARG;    // 0: by shape (default); 1: lean one-tile; 2, 3: walk; 4: general; 5, 6: resident
  g_kblock_pipe = mode;
  return MGP_OK;
}

extern "C" int mgp_kernel_diag(const float* Z1, const float* Z2, int64_t n, int m, float scale, float* out,
                               void* stream) {
  if (!Z1 || !Z2 || !out || n <= 0 || m <= 0) return MGP_ERR_ARG;
  int64_t grid = mgp_cdiv(n * 16, kBlock);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(kernel_diag_kernel, dim3((int)grid), dim3(kBlock), 0, mgp_stream(stream), Z1, Z2, n, m, scale, out);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// out = scale * (V - Z T): the last step of the Woodbury solve (K + noise I)^-1 v with K = s Z Z^T, T = the
// m x C solution of the m x m system in fp64.  The row sums run in fp64 (v - Z t cancels when the fit is good).
namespace {
__global__ __launch_bounds__(256) void lowrank_residual_kernel(const float* __restrict__ Z, int64_t n, int m,
                                                               const double* __restrict__ T, const float* __restrict__ V, int C,
                                                               double scale, float* __restrict__ out) {
  extern __shared__ double t_s[];                      // [m * C]
  for (int i = threadIdx.x; i < m * C; i += 256) t_s[i] = T[i];
  __syncthreads();
  const int64_t total = n * C;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / C;
    const int c = (int)(e % C);
    const float* zr = Z + r * m;
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc = fma((double)zr[j], t_s[j * C + c], acc);
    out[e] = (float)(scale * ((double)V[e] - acc));
  }
}
}  // namespace

extern "C" int mgp_lowrank_residual(const float* Z, int64_t n, int m, const double* T, const float* V, int C, double scale,
                                    float* out, void* stream) {
  if (!Z || !T || !V || !out || n <= 0 || m <= 0 || C <= 0) return MGP_ERR_ARG;
  const size_t lds = (size_t)m * C * sizeof(double);
  if (lds > 48 * 1024) return MGP_ERR_UNSUPPORTED;
  const int grid = (int)std::min<int64_t>(8192, mgp_cdiv(n * C, 256));
  hipLaunchKernelGGL(lowrank_residual_kernel, dim3(grid), dim3(256), lds, mgp_stream(stream), Z, n, m, T, V, C, scale, out);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" size_t mgp_lowrank_workspace_bytes(int m, int C) {
  if (m <= 0 || C <= 0) return 0;
  return mgp_align((size_t)513 * m * C * sizeof(float)) + 256;
}

extern "C" int mgp_lowrank_apply(const float* Z, int64_t n, int m, const float* X, int C, float alpha, float beta,
                                 float* Y, void* work, size_t work_bytes, void* stream) {
  if (!Z || !X || !Y || !work || n <= 0 || m <= 0 || C <= 0) return MGP_ERR_ARG;
  if (work_bytes < mgp_lowrank_workspace_bytes(m, C)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  const int nblk = lowrank_blocks(n);
  const int64_t rpb = mgp_cdiv(n, nblk);
  const int nb = (int)mgp_cdiv(n, rpb);
  float* t = static_cast<float*>(work);
  float* partial = t + (size_t)m * C;
  hipLaunchKernelGGL(zt_x_partial, dim3(nb), dim3(kBlock), 0, st, Z, n, m, X, C, partial, rpb);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(zt_x_finalize, dim3((int)mgp_cdiv((int64_t)m * C, kBlock)), dim3(kBlock), 0, st, partial, nb,
                     m * C, t);
  MGP_LAUNCH_CHECK();
  int64_t grid = mgp_cdiv(n * C * 16, kBlock);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(z_t_axpby, dim3((int)grid), dim3(kBlock), 0, st, Z, n, m, t, X, C, alpha, beta, Y);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}
