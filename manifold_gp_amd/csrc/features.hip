// Spectral side of the Riemann kernel: eigenvector post-processing, in-sample and out-of-sample
// features, the dense kernel block K = Z1 Z2^T on the fp32 MFMA, its diagonal, and the low-rank
// covariance apply  y = alpha Z (Z^T x) + beta x.
//
// Reference: manifold_gp/kernels/riemann_kernel.py:79-149, riemann_matern_kernel.py:21-22,
// manifold_gp/operators/graph_laplacian_operator.py:146-157, manifold_gp/utils/torch_utils.py:38-41.
#include <math.h>
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
constexpr int kKB = 128;   // workgroup tile (rows of Z1 x rows of Z2)
constexpr int kKC = 16;    // modes per LDS stage (32 is no faster and wastes the tail stage at m = 100)

// VEC (m % 4 == 0, 16-byte aligned rows): 16-byte global loads, and the next 16-mode stage is fetched
// into registers while the MFMAs of the current one run -- with scalar staging and no prefetch every
// stage exposed a full memory round trip (~1 us against ~0.85 us of MFMA work per stage).
template <bool VEC>
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
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
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

// K row stride ldk >= n2 (internal: the eigensolver rotates blocks in place of wider buffers)
int mgp_kernel_block_ld(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale, float* K,
                        int64_t ldk, void* stream) {
  if (!Z1 || !Z2 || !K || n1 <= 0 || n2 <= 0 || m <= 0 || ldk < n2) return MGP_ERR_ARG;
  dim3 grid((unsigned)mgp_cdiv(n2, kKB), (unsigned)mgp_cdiv(n1, kKB));
  const bool vec = (m % 4 == 0) && ((reinterpret_cast<uintptr_t>(Z1) | reinterpret_cast<uintptr_t>(Z2)) & 15) == 0;
  if (vec) hipLaunchKernelGGL(kernel_block_mfma<true>, grid, dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, ldk);
  else hipLaunchKernelGGL(kernel_block_mfma<false>, grid, dim3(kBlock), 0, mgp_stream(stream), Z1, n1, Z2, n2, m, scale, K, ldk);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" int mgp_kernel_block(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale,
                                float* K, void* stream) {
  return mgp_kernel_block_ld(Z1, n1, Z2, n2, m, scale, K, n2, stream);
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
