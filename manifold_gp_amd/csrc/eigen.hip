// Eigensolve for the m smallest eigenpairs of the symmetric graph Laplacian, and the k-step
// Lanczos tridiagonalisation of a precision-family operator.
//
// Replaces torch.linalg.eigh on the densified N x N matrix (manifold_gp/kernels/riemann_kernel.py:
// 121-125, O(N^3) and 14 GB at N = 60k) and GraphLaplacianOperator.diagonalization
// (manifold_gp/operators/graph_laplacian_operator.py:132-144; its Lanczos branch is
// linear_operator's lanczos_tridiag with full re-orthogonalisation).
//
// mgp_lanczos_smallest: Chebyshev-filtered block Krylov iteration with Rayleigh-Ritz
//   The low end of a graph-Laplacian spectrum is clustered and, for a k-NN graph with several
//   connected components, degenerate; single-vector Lanczos needs thousands of steps there and
//   misses multiplicities (the reference itself abandoned its Lanczos call for dense eigh,
//   riemann_kernel.py:120).  So: (1) Gershgorin gives a safe upper bound ub of the spectrum (the norm the
//   residual test is relative to) and a 32-dimensional Krylov space a tight one, ubf ~ 1.03 lambda_max -- about half
//   of Gershgorin on k-NN graph Laplacians, checked against the Ritz values of every round, Gershgorin as fallback,
//   (2) a block of b = m + pad vectors is filtered by a scaled Chebyshev polynomial of L that
//   damps [a, ubf] (a = largest Ritz value of the previous round) -- d fused SpMM launches, the
//   matrix is streamed once per launch for all b columns, (3) Rayleigh-Ritz in the filtered
//   block: Gram matrices V^T V, V^T L V accumulated in fp64 on device, b x b generalized
//   eigenproblem in fp64 on the host (Householder + QL on a worker pool), rotation V <- V W on the fp32
//   MFMA kernel, (4) residuals ||L v - theta v|| <= tol ub decide convergence.
//
// mgp_lanczos_tridiag: q_{j+1} beta_j = A q_j - alpha_j q_j - beta_{j-1} q_{j-1} with classical
//   Gram-Schmidt against ALL previous vectors, twice; alpha/beta stay on device until the end.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <chrono>
#include <cstdio>
#include <atomic>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------- small kernels
__global__ void gershgorin_kernel(int64_t n, const int32_t* __restrict__ rowptr, const float* __restrict__ vals,
                                  const float* __restrict__ diag, float* __restrict__ block_max) {
  __shared__ float sh[kBlock];
  float mx = 0.f;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
    float s = fabsf(diag[r]);
    for (int i = rowptr[r]; i < rowptr[r + 1]; ++i) s += fabsf(vals[i]);
    mx = fmaxf(mx, s);
  }
  sh[threadIdx.x] = mx;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) block_max[blockIdx.x] = sh[0];
}

__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}

// V[r, c0:c1) = uniform(-1, 1), counter-based (reproducible for a seed)
__global__ void random_cols_kernel(float* __restrict__ V, int64_t n, int ld, int c0, int c1, uint64_t seed) {
  const int w = c1 - c0;
  const int64_t total = n * w;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / w;
    const int c = c0 + (int)(i % w);
    const uint32_t h = hash32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)r * 1315423911ULL + (uint64_t)c * 2654435761ULL + 12345);
    V[r * ld + c] = (float)h * (2.0f / 4294967296.0f) - 1.0f;
  }
}

// partial[chunk][i][j] = sum_{r in chunk} A[r,i] * B[r,j], fp64 accumulation, 64 x 64 tile per block
__global__ __launch_bounds__(kBlock) void gram_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      int64_t n, int b, int64_t rows_per_chunk,
                                                      double* __restrict__ partial) {
  __shared__ float As[16][64 + 1];
  __shared__ float Bs[16][64 + 1];
  const int ti = blockIdx.x, tj = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.z * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // outputs i = ti*64 + ty*4 + a, j = tj*64 + tx*4 + c
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = 0.0;
  for (int64_t rr = r0; rr < r1; rr += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += kBlock) {
      const int lr = e >> 6, lc = e & 63;
      const int64_t r = rr + lr;
      const int ci = ti * 64 + lc, cj = tj * 64 + lc;
      As[lr][lc] = (r < r1 && ci < b) ? A[r * b + ci] : 0.f;
      Bs[lr][lc] = (r < r1 && cj < b) ? B[r * b + cj] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int lr = 0; lr < 16; ++lr) {
      double av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) av[a] = (double)As[lr][ty * 4 + a];
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = (double)Bs[lr][tx * 4 + c];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = fma(av[a], bv[c], acc[a][c]);
    }
    __syncthreads();
  }
  double* out = partial + (int64_t)blockIdx.z * b * b;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int i = ti * 64 + ty * 4 + a, j = tj * 64 + tx * 4 + c;
      if (i < b && j < b) out[(int64_t)i * b + j] = acc[a][c];
    }
}

// The same partial Gram blocks on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: exact products of the converted fp32 entries,
// fp64 accumulation): same 64 x 64 tile per block and 16-row LDS stages as gram_kernel; a wave owns 32 x 32 of the tile as
// 2 x 2 MFMA tiles and per four rows reads two A^T and two B operands from LDS (lane (i, k) = As[row k][col i]: conflict
// free), converts them and issues four MFMAs -- 16 MFMAs per stage where the vector form issues 256 DFMAs and 128 converts.
using f64x4 = __attribute__((ext_vector_type(4))) double;
__global__ __launch_bounds__(kBlock) void gram_mfma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           int64_t n, int b, int64_t rows_per_chunk,
                                                           double* __restrict__ partial) {
  // Staging (round 5, second half): 32 rows x 64 columns of A and of B per stage, one 16-byte global load per thread, item and
  // matrix (8 scalar loads per stage of 16 rows before), requested a stage ahead into registers and written to LDS behind the
  // barrier; rows padded to 80 floats: the four k-groups of a ds_read_b32 (lanes 16 apart read rows one apart) then fall 16 banks
  // apart (65 floats: one bank apart, SQ_LDS_BANK_CONFLICT 40 % of the LDS cycles).  Half the barriers per row of the chunk.
  constexpr int kGR = 32, kGP = 64 + 16;
  __shared__ __attribute__((aligned(16))) float As[kGR][kGP];
  __shared__ __attribute__((aligned(16))) float Bs[kGR][kGP];
  const int ti = blockIdx.x, tj = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.z * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  f64x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[a][c] = f64x4{0.0, 0.0, 0.0, 0.0};
  // thread -> (row lr, 4 columns from lc) of the stage, two such items per thread and matrix (32 x 64 floats = 512 float4)
  const bool vec = (b & 3) == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  float4 pa[2], pb[2];
  auto fetch = [&](int64_t rr) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = threadIdx.x + u * kBlock;
      const int lr = e >> 4, lc = (e & 15) * 4;
      const int64_t r = rr + lr;
      const int ci = ti * 64 + lc, cj = tj * 64 + lc;
      pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      pb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < r1) {
        if (vec && ci + 3 < b) pa[u] = *reinterpret_cast<const float4*>(A + r * b + ci);
        else {
          if (ci < b) pa[u].x = A[r * b + ci];
          if (ci + 1 < b) pa[u].y = A[r * b + ci + 1];
          if (ci + 2 < b) pa[u].z = A[r * b + ci + 2];
          if (ci + 3 < b) pa[u].w = A[r * b + ci + 3];
        }
        if (vec && cj + 3 < b) pb[u] = *reinterpret_cast<const float4*>(B + r * b + cj);
        else {
          if (cj < b) pb[u].x = B[r * b + cj];
          if (cj + 1 < b) pb[u].y = B[r * b + cj + 1];
          if (cj + 2 < b) pb[u].z = B[r * b + cj + 2];
          if (cj + 3 < b) pb[u].w = B[r * b + cj + 3];
        }
      }
    }
  };
  if (r0 < r1) fetch(r0);
  for (int64_t rr = r0; rr < r1; rr += kGR) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = threadIdx.x + u * kBlock;
      const int lr = e >> 4, lc = (e & 15) * 4;
      *reinterpret_cast<float4*>(&As[lr][lc]) = pa[u];
      *reinterpret_cast<float4*>(&Bs[lr][lc]) = pb[u];
    }
    __syncthreads();
    if (rr + kGR < r1) fetch(rr + kGR);            // the next stage's rows are in flight during this stage's MFMAs
#pragma unroll
    for (int k0 = 0; k0 < kGR; k0 += 4) {
      double av[2], bv[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) av[a] = (double)As[k0 + kq][wi + 16 * a + l16];
#pragma unroll
      for (int c = 0; c < 2; ++c) bv[c] = (double)Bs[k0 + kq][wj + 16 * c + l16];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[c], acc[a][c], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[a][c][r]: row i = wi + 16 a + 4 r + kq, column j = wj + 16 c + l16 of the tile (the f64 16x16x4 result interleaves the
  // four k-groups of lanes over the rows: register r of lane group kq is row 4 r + kq -- not the f32 layout's 4 kq + r;
  // found with tools/lab/dbg_gram.py)
  double* out = partial + (int64_t)blockIdx.z * b * b;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti * 64 + wi + 16 * a + 4 * r + kq, j = tj * 64 + wj + 16 * c + l16;
        if (i < b && j < b) out[(int64_t)i * b + j] = acc[a][c][r];
      }
}

__global__ void gram_reduce_kernel(const double* __restrict__ partial, int chunks, int bb, double* __restrict__ G) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < bb; i += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * bb + i];
    G[i] = s;
  }
}

// column partial sums of (LV[r,c] - theta[c] * V[r,c])^2 in fp64
__global__ __launch_bounds__(kBlock) void residual_kernel(const float* __restrict__ LV, const float* __restrict__ V,
                                                          const float* __restrict__ theta, int64_t n, int b,
                                                          int64_t rows_per_chunk, double* __restrict__ partial) {
  __shared__ double sh[kBlock];
  int TC = 1;
  while (TC < b && TC < kBlock) TC <<= 1;
  const int TS = kBlock / TC;
  const int cc = threadIdx.x % TC, sl = threadIdx.x / TC;
  const int64_t r0 = blockIdx.x * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  double acc = 0.0;
  if (cc < b) {
    const float th = theta[cc];
    for (int64_t r = r0 + sl; r < r1; r += TS) {
      const float d = LV[r * b + cc] - th * V[r * b + cc];
      acc += (double)d * (double)d;
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && cc < b) {
    double t = 0.0;
    for (int s = 0; s < TS; ++s) t += sh[s * TC + cc];
    partial[(int64_t)blockIdx.x * b + cc] = t;
  }
}

// dst[r, dc0 + j] = src[r, sc0 + j], j < m  (column sub-blocks between row-major blocks of different widths)
__global__ void move_cols_kernel(const float* __restrict__ src, int64_t n, int sld, int sc0, int m, float* __restrict__ dst,
                                 int dld, int dc0) {
  const int64_t total = n * m;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    dst[(i / m) * dld + dc0 + (i % m)] = src[(i / m) * sld + sc0 + (i % m)];
}

__global__ void copy_cols_kernel(const float* __restrict__ V, int64_t n, int ld, int m, float* __restrict__ out) {
  const int64_t total = n * m;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = V[(i / m) * ld + (i % m)];
}

// K[r, j] = KT[j, r] for kk <= 64 vectors of length n kept one after the other (the Krylov vectors of the lambda_max estimate: each is
// written in place by its SpMV launch; one transpose instead of a strided column copy behind every launch)
__global__ __launch_bounds__(256) void vecs_to_cols_kernel(const float* __restrict__ KT, int64_t n, int kk, float* __restrict__ K) {
  __shared__ float tile[64][64 + 1];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * kk; e += 256) {
    const int j = e >> 6, lr = e & 63;
    tile[j][lr] = r0 + lr < n ? KT[(int64_t)j * n + r0 + lr] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * kk; e += 256) {
    const int lr = e / kk, j = e - lr * kk;
    if (r0 + lr < n) K[(r0 + lr) * kk + j] = tile[j][lr];
  }
}

// ---------------------------------------------------------------- host: symmetric eigensolver (fp64)
// Householder tridiagonalisation + implicit-shift QL, restated so that every O(n^3) loop walks a ROW of a row-major
// array and the eigenvector update runs on host threads:
//   1. T = Q^T A Q on the lower triangle (symmetric rank-2 updates, 4/3 n^3 flops);
//   2. Z^T = Q^T accumulated by right multiplications (4/3 n^3);
//   3. QL on (d, e) alone -- its plane rotations are RECORDED (they do not depend on the vectors);
//   4. the ~n^2 recorded rotations are applied to Z^T, rows i / i+1, over column slices of 32: a slice is an
//      L1-resident private copy owned by one host thread (3 n^3 flops, the largest part, now parallel).
// Round 3: the b = 128 Rayleigh-Ritz problem took 2.0 ms per round in the column-walking EISPACK form this replaces
// (4 rounds = 8 of the 48 ms of the 60k eigensolve).  Dot products use four interleaved partial sums in a fixed
// order and -ffp-contract=off holds for the host pass too, so the result does not depend on the thread count or on
// whether the AVX2 clones run.  A [n x n] row-major symmetric; evals ascending; eigenvectors = columns of V.
#define MGP_HOST_INLINE static inline __attribute__((always_inline))

MGP_HOST_INLINE double dot4(const double* __restrict__ a, const double* __restrict__ b, int n) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    s0 += a[i] * b[i];
    s1 += a[i + 1] * b[i + 1];
    s2 += a[i + 2] * b[i + 2];
    s3 += a[i + 3] * b[i + 3];
  }
  double s = (s0 + s2) + (s1 + s3);
  for (; i < n; ++i) s += a[i] * b[i];
  return s;
}

struct PlaneRot { int i; double c, s; };

// T = Q^T A Q, Q = H_0 H_1 ... H_{n-3}, H_k = I - tau_k v_k v_k^T with v_k in indices k+1 .. n-1 (v_k[k+1] = 1; row k
// of hv).  Only the LOWER triangle of W is read and written.  d = diagonal of T, e[i] = T[i+1, i].
MGP_HOST_INLINE void householder_tridiag_impl(int n, int ld, double* W, double* d, double* e, double* hv, double* tau,
                                              double* p, double* w) {
  for (int k = 0; k + 2 < n; ++k) {
    const int s = n - k - 1;
    double* v = hv + (size_t)k * ld + (k + 1);
    double scale = 0.0, tail = 0.0;
    for (int j = 0; j < s; ++j) {
      v[j] = W[(size_t)(k + 1 + j) * ld + k];
      scale = std::max(scale, fabs(v[j]));
      if (j) tail = std::max(tail, fabs(v[j]));
    }
    d[k] = W[(size_t)k * ld + k];
    const double alpha = v[0];
    if (tail == 0.0) { tau[k] = 0.0; e[k] = alpha; continue; }   // the column is tridiagonal already: H_k = I
    double ss = 0.0;
    for (int j = 0; j < s; ++j) { const double t = v[j] / scale; ss += t * t; }
    const double nrm = scale * sqrt(ss);
    const double beta = alpha > 0.0 ? -nrm : nrm;
    const double tk = (beta - alpha) / beta;
    const double inv = 1.0 / (alpha - beta);
    v[0] = 1.0;
    for (int j = 1; j < s; ++j) v[j] *= inv;
    tau[k] = tk;
    e[k] = beta;
    double* B = W + (size_t)(k + 1) * ld + (k + 1);
    // p = tau B v: row j of the lower triangle gives its own dot product and its share of the entries above it
    for (int j = 0; j < s; ++j) {
      const double* __restrict__ row = B + (size_t)j * ld;
      const double vj = v[j];
      const double t = dot4(row, v, j);
      for (int i = 0; i < j; ++i) p[i] += row[i] * vj;
      p[j] = t + row[j] * vj;
    }
    for (int j = 0; j < s; ++j) p[j] *= tk;
    const double hh = 0.5 * tk * dot4(p, v, s);
    for (int j = 0; j < s; ++j) w[j] = p[j] - hh * v[j];
    for (int j = 0; j < s; ++j) {          // B -= v w^T + w v^T
      double* __restrict__ row = B + (size_t)j * ld;
      const double vj = v[j], wj = w[j];
      for (int i = 0; i <= j; ++i) row[i] -= vj * w[i] + wj * v[i];
    }
  }
  if (n >= 2) { d[n - 2] = W[(size_t)(n - 2) * ld + (n - 2)]; e[n - 2] = W[(size_t)(n - 1) * ld + (n - 2)]; }
  d[n - 1] = W[(size_t)(n - 1) * ld + (n - 1)];
  e[n - 1] = 0.0;
}

// Zt = Q^T = H_{n-3} ... H_0 as ((I H_{n-3}) H_{n-4}) ... H_0: M <- M - tau (M v) v^T touches rows and columns
// k+1 .. n-1 only (the rows above are still rows of the identity).
MGP_HOST_INLINE void householder_accumulate_impl(int n, int ld, const double* hv, const double* tau, double* Zt, int r0, int r1) {
  // rows [r0, r1) only: a row of M runs through all the H_k on its own, so row ranges are independent jobs
  for (int r = r0; r < r1; ++r) {
    for (int c = 0; c < n; ++c) Zt[(size_t)r * ld + c] = 0.0;
    Zt[(size_t)r * ld + r] = 1.0;
  }
  for (int k = std::min(n - 3, r1 - 2); k >= 0; --k) {
    if (tau[k] == 0.0) continue;
    const int s = n - k - 1;
    const double* __restrict__ v = hv + (size_t)k * ld + (k + 1);
    for (int r = std::max(r0, k + 1); r < r1; ++r) {
      double* __restrict__ row = Zt + (size_t)r * ld + (k + 1);
      const double g = tau[k] * dot4(row, v, s);
      for (int i = 0; i < s; ++i) row[i] -= g * v[i];
    }
  }
}

// rows i, i+1 of Zt <- the recorded rotations, columns [k0, k1): worked on in a compact private copy (n x len: 32 KB
// at n = 128 and 32 columns), so no cache line is shared with the neighbouring slices' threads
MGP_HOST_INLINE void apply_rots_impl(int n, int ld, double* Zt, const std::vector<PlaneRot>& rots, int k0, int k1) {
  const int len = k1 - k0;
  if (len <= 0) return;
  std::vector<double> loc((size_t)n * len);
  for (int r = 0; r < n; ++r) memcpy(&loc[(size_t)r * len], Zt + (size_t)r * ld + k0, len * sizeof(double));
  for (const PlaneRot& q : rots) {
    double* __restrict__ r0 = &loc[(size_t)q.i * len];
    double* __restrict__ r1 = r0 + len;
    const double c = q.c, s = q.s;
    for (int k = 0; k < len; ++k) {
      const double h = r1[k], g = r0[k];
      r1[k] = s * g + c * h;
      r0[k] = c * g - s * h;
    }
  }
  for (int r = 0; r < n; ++r) memcpy(Zt + (size_t)r * ld + k0, &loc[(size_t)r * len], len * sizeof(double));
}

// the same three loops compiled twice: baseline x86-64 and AVX2 (picked at run time; identical arithmetic)
void householder_tridiag_base(int n, int ld, double* W, double* d, double* e, double* hv, double* tau, double* p, double* w) {
  householder_tridiag_impl(n, ld, W, d, e, hv, tau, p, w);
}
__attribute__((target("avx2"))) void householder_tridiag_avx2(int n, int ld, double* W, double* d, double* e, double* hv,
                                                              double* tau, double* p, double* w) {
  householder_tridiag_impl(n, ld, W, d, e, hv, tau, p, w);
}
void householder_accumulate_base(int n, int ld, const double* hv, const double* tau, double* Zt, int r0, int r1) {
  householder_accumulate_impl(n, ld, hv, tau, Zt, r0, r1);
}
__attribute__((target("avx2"))) void householder_accumulate_avx2(int n, int ld, const double* hv, const double* tau, double* Zt, int r0,
                                                                 int r1) {
  householder_accumulate_impl(n, ld, hv, tau, Zt, r0, r1);
}
void apply_rots_base(int n, int ld, double* Zt, const std::vector<PlaneRot>& rots, int k0, int k1) {
  apply_rots_impl(n, ld, Zt, rots, k0, k1);
}
__attribute__((target("avx2"))) void apply_rots_avx2(int n, int ld, double* Zt, const std::vector<PlaneRot>& rots, int k0, int k1) {
  apply_rots_impl(n, ld, Zt, rots, k0, k1);
}

inline double pythag(double a, double b) {
  const double r2 = a * a + b * b;
  if (r2 > 1e-280 && r2 < 1e280) return sqrt(r2);
  return hypot(a, b);
}

// implicit-shift QL on the symmetric tridiagonal (d, e[i] = T[i+1, i]): eigenvalues into d (unsorted), the plane
// rotations (acting on vector indices i, i+1) appended to `rots` in the order they have to be applied
void tridiag_ql(int n, double* d, double* e, std::vector<PlaneRot>& rots) {
  double f = 0.0, tst1 = 0.0;
  const double eps = 2.220446049250313e-16;
  e[n - 1] = 0.0;
  for (int l = 0; l < n; ++l) {
    tst1 = std::max(tst1, fabs(d[l]) + fabs(e[l]));
    int m = l;
    while (m < n) {
      if (fabs(e[m]) <= eps * tst1) break;
      ++m;
    }
    if (m > l) {
      int iter = 0;
      do {
        ++iter;
        double g = d[l];
        double p = (d[l + 1] - g) / (2.0 * e[l]);
        double r = pythag(p, 1.0);
        if (p < 0) r = -r;
        d[l] = e[l] / (p + r);
        d[l + 1] = e[l] * (p + r);
        const double dl1 = d[l + 1];
        double h = g - d[l];
        for (int i = l + 2; i < n; ++i) d[i] -= h;
        f += h;
        p = d[m];
        double c = 1.0, c2 = c, c3 = c;
        const double el1 = e[l + 1];
        double s = 0.0, s2 = 0.0;
        for (int i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * e[i];
          h = c * p;
          r = pythag(p, e[i]);
          e[i + 1] = s * r;
          s = e[i] / r;
          c = p / r;
          p = c * d[i] - s * g;
          d[i + 1] = h + s * (c * g + s * d[i]);
          rots.push_back(PlaneRot{i, c, s});
        }
        p = -s * s2 * c3 * el1 * e[l] / dl1;
        e[l] = s * p;
        d[l] = c * p;
      } while (fabs(e[l]) > eps * tst1 && iter < 200);
    }
    d[l] = d[l] + f;
    e[l] = 0.0;
  }
}

// Host worker pool of one eigensolve: the Rayleigh-Ritz step has four short parallel sections per round (two b^3
// products, the eigenvector rotations, W = T S); starting fresh threads for each cost as much as their arithmetic.
// run(njobs, fn) calls fn(job) once per job on the workers and the calling thread and returns when all are done.  Which
// thread takes which job varies, the arithmetic of a job does not: every output element belongs to exactly one job.
class HostPool {
 public:
  explicit HostPool(int workers) {
    for (int t = 0; t < workers; ++t) th_.emplace_back([this]() { work(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& x : th_) x.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;
  int threads() const { return (int)th_.size() + 1; }
  void run(int njobs, const std::function<void(int)>& fn) {
    if (njobs <= 0) return;
    std::unique_lock<std::mutex> lk(mu_);
    job_ = &fn; njobs_ = njobs; next_ = 0; pending_ = njobs; ++gen_;
    lk.unlock();
    cv_.notify_all();
    lk.lock();
    take(lk);
    done_.wait(lk, [&]() { return pending_ == 0; });
    job_ = nullptr;
  }
  // fn(i) for i in [0, n): contiguous chunks of `chunk` rows as jobs
  void rows(int n, int chunk, const std::function<void(int)>& fn) {
    const int nj = (n + chunk - 1) / chunk;
    const std::function<void(int)> job = [&](int j) {
      const int i1 = std::min(n, (j + 1) * chunk);
      for (int i = j * chunk; i < i1; ++i) fn(i);
    };
    run(nj, job);
  }

 private:
  void take(std::unique_lock<std::mutex>& lk) {      // called with the lock held
    while (next_ < njobs_) {
      const int j = next_++;
      const std::function<void(int)>* f = job_;
      lk.unlock();
      (*f)(j);
      lk.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  void work() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [&]() { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      take(lk);
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  int njobs_ = 0, next_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

int host_pool_workers() {
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(hw ? hw : 1u, 8u) - 1;
}

// Gn (b x b, unit diagonal, symmetric positive definite) = C C^T; T = D C^-T (b x b) so that
// T^T (D^-1 Gn D^-1) T = I.  False when a pivot falls under 1e-10 (relative to the unit diagonal): the caller
// then needs the rank-revealing path.
bool cholesky_whiten(int b, const std::vector<double>& Gn, const std::vector<double>& dg, std::vector<double>& T) {
  // every inner loop walks rows: the dot products with four interleaved partial sums (a single running sum is a chain of
  // dependent adds), the inverse row by row as axpys
  std::vector<double> C((size_t)b * b, 0.0);
  for (int i = 0; i < b; ++i) {
    double* ci = &C[(size_t)i * b];
    for (int j = 0; j < i; ++j) {
      const double* cj = &C[(size_t)j * b];
      ci[j] = (Gn[(size_t)i * b + j] - dot4(ci, cj, j)) / cj[j];
    }
    const double s = Gn[(size_t)i * b + i] - dot4(ci, ci, i);
    if (!(s > 1e-10)) return false;
    ci[i] = sqrt(s);
  }
  // Ci = C^-1 (lower): row i = (e_i - sum_{k < i} C[i][k] Ci[k][:]) / C[i][i]
  std::vector<double> Ci((size_t)b * b, 0.0);
  for (int i = 0; i < b; ++i) {
    double* __restrict__ ri = &Ci[(size_t)i * b];
    const double* ci = &C[(size_t)i * b];
    for (int k = 0; k < i; ++k) {
      const double c = ci[k];
      const double* __restrict__ rk = &Ci[(size_t)k * b];
      for (int j = 0; j <= k; ++j) ri[j] -= c * rk[j];
    }
    const double inv = 1.0 / ci[i];
    for (int j = 0; j < i; ++j) ri[j] *= inv;
    ri[i] = inv;
  }
  // T = D C^-T (upper triangular): T[i][j] = dg[i] * Ci[j][i]
  T.assign((size_t)b * b, 0.0);
  for (int i = 0; i < b; ++i)
    for (int j = i; j < b; ++j) T[(size_t)i * b + j] = dg[i] * Ci[(size_t)j * b + i];
  return true;
}

void jacobi_eigh(int n, std::vector<double>& A, std::vector<double>& evals, std::vector<double>& V, HostPool* pool = nullptr) {
  // (name kept from the Jacobi days: every caller wants "eigh of a small symmetric matrix")
  // Padded leading dimension: with ld = n a power-of-two n (block sizes 128, 256) maps the rows of a column slice onto
  // a handful of L1 sets of the host CPU.
  evals.resize(n);
  V.assign((size_t)n * n, 0.0);
  if (n == 1) { evals[0] = A[0]; V[0] = 1.0; return; }
  const int ld = (n + 7) / 8 * 8 + 8;
  std::vector<double> W((size_t)n * ld), hv((size_t)n * ld, 0.0), Zt((size_t)n * ld), d(n), e(n), tau(n, 0.0), p(n, 0.0), w(n);
  for (int i = 0; i < n; ++i)      // use the symmetric part
    for (int j = 0; j <= i; ++j) W[(size_t)i * ld + j] = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]);
  const bool avx2 = __builtin_cpu_supports("avx2");
  (avx2 ? householder_tridiag_avx2 : householder_tridiag_base)(n, ld, W.data(), d.data(), e.data(), hv.data(), tau.data(),
                                                               p.data(), w.data());
  auto accumulate = avx2 ? householder_accumulate_avx2 : householder_accumulate_base;
  if (pool && n >= 64) {
    // later rows pass through more reflectors: jobs of 8 rows, taken in turn by whoever is free
    const int nj = (n + 7) / 8;
    pool->run(nj, [&](int j) { accumulate(n, ld, hv.data(), tau.data(), Zt.data(), j * 8, std::min(n, j * 8 + 8)); });
  } else {
    accumulate(n, ld, hv.data(), tau.data(), Zt.data(), 0, n);
  }
  std::vector<PlaneRot> rots;
  rots.reserve((size_t)n * n);
  tridiag_ql(n, d.data(), e.data(), rots);
  auto apply = avx2 ? apply_rots_avx2 : apply_rots_base;
  // column slices of 32 (the last one takes the remainder); on the caller's pool when there is one
  const int nsl = std::max(1, n / 32);
  auto slice = [&](int t) {
    const int k0 = t * 32, k1 = t + 1 == nsl ? n : (t + 1) * 32;
    apply(n, ld, Zt.data(), rots, k0, k1);
  };
  if (nsl == 1) {
    slice(0);
  } else if (pool) {
    pool->run(nsl, slice);
  } else {
    for (int q = 0; q < nsl; ++q) slice(q);
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
  for (int j = 0; j < n; ++j) {
    evals[j] = d[order[j]];
    const double* z = &Zt[(size_t)order[j] * ld];
    for (int k = 0; k < n; ++k) V[(size_t)k * n + j] = z[k];
  }
}

// symmetric tridiagonal (alpha[k], beta[k-1]) eigenvalues + first components of eigenvectors
void tridiag_eigh(int k, const std::vector<double>& alpha, const std::vector<double>& beta, std::vector<double>& evals,
                  std::vector<double>& evecs) {
  std::vector<double> T((size_t)k * k, 0.0);
  for (int i = 0; i < k; ++i) {
    T[(size_t)i * k + i] = alpha[i];
    if (i + 1 < k) { T[(size_t)i * k + i + 1] = beta[i]; T[(size_t)(i + 1) * k + i] = beta[i]; }
  }
  jacobi_eigh(k, T, evals, evecs);
}

// largest Ritz value of (H, G) on the leading k of kfull basis vectors (G = K^T K, H = K^T L K, fp64): rank-revealing
// whitening (directions under 1e-6 of the largest Gram eigenvalue are rounding of the fp32 vectors), then one small
// symmetric eigensolve.  NaN when the Gram block is unusable.
double top_ritz(int kfull, int k, const std::vector<double>& G, const std::vector<double>& H) {
  std::vector<double> dg(k), Gn((size_t)k * k), lam, U;
  for (int i = 0; i < k; ++i) {
    const double g = G[(size_t)i * kfull + i];
    if (!(g > 0.0) || !std::isfinite(g)) return NAN;
    dg[i] = 1.0 / sqrt(g);
  }
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < k; ++j) Gn[(size_t)i * k + j] = 0.5 * (G[(size_t)i * kfull + j] + G[(size_t)j * kfull + i]) * dg[i] * dg[j];
  jacobi_eigh(k, Gn, lam, U);
  int k0 = 0;
  while (k0 < k && !(lam[k0] > 1e-6 * lam[k - 1])) ++k0;
  const int kept = k - k0;
  if (kept < 1) return NAN;
  std::vector<double> T((size_t)k * kept), HT((size_t)k * kept, 0.0), Hp((size_t)kept * kept, 0.0), th, S;
  for (int i = 0; i < k; ++i)
    for (int j = 0; j < kept; ++j) T[(size_t)i * kept + j] = dg[i] * U[(size_t)i * k + k0 + j] / sqrt(lam[k0 + j]);
  for (int i = 0; i < k; ++i)
    for (int l = 0; l < k; ++l) {
      const double h = 0.5 * (H[(size_t)i * kfull + l] + H[(size_t)l * kfull + i]);
      for (int j = 0; j < kept; ++j) HT[(size_t)i * kept + j] += h * T[(size_t)l * kept + j];
    }
  for (int j = 0; j < kept; ++j)
    for (int i = 0; i < k; ++i) {
      const double t = T[(size_t)i * kept + j];
      for (int l = 0; l < kept; ++l) Hp[(size_t)j * kept + l] += t * HT[(size_t)i * kept + l];
    }
  jacobi_eigh(kept, Hp, th, S);
  return th[kept - 1];
}

std::atomic<int> g_eig_bound_mode{1};   // 1: the filter's upper end from a Krylov estimate of lambda_max; 0: Gershgorin;
                            // 2 (tests): HALF the estimate, a bound that is certainly short -- the fallback must catch it

struct EigWork {
  float* buf[5];
  double* gpart;
  double* G;
  double* H;
  double* rpart;
  float* wt;        // W^T upload [b x b]
  float* theta;     // [b]
  float* bmax;      // gershgorin partials
  void* opwork;
  size_t opwork_bytes;
  int chunks;
  int64_t rows_per_chunk;
  int rchunks;
  int64_t rrows;
};

int block_size_for(int m, const mgp_lanczos_params_t* p) {
  // The SpMM's cost steps with every 64 columns (one more accumulator per lane), the host side grows with b^3 and,
  // with filter degrees up to 200, a handful of guard vectors is enough: the next multiple of 64 above
  // m + max(m / 8, 12).  Measured at N = 60k, m = 100 (degree cap 200): 112 columns 47 ms, 128: 49 ms, 136: 67 ms,
  // 160: 65 ms, 192: 94 ms; N = 1M, m = 50: 64 columns 0.69 s, 96: 0.82 s, 128: 0.94 s.  (With the degree capped
  // at 80 the same sweep preferred ~2 m columns and took 138 ms / 0.84 s.)
  int b = (p && p->max_basis > 0) ? p->max_basis : m + std::max(m / 8, 12);
  if (b < m + 2) b = m + 2;
  if (!(p && p->max_basis > 0)) {
    b = (b + 63) / 64 * 64;
    if (b > 256 && m + 2 <= 256) b = 256;
  }
  return b;
}

void chunking(int64_t n, int* chunks, int64_t* rpc, int max_chunks, int64_t min_rows) {
  int64_t r = std::max<int64_t>(min_rows, mgp_cdiv(mgp_cdiv(n, max_chunks), 16) * 16);
  *rpc = r;
  *chunks = (int)mgp_cdiv(n, r);
}

size_t eig_bytes(int64_t n, int m, const mgp_lanczos_params_t* p) {
  const int b = block_size_for(m, p);
  int chunks, rch; int64_t rpc, rr;
  chunking(n, &chunks, &rpc, 96, 512);
  chunking(n, &rch, &rr, 256, 256);
  size_t s = 5 * mgp_align((size_t)n * b * sizeof(float));
  s += mgp_align((size_t)chunks * b * b * sizeof(double));
  s += 2 * mgp_align((size_t)b * b * sizeof(double));
  s += mgp_align((size_t)rch * b * sizeof(double));
  s += mgp_align((size_t)b * b * sizeof(float)) + mgp_align(b * sizeof(float)) + mgp_align(1024 * sizeof(float));
  return s + 4096;
}

// 1 (default): the partial Gram blocks on the fp64 matrix cores; 0: fp64 vector FMAs (mgp_gram_set_mfma: A/B, tests)
std::atomic<int> g_gram_mfma{1};

int launch_gram(const float* A, const float* B, int64_t n, int b, EigWork& w, double* out, hipStream_t st) {
  dim3 grid((unsigned)mgp_cdiv(b, 64), (unsigned)mgp_cdiv(b, 64), (unsigned)w.chunks);
  if (g_gram_mfma) hipLaunchKernelGGL(gram_mfma_kernel, grid, dim3(kBlock), 0, st, A, B, n, b, w.rows_per_chunk, w.gpart);
  else hipLaunchKernelGGL(gram_kernel, grid, dim3(kBlock), 0, st, A, B, n, b, w.rows_per_chunk, w.gpart);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)mgp_cdiv((int64_t)b * b, kBlock)), dim3(kBlock), 0, st, w.gpart,
                     w.chunks, b * b, out);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

}  // namespace

int mgp_kernel_block_ld(const float* Z1, int64_t n1, const float* Z2, int64_t n2, int m, float scale, float* K,
                        int64_t ldk, void* stream);

// G = A^T A (b x b, fp64 accumulation of the fp32 entries' exact products) for a tall block A [n, b]:
// the Gram kernel of the eigensolver behind the C-ABI.  rocBLAS' dgemm takes 53 ms for this shape at
// n = 1M, b = 50 (one long K loop per output tile); here the rows are split over up to 512 chunks.
extern "C" size_t mgp_gram_workspace_bytes(int64_t n, int b) {
  if (n <= 0 || b <= 0 || b > 512) return 0;
  int chunks; int64_t rpc;
  chunking(n, &chunks, &rpc, 512, 512);
  return mgp_align((size_t)chunks * b * b * sizeof(double)) + 256;
}

extern "C" int mgp_gram_f64(const float* A, int64_t n, int b, double* G, void* work, size_t work_bytes, void* stream) {
  if (!A || !G || !work || n <= 0 || b <= 0 || b > 512) return MGP_ERR_ARG;
  if (work_bytes < mgp_gram_workspace_bytes(n, b)) return MGP_ERR_WORKSPACE;
  EigWork w{};
  chunking(n, &w.chunks, &w.rows_per_chunk, 512, 512);
  MgpArena ar(work, work_bytes);
  w.gpart = ar.take<double>((size_t)w.chunks * b * b);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  return launch_gram(A, A, n, b, w, G, mgp_stream(stream));
}

extern "C" int mgp_gram_set_mfma(int on) {
  const int prev = g_gram_mfma;
  g_gram_mfma = on ? 1 : 0;
  return prev;
}

// host-only: the small dense symmetric eigensolver used inside the block eigensolver (exported so that
// the CPU test suite can pin it against LAPACK).  A [n x n] row-major; evals [n]; V [n x n] columns.
extern "C" int mgp_host_symeig(int n, const double* A, double* evals, double* V) {
  if (n <= 0 || !A || !evals || !V) return MGP_ERR_ARG;
  std::vector<double> a(A, A + (size_t)n * n), ev, vv;
  HostPool pool(n >= 64 ? host_pool_workers() : 0);      // as inside the block eigensolver
  jacobi_eigh(n, a, ev, vv, &pool);
  memcpy(evals, ev.data(), (size_t)n * sizeof(double));
  memcpy(V, vv.data(), (size_t)n * n * sizeof(double));
  return MGP_OK;
}

extern "C" size_t mgp_lanczos_workspace_bytes(int64_t n, int m, const mgp_lanczos_params_t* p) {
  if (n <= 0 || m <= 0) return 0;
  return eig_bytes(n, m, p);
}

extern "C" int mgp_lanczos_set_bound_mode(int mode) {
  g_eig_bound_mode = mode == 2 ? 2 : (mode ? 1 : 0);
  return MGP_OK;
}

extern "C" int mgp_lanczos_block_size(int m, const mgp_lanczos_params_t* p) { return m > 0 ? block_size_for(m, p) : 0; }

extern "C" int mgp_lanczos_smallest(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals,
                                    float* evecs, float* resid, int32_t* info, void* work, size_t work_bytes,
                                    void* stream) {
  return mgp_lanczos_smallest_ex(L, m, p, evals, evecs, resid, info, nullptr, nullptr, nullptr, work, work_bytes, stream);
}

static int lanczos_smallest_impl(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals, float* evecs, float* resid,
                                 int32_t* info, float* block_evals, float* block_evecs, float* block_resid, const float* warm_block,
                                 const float* warm_evals, void* work, size_t work_bytes, void* stream);

extern "C" int mgp_lanczos_smallest_ex(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals,
                                       float* evecs, float* resid, int32_t* info, float* block_evals, float* block_evecs,
                                       float* block_resid, void* work, size_t work_bytes, void* stream) {
  return lanczos_smallest_impl(L, m, p, evals, evecs, resid, info, block_evals, block_evecs, block_resid, nullptr, nullptr, work,
                               work_bytes, stream);
}

// Warm start (round 5): the reference re-runs the whole eigendecomposition on every eval() (riemann_kernel.py:117-130); while the
// hyper-parameters move a little per step the previous Rayleigh-Ritz block is already close.  warm_block [n, b] (device, the
// block_evecs of an earlier call on the SAME sparsity pattern and row order, b = mgp_lanczos_block_size) replaces the random
// start, warm_evals [b] (host, its Ritz values, ascending) set the first round's filter: damping interval from the block's
// largest Ritz value, degree from the gap behind mode m -- i.e. the first round is already a full-strength round.
extern "C" int mgp_lanczos_smallest_warm(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals, float* evecs,
                                         float* resid, int32_t* info, float* block_evals, float* block_evecs, float* block_resid,
                                         const float* warm_block, const float* warm_evals, void* work, size_t work_bytes,
                                         void* stream) {
  if (!warm_block || !warm_evals) return MGP_ERR_ARG;
  return lanczos_smallest_impl(L, m, p, evals, evecs, resid, info, block_evals, block_evecs, block_resid, warm_block, warm_evals, work,
                               work_bytes, stream);
}

static int lanczos_smallest_impl(const mgp_csr_t* L, int m, const mgp_lanczos_params_t* p, float* evals, float* evecs, float* resid,
                                 int32_t* info, float* block_evals, float* block_evecs, float* block_resid, const float* warm_block,
                                 const float* warm_evals, void* work, size_t work_bytes, void* stream) {
  if (!L || !L->rowptr || !L->col || !L->vals || !L->diag || !evals || !evecs || !work) return MGP_ERR_ARG;
  const int64_t n = L->n;
  if (n <= 0 || m <= 0 || m > n) return MGP_ERR_ARG;
  int b = block_size_for(m, p);
  if (b > n) b = (int)n;
  if (b > 256) return MGP_ERR_UNSUPPORTED;   // one SpMM launch handles <= 256 columns
  if (work_bytes < eig_bytes(n, m, p)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  const float tol = (p && p->tol > 0.f) ? p->tol : 1e-5f;
  // a warm start that has not converged within four rounds was not close enough (the matrix moved too far, or the wanted block
  // sits in a cluster that no start resolves): the solve is then repeated from a cold start, with its own floor exits
  const int max_outer = warm_block ? 4 : ((p && p->max_restarts > 0) ? p->max_restarts : 40);
  const uint64_t seed = p ? p->seed : 1337;

  EigWork w;
  MgpArena ar(work, work_bytes);
  for (int i = 0; i < 5; ++i) w.buf[i] = ar.take<float>((size_t)n * b);
  chunking(n, &w.chunks, &w.rows_per_chunk, 96, 512);
  chunking(n, &w.rchunks, &w.rrows, 256, 256);
  w.gpart = ar.take<double>((size_t)w.chunks * b * b);
  w.G = ar.take<double>((size_t)b * b);
  w.H = ar.take<double>((size_t)b * b);
  w.rpart = ar.take<double>((size_t)w.rchunks * b);
  w.wt = ar.take<float>((size_t)b * b);
  w.theta = ar.take<float>(b);
  w.bmax = ar.take<float>(1024);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;

  // ---- spectrum upper bound (Gershgorin: always >= lambda_max)
  const int gb = (int)std::min<int64_t>(1024, mgp_cdiv(n, kBlock));
  hipLaunchKernelGGL(gershgorin_kernel, dim3(gb), dim3(kBlock), 0, st, n, L->rowptr, L->vals, L->diag, w.bmax);
  MGP_LAUNCH_CHECK();
  std::vector<float> hb(gb);
  MGP_HIP_TRY(hipMemcpyAsync(hb.data(), w.bmax, gb * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  double ub = 0.0;
  for (float v : hb) ub = std::max(ub, (double)v);
  ub *= 1.0 + 1e-6;
  if (!(ub > 0.0)) return MGP_ERR_ARG;

  // ---- the filter's upper end.  Gershgorin is rigorous and loose: on the k-NN graph Laplacians of this package lambda_max
  // is about HALF of it (60k RMNIST-like graph: 15.4 of 29.7; 1M swiss roll: 649 of 1239), and the degree a Chebyshev filter
  // needs grows with sqrt(ub - a) -- a bound twice too large costs 40 % more applies.  So: a 32-dimensional Krylov space of
  // one random vector in the Chebyshev basis of [0, ub] (the same fused three-term SpMV launches as the filter, C = 1; the
  // basis stays bounded and well conditioned, unlike the monomials), Rayleigh-Ritz with the Gram kernels, the largest
  // Ritz value theta_32 <= lambda_max; the filter gets theta_32 + max(3 %, twice what the last 16 dimensions still moved).
  // The residual test keeps the Gershgorin bound as its norm of L, so `tol` means what it meant.  An estimate that fell
  // short would let the filter AMPLIFY the top of the spectrum; that shows as a Ritz value above the supposed bound in
  // the next Rayleigh-Ritz step and is answered there (Gershgorin, fresh block).
  double ubf = ub;
  const int kk = 32;
  const int bound_mode = g_eig_bound_mode;      // (lab switch: read once per call)
  if (bound_mode && b >= kk && n >= 8 * kk) {
    float* K = w.buf[0];
    float* LK = w.buf[1];
    float* KT = w.buf[2];          // the kk Krylov vectors one after the other (b >= kk columns of room): vector j at KT + j n
    const int g1 = (int)std::min<int64_t>(4096, mgp_cdiv(n, kBlock));
    hipLaunchKernelGGL(random_cols_kernel, dim3(g1), dim3(kBlock), 0, st, KT, n, 1, 0, 1, seed ^ 0x5bd1e995ULL);
    MGP_LAUNCH_CHECK();
    const double ce = ub / 2.0;     // centre = half width of [0, ub]
    // T_1 = (L - c) / e
    MGP_TRY(mgp_spmm_fused_ex(L, KT, 1, KT + n, (float)(-1.0), (float)(1.0 / ce), nullptr, nullptr, nullptr, 0.f, 1.f, nullptr,
                              nullptr, nullptr, nullptr, stream));
    for (int j = 2; j < kk; ++j) {
      // T_j = 2 (L - c) / e T_{j-1} - T_{j-2}, written where it stays (round 5: no column copy behind every launch)
      MGP_TRY(mgp_spmm_fused_ex(L, KT + (int64_t)(j - 1) * n, 1, KT + (int64_t)j * n, (float)(-2.0), (float)(2.0 / ce), nullptr, nullptr,
                                KT + (int64_t)(j - 2) * n, -1.f, 1.f, nullptr, nullptr, nullptr, nullptr, stream));
    }
    hipLaunchKernelGGL(vecs_to_cols_kernel, dim3((unsigned)mgp_cdiv(n, 64)), dim3(256), 0, st, KT, n, kk, K);
    MGP_LAUNCH_CHECK();
    MGP_TRY(mgp_spmm_fused_ex(L, K, kk, LK, 0.f, 1.f, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr, nullptr, nullptr, nullptr,
                              stream));
    MGP_TRY(launch_gram(K, K, n, kk, w, w.G, st));
    MGP_TRY(launch_gram(K, LK, n, kk, w, w.H, st));
    std::vector<double> Gk((size_t)kk * kk), Hk((size_t)kk * kk);
    MGP_HIP_TRY(hipMemcpyAsync(Gk.data(), w.G, (size_t)kk * kk * sizeof(double), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipMemcpyAsync(Hk.data(), w.H, (size_t)kk * kk * sizeof(double), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    const double th_full = top_ritz(kk, kk, Gk, Hk), th_half = top_ritz(kk, kk / 2, Gk, Hk);
    if (std::isfinite(th_full) && std::isfinite(th_half) && th_full > 0.0) {
      const double cand = th_full + std::max(0.03 * th_full, 2.0 * fabs(th_full - th_half));
      if (cand < ub) ubf = cand;
      if (bound_mode == 2) ubf = 0.5 * th_full;
    }
    if (getenv("MGP_EIG_TIMING"))
      fprintf(stderr, "[eig] upper end: Gershgorin %.5g, Krylov(32) theta %.5g (16: %.5g) -> filter bound %.5g\n", ub, th_full, th_half, ubf);
  }

  const int rgrid = (int)std::min<int64_t>(4096, mgp_cdiv(n * b, kBlock));
  if (warm_block) {
    MGP_HIP_TRY(hipMemcpyAsync(w.buf[0], warm_block, (size_t)n * b * sizeof(float), hipMemcpyDeviceToDevice, st));
  } else {
    hipLaunchKernelGGL(random_cols_kernel, dim3(rgrid), dim3(kBlock), 0, st, w.buf[0], n, b, 0, b, seed);
    MGP_LAUNCH_CHECK();
  }

  // Buffers: bV / bLV hold the current block V and L V (full width b); three more serve the filter.
  // Soft locking: the leading run of converged Ritz vectors (a multiple of 4 columns) is no longer filtered --
  // the Chebyshev recurrence and the L apply run on the remaining `ba` columns, compacted to an [n, ba]
  // block -- but stays in the Rayleigh-Ritz basis, so it keeps being refined and the block stays orthogonal.
  int bV = 0, bLV = 1, c0 = 2, c1 = 3, c2 = 4;
  int nlock = 0;
  // Degree cap: 200 was tuned with the Gershgorin bound as the filter's upper end (80: 8 rounds / 497 applies at m = 100, 200:
  // 4 / 343, 300: 4 / 443).  What a degree buys goes with 1 / sqrt(ub - a), so under a tighter bound the same filter strength
  // is degree 200 sqrt(ubf / ub) (146 when lambda_max is half of Gershgorin); more only over-solves the last round (60k graph,
  // cap 120 / 146 / 200 / 240: 247 / 273 / 327 / 367 applies for residuals 2.7 / 2.6 / 1.9 / 1.9e-4, tolerance 3.0e-4).
  int kCap = std::max(100, std::min(200, (int)lround(200.0 * sqrt(ubf / ub))));
  double a = ubf / 4.0, a0 = 0.0;
  double top_prev = 1e300;      // largest Ritz value of the previous round's block (they only come down)
  int deg = (p && p->degree > 0) ? p->degree : 10;
  if (warm_block && warm_evals && b > m) {
    // the warm block's own Ritz values stand in for a first Rayleigh-Ritz round (a little margin on the interval's lower end: the
    // matrix has moved since they were computed)
    const double top = (double)warm_evals[b - 1], thm = (double)warm_evals[m - 1];
    if (std::isfinite(top) && top > 0.0 && top < ubf && std::isfinite(thm) && thm < top) {
      a = std::min(0.5 * (top + ubf), 1.02 * top);
      a0 = std::min((double)warm_evals[0], 0.0);
      const double gap = std::max(a - thm, 1e-12 * ub);
      const int dnew = (int)ceil(3.0 / (2.0 * sqrt(gap / (ubf - a))));
      if (!(p && p->degree > 0)) deg = std::min(std::max(dnew, 8), kCap);
    }
  }
  HostPool pool(host_pool_workers());      // lives for this call: joined on every return path
  std::vector<double> G((size_t)b * b), H((size_t)b * b), th, S, lam, U;
  std::vector<float> wt((size_t)b * b), thf(b);
  std::vector<double> rp((size_t)w.rchunks * b), res(b, 1e300);
  int outer = 0, nspmm = 0, nconv = 0, kept = b;
  double rmax_prev1 = 1e300;
  int nconv_prev1 = 0, deg_used = 0;
  bool floor_hit = false;
  // the two early exits below hand the block back with MGP_OK and info[2] < m; they apply only once the largest residual
  // is within 50 x the tolerance asked for, or within 10 x the measured fp32 floor (1.8e-6 ub, see there) for tolerances
  // below it -- a block further out than that is NOT "at the floor" and keeps iterating / ends as MGP_ERR_NOT_CONVERGED
  const double floor_guard = std::max(50.0 * tol, 2e-5);
  auto move_cols = [&](const float* src, int sld, int sc0, int mcols, float* dst, int dld, int dc0) {
    const int grid = (int)std::min<int64_t>(4096, mgp_cdiv(n * mcols, kBlock));
    hipLaunchKernelGGL(move_cols_kernel, dim3(grid), dim3(kBlock), 0, st, src, n, sld, sc0, mcols, dst, dld, dc0);
  };
  for (outer = 0; outer < max_outer; ++outer) {
    // ---- scaled Chebyshev filter of degree `deg` damping [a, ub], normalised at a0, on the active columns
    const auto tr0 = std::chrono::steady_clock::now();
    const int ba = b - nlock;
    deg_used = deg;
    const double e = (ubf - a) / 2.0, c = (ubf + a) / 2.0;
    double sig = e / (a0 - c);
    const double tau = 2.0 / sig;
    int iX = c0, iY = c1, iN = c2;
    move_cols(w.buf[bV], b, nlock, ba, w.buf[iX], ba, 0);
    MGP_LAUNCH_CHECK();
    // Y = (sig/e) (L X - c X)
    MGP_TRY(mgp_spmm_fused_ex(L, w.buf[iX], ba, w.buf[iY], (float)(-c * sig / e), (float)(sig / e), nullptr, nullptr,
                              nullptr, 0.f, 1.f, nullptr, nullptr, nullptr, nullptr, stream));
    ++nspmm;
    for (int i = 2; i <= deg; ++i) {
      const double sn = 1.0 / (tau - sig);
      // Ynew = (2 sn / e) (L Y - c Y) - (sig sn) X
      MGP_TRY(mgp_spmm_fused_ex(L, w.buf[iY], ba, w.buf[iN], (float)(-c * 2.0 * sn / e), (float)(2.0 * sn / e),
                                nullptr, nullptr, w.buf[iX], (float)(-sig * sn), 1.f, nullptr, nullptr, nullptr,
                                nullptr, stream));
      ++nspmm;
      const int t = iX; iX = iY; iY = iN; iN = t;
      sig = sn;
    }
    // filtered active block in iY; L (filtered) into iN; both back into the active columns of V / L V
    MGP_TRY(mgp_spmm_fused_ex(L, w.buf[iY], ba, w.buf[iN], 0.f, 1.f, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr,
                              nullptr, nullptr, nullptr, stream));
    ++nspmm;
    move_cols(w.buf[iY], ba, 0, ba, w.buf[bV], b, nlock);
    move_cols(w.buf[iN], ba, 0, ba, w.buf[bLV], b, nlock);
    MGP_LAUNCH_CHECK();
    const int iF = bV, iLV = bLV, iVn = c0, iLVn = c1;
    // ---- Rayleigh-Ritz: G = V^T V, H = V^T L V (fp64), generalized eigenproblem on the host
    MGP_TRY(launch_gram(w.buf[iF], w.buf[iF], n, b, w, w.G, st));
    MGP_TRY(launch_gram(w.buf[iF], w.buf[iLV], n, b, w, w.H, st));
    const auto tr1 = std::chrono::steady_clock::now();   // everything of this round is queued; the copies below wait for it
    MGP_HIP_TRY(hipMemcpyAsync(G.data(), w.G, (size_t)b * b * sizeof(double), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipMemcpyAsync(H.data(), w.H, (size_t)b * b * sizeof(double), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    auto tp0 = std::chrono::steady_clock::now();
    std::vector<double> dg(b);
    for (int i = 0; i < b; ++i) {
      const double g = G[(size_t)i * b + i];
      if (!(g > 0.0) || !std::isfinite(g)) return MGP_ERR_NOT_CONVERGED;
      dg[i] = 1.0 / sqrt(g);
    }
    std::vector<double> Gn((size_t)b * b);
    for (int i = 0; i < b; ++i)
      for (int j = 0; j < b; ++j) Gn[(size_t)i * b + j] = 0.5 * (G[(size_t)i * b + j] + G[(size_t)j * b + i]) * dg[i] * dg[j];
    // whitening T (b x kept) with T^T G T = I: Cholesky Gn = C C^T, T = D C^-T (0.2 ms); a block that has
    // (nearly) dependent columns -- pivot ratio under 1e-5, i.e. cond(Gn) ~ 1e10 -- takes the rank-revealing
    // eigendecomposition instead (3 ms) and drops the dependent directions
    std::vector<double> T;
    bool tri = true;
    if (cholesky_whiten(b, Gn, dg, T)) {
      kept = b;
    } else {
      tri = false;
      jacobi_eigh(b, Gn, lam, U, &pool);
      const double lmax = lam[b - 1];
      int k0 = 0;
      while (k0 < b && lam[k0] <= 1e-10 * lmax) ++k0;
      kept = b - k0;
      // T = D U[:, k0:] Lambda^-1/2   (b x kept)
      T.assign((size_t)b * kept, 0.0);
      for (int i = 0; i < b; ++i)
        for (int j = 0; j < kept; ++j) T[(size_t)i * kept + j] = dg[i] * U[(size_t)i * b + k0 + j] / sqrt(lam[k0 + j]);
    }
    auto tp1 = std::chrono::steady_clock::now();
    // Hp = T^T Hs T   (rows of the outputs are independent: jobs of 8 rows on the pool, fixed order inside).  After the
    // Cholesky whitening T is upper triangular (tri): T[l][j] = 0 for j < l, which leaves 1/2 and 1/3 of the two products.
    std::vector<double> HT((size_t)b * kept, 0.0), Hp((size_t)kept * kept, 0.0);
    pool.rows(b, 8, [&](int i) {
      double* __restrict__ o = &HT[(size_t)i * kept];
      for (int l = 0; l < b; ++l) {
        const double h = 0.5 * (H[(size_t)i * b + l] + H[(size_t)l * b + i]);
        if (h == 0.0) continue;
        const double* __restrict__ t = &T[(size_t)l * kept];
        for (int j = tri ? l : 0; j < kept; ++j) o[j] += h * t[j];
      }
    });
    pool.rows(kept, 8, [&](int j) {
      double* __restrict__ o = &Hp[(size_t)j * kept];
      const int i1 = tri ? j + 1 : b;
      for (int i = 0; i < i1; ++i) {
        const double t = T[(size_t)i * kept + j];
        const double* __restrict__ h = &HT[(size_t)i * kept];
        for (int l = 0; l < kept; ++l) o[l] += t * h[l];
      }
    });
    auto tp2 = std::chrono::steady_clock::now();
    jacobi_eigh(kept, Hp, th, S, &pool);
    auto tp3 = std::chrono::steady_clock::now();
    // The estimated upper end fell short if a Ritz value lies above it (proof: Ritz values never exceed lambda_max), or if
    // the block's largest Ritz value, which only comes down from round to round, jumps up towards the top of the spectrum
    // (the filter amplified what it should have damped).  Then: Gershgorin from here on and a fresh block.
    if (ubf < ub && (th[kept - 1] > ubf * (1.0 + 1e-3) || (th[kept - 1] > 2.0 * top_prev && th[kept - 1] > 0.25 * ubf))) {
      if (getenv("MGP_EIG_TIMING"))
        fprintf(stderr, "[eig] round %d: largest Ritz value %.5g against the estimated bound %.5g: back to Gershgorin %.5g\n", outer,
                th[kept - 1], ubf, ub);
      ubf = ub;
      kCap = 200;
      hipLaunchKernelGGL(random_cols_kernel, dim3(rgrid), dim3(kBlock), 0, st, w.buf[bV], n, b, 0, b, seed + 104729ULL * (outer + 1));
      MGP_LAUNCH_CHECK();
      nlock = 0;
      a = ub / 4.0;
      a0 = 0.0;
      deg = (p && p->degree > 0) ? p->degree : 10;
      top_prev = 1e300;
      rmax_prev1 = 1e300;
      nconv_prev1 = 0;
      nconv = 0;
      continue;
    }
    top_prev = th[kept - 1];
    // W = T S (b x kept); upload W^T rows = Ritz directions, zero-padded to b
    std::fill(wt.begin(), wt.end(), 0.f);
    pool.rows(b, 8, [&](int i) {
      std::vector<double> acc(kept, 0.0);
      for (int l = tri ? i : 0; l < kept; ++l) {
        const double t = T[(size_t)i * kept + l];
        const double* __restrict__ sr = &S[(size_t)l * kept];
        for (int j = 0; j < kept; ++j) acc[j] += t * sr[j];
      }
      for (int j = 0; j < kept; ++j) wt[(size_t)j * b + i] = (float)acc[j];
    });
    for (int j = 0; j < b; ++j) thf[j] = j < kept ? (float)th[j] : 0.f;
    auto tp4 = std::chrono::steady_clock::now();
    if (getenv("MGP_EIG_TIMING")) {
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      fprintf(stderr, "[eig] round %d: whiten %.2f  HT/Hp %.2f  eigh(Hp) %.2f  W %.2f ms\n", outer, ms(tp0, tp1), ms(tp1, tp2),
              ms(tp2, tp3), ms(tp3, tp4));
    }
    MGP_HIP_TRY(hipMemcpyAsync(w.wt, wt.data(), (size_t)b * b * sizeof(float), hipMemcpyHostToDevice, st));
    MGP_HIP_TRY(hipMemcpyAsync(w.theta, thf.data(), b * sizeof(float), hipMemcpyHostToDevice, st));
    // ---- rotate on the MFMA: Vn = V W, LVn = LV W   (K = Z1 Z2^T with Z2 = W^T)
    MGP_TRY(mgp_kernel_block_ld(w.buf[iF], n, w.wt, b, b, 1.f, w.buf[iVn], b, stream));
    MGP_TRY(mgp_kernel_block_ld(w.buf[iLV], n, w.wt, b, b, 1.f, w.buf[iLVn], b, stream));
    hipLaunchKernelGGL(residual_kernel, dim3(w.rchunks), dim3(kBlock), 0, st, w.buf[iLVn], w.buf[iVn], w.theta, n, b,
                       w.rrows, w.rpart);
    MGP_LAUNCH_CHECK();
    MGP_HIP_TRY(hipMemcpyAsync(rp.data(), w.rpart, (size_t)w.rchunks * b * sizeof(double), hipMemcpyDeviceToHost, st));
    MGP_HIP_TRY(hipStreamSynchronize(st));
    if (getenv("MGP_EIG_TIMING")) {
      auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      const auto tr2 = std::chrono::steady_clock::now();
      fprintf(stderr, "[eig] round %d: enqueue %.2f  filter + Gram on the GPU (wait) %.2f  host %.2f  rotate + residual %.2f ms\n", outer, ms(tr0, tr1), ms(tr1, tp0),
              ms(tp0, tp4), ms(tp4, tr2));
    }
    for (int j = 0; j < b; ++j) {
      double s = 0.0;
      for (int cch = 0; cch < w.rchunks; ++cch) s += rp[(size_t)cch * b + j];
      res[j] = sqrt(s);
    }
    if (kept < b) {   // refill dropped directions with fresh random vectors (their L V column is rebuilt by
                      // the next round's filter: dropped directions sit at the end, locked ones at the start)
      hipLaunchKernelGGL(random_cols_kernel, dim3(rgrid), dim3(kBlock), 0, st, w.buf[iVn], n, b, kept, b,
                         seed + 7919ULL * (outer + 1));
      MGP_LAUNCH_CHECK();
    }
    { const int ov = bV, olv = bLV; bV = iVn; bLV = iLVn; c0 = ov; c1 = olv; }
    nconv = 0;
    nlock = 0;
    if (kept >= m) {
      for (int j = 0; j < m; ++j) nconv += (res[j] <= tol * ub) ? 1 : 0;
      if (getenv("MGP_EIG_TIMING")) {
        int lead = 0;
        while (lead < m && res[lead] <= tol * ub) ++lead;
        double rmx = 0.0;
        for (int j = 0; j < m; ++j) rmx = std::max(rmx, res[j]);
        fprintf(stderr, "[eig] round %d: deg %d, converged %d of %d (leading run %d), max resid %.3e (tol*ub %.3e)\n", outer,
                deg, nconv, m, lead, rmx, tol * ub);
      }
      if (nconv == m) { ++outer; break; }
      // The attainable residual of an fp32 iteration is a few ulp of |L| (the SpMM's own rounding: measured 1.8e-6 ub on
      // the 60k RMNIST-like graph, 7e-8 ub on the smooth modes of the dumbbell): a tolerance under that floor can never
      // be met, and the rounds past it only shuffle round-off (60 rounds / 1.3 s where 5 reach the floor).  With the
      // filter at its degree cap a round multiplies the error of the slowest wanted pair by <= e^-3 unless the gap
      // behind the block is tiny; a round at the cap that does not even halve the largest residual, with no further
      // pair converging, is therefore taken as the floor: the caller gets the block as it stands, the true residuals in
      // `resid`, info[2] = pairs under tol (< m) and MGP_OK.
      {
        double rmx = 0.0;
        for (int j = 0; j < m; ++j) rmx = std::max(rmx, res[j]);
        if (deg_used >= kCap && rmx > 0.5 * rmax_prev1 && nconv <= nconv_prev1 && rmx <= floor_guard * ub) { floor_hit = true; ++outer; break; }
        rmax_prev1 = rmx;
        nconv_prev1 = nconv;
      }
      {
        int lead = 0;
        while (lead < m && res[lead] <= tol * ub) ++lead;
        nlock = lead / 4 * 4;
        if (b - nlock < 8) nlock = 0;
        // the matrix-core tile SpMM serves 48 columns and more: a block locked down to fewer active columns falls back to the gather
        // kernel, whose 28 columns cost MORE per product than 64 on the tiles (1M nodes, b = 64: 0.48 against 0.38 ms) -- keep 48
        if (L->mt_img && b >= 48 && b - nlock < 48) nlock = (b - 48) / 4 * 4;
      }
      a = th[kept - 1];
      a0 = std::min(th[0], 0.0);
      const double gap = std::max(a - th[m - 1], 1e-12 * ub);
      // dunit: the filter degree per unit of damping exponent at the measured gap; dnew: the degree of an e^-3 round (rounds 1-4
      // ran every round at it; the exits below are written in it).
      const double dunit = 1.0 / (2.0 * sqrt(gap / (ubf - a)));
      int dnew = (int)ceil(3.0 * dunit);
      // Round 5: a Rayleigh-Ritz round costs ~2 ms of host + Gram + rotation at b = 128 -- as much as 57 block products at 60k -- so
      // a round should do what the arithmetic allows, and the last one no more than is left to do.  The largest wanted residual falls
      // by ~exp(-t / 2) in a round of exponent t (measured: t = 3 / 4.5 / 9 -> x 0.22 / 0.1 / 0.01), so t_fin = 2 ln(r_max / (0.3 tol
      // ub)) would finish; inside the wanted block the filter lifts mode 1 over mode m by exp(t (sqrt(a - th_1) - sqrt(gap)) /
      // sqrt(gap)), which float32 columns survive up to ~1e4 (t_safe; more and mode m drops under the round-off of mode 1), and 9
      // at most (t_cap, below).  Never less than the e^-3 round.  Conditioned 60k swiss roll: 7 rounds / 105 products / 18.3 ms -> 4 / ~120 /
      // ~13.5 ms; RMNIST-like 60k (ends at the fp32 floor): 4 / 280 / 18.4 ms -> 3 / ~220 / ~14 ms.
      double rmx_w = 0.0;
      for (int j = 0; j < m; ++j) rmx_w = std::max(rmx_w, res[j]);
      const double t_fin = 2.0 * log(std::max(rmx_w, 1e-300) / (0.3 * tol * ub));
      const double s1 = sqrt(std::max(a - th[0], 0.0)), sm = sqrt(gap);
      const double t_safe = s1 > sm ? log(1e4) * sm / (s1 - sm) : 9.0;
      // ... where rounds are expensive against block products: a round is ~57 products at n b = 7.7e6 (60k x 128) but ~5 at
      // 6.4e7 (1M x 64), where the stronger rounds only add products (530 against 438, 229 against 225 ms): the cap goes from 9
      // under n b = 1.6e7 to the e^-3 round at 6.4e7 (logarithmically in between; deterministic, no timing involved).
      const double nb = (double)n * (double)b;
      const double t_cap = nb <= 1.6e7 ? 9.0 : nb >= 6.4e7 ? 3.0 : 9.0 - 6.0 * log(nb / 1.6e7) / log(4.0);
      double target = std::min(std::max(t_fin, 3.0), std::min(t_cap, std::max(t_safe, 3.0)));
      if (const char* e = getenv("MGP_EIG_TARGET")) target = atof(e);       // lab: fixed per-round damping exponent
      const int dask = (int)ceil(target * dunit);
      // degree cap 200 (80 until late in round 1: 8 rounds / 497 applies at m = 100 where 200 needs 4 / 343; the
      // scaled three-term recurrence is normalised at a0, so the block does not overflow at these degrees)
      deg = std::min(std::max(dask, 8), kCap);
      if (getenv("MGP_EIG_TIMING"))
        fprintf(stderr, "[eig] round %d: a %.4e  theta_m %.4e  gap %.3e  exponent %.2f (finish %.2f, safe %.2f)  degree asked %d (e^-3: %d)\n",
                outer, a, th[m - 1], gap, target, t_fin, t_safe, dask, dnew);
      if (p && p->degree > 0) deg = p->degree;
      // The same exit, predicted instead of observed.  A round of degree d multiplies the slowest wanted pair's error by
      // about exp(-3 d / dnew) (dnew = the degree that gives e^-3 at the measured gap between the wanted block and its
      // last guard).  When a round at the cap has been run and the gap asks for more than ~4.3 caps (predicted factor
      // > 0.5: the wanted modes sit in a cluster with their guards -- on the 60k RMNIST-like graph 128 Ritz values lie
      // within 1e-6 lambda_max), every further 200-apply round would buy less than a factor 2: stop before running it
      // (C3 at tol 1e-6: 4 rounds / 50 ms instead of 5 / 67 ms, same residual 2.5e-4 as tol 1e-5 reaches).
      // (Round 5: 0.5 -> 0.36.  The largest RESIDUAL follows the square root of that factor -- see the exponent rule above -- so a
      // cap round predicted at 0.36 improves it by less than 1.7 x for ~7 ms at 60k; and with the stronger early rounds the first
      // cap round is reached a round sooner, at a gap that asks for ~3.8 caps instead of ~5.8: the same block quality -- RMNIST-like
      // 60k: 2.4e-4 after 3 rounds / 249 products against 2.2e-4 after 4 / 280 -- must end the same way.)
      if (!(p && p->degree > 0) && deg_used >= kCap && dnew > kCap && exp(-3.0 * kCap / (double)dnew) > 0.36) {
        double rmx = 0.0;
        for (int j = 0; j < m; ++j) rmx = std::max(rmx, res[j]);
        if (rmx <= floor_guard * ub) { floor_hit = true; ++outer; break; }
      }
    }
  }
  const int cgrid = (int)std::min<int64_t>(4096, mgp_cdiv(n * m, kBlock));
  hipLaunchKernelGGL(copy_cols_kernel, dim3(cgrid), dim3(kBlock), 0, st, w.buf[bV], n, b, m, evecs);
  MGP_LAUNCH_CHECK();
  if (block_evecs) {     // the whole Rayleigh-Ritz block, guard columns included: [n, b] row-major
    const int bgrid = (int)std::min<int64_t>(4096, mgp_cdiv(n * b, kBlock));
    hipLaunchKernelGGL(copy_cols_kernel, dim3(bgrid), dim3(kBlock), 0, st, w.buf[bV], n, b, b, block_evecs);
    MGP_LAUNCH_CHECK();
  }
  MGP_HIP_TRY(hipStreamSynchronize(st));
  for (int j = 0; j < m; ++j) {
    evals[j] = (kept >= m) ? (float)th[j] : 0.f;
    if (resid) resid[j] = (float)res[j];
  }
  for (int j = 0; j < b; ++j) {
    if (block_evals) block_evals[j] = j < kept ? (float)th[j] : 0.f;
    if (block_resid) block_resid[j] = (float)res[j];
  }
  if (warm_block && nconv != m) {
    const int rc = lanczos_smallest_impl(L, m, p, evals, evecs, resid, info, block_evals, block_evecs, block_resid, nullptr, nullptr, work,
                                         work_bytes, stream);
    if (info) { info[0] += outer; info[1] += nspmm; }       // rounds / block products of the abandoned warm attempt included
    return rc;
  }
  if (info) { info[0] = outer; info[1] = nspmm; info[2] = nconv; info[3] = b; }
  if (nconv == m) return MGP_OK;
  return floor_hit ? MGP_OK : MGP_ERR_NOT_CONVERGED;
}

// ================================================================= Lanczos tridiagonalisation
namespace {

// partial[blk][j] = sum_{r in chunk} w[r] * Q[j][r],  j = 0..nq-1   (Q column vectors contiguous)
__global__ __launch_bounds__(kBlock) void lz_dots_kernel(const float* __restrict__ w, const float* __restrict__ Q,
                                                         int64_t n, int nq, int64_t rows_per_block,
                                                         float* __restrict__ partial) {
  __shared__ float sh[kBlock / 64];
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  for (int j = 0; j < nq; ++j) {
    const float* q = Q + (int64_t)j * n;
    float acc = 0.f;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += kBlock) acc = fmaf(w[r], q[r], acc);
    acc = mgp_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * nq + j] = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
  }
}

// h[j] = sum_blk partial[blk][j]; w -= sum_j h[j] Q[j]; alpha_acc += h[nq-1] (block 0); norm partials of new w
__global__ __launch_bounds__(kBlock) void lz_update_kernel(float* __restrict__ w, const float* __restrict__ Q, int64_t n,
                                                           int nq, int64_t rows_per_block, const float* __restrict__ partial,
                                                           int nblk, float* __restrict__ alpha_slot, int accumulate,
                                                           float* __restrict__ norm_partial) {
  extern __shared__ float hs[];   // [nq] + reduction scratch [4]
  float* red = hs + nq;
  for (int j = threadIdx.x; j < nq; j += kBlock) {
    float s = 0.f;
    for (int bI = 0; bI < nblk; ++bI) s += partial[(int64_t)bI * nq + j];
    hs[j] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (accumulate) *alpha_slot += hs[nq - 1]; else *alpha_slot = hs[nq - 1];
  }
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  float nn = 0.f;
  for (int64_t r = r0 + threadIdx.x; r < r1; r += kBlock) {
    float v = w[r];
    for (int j = 0; j < nq; ++j) v = fmaf(-hs[j], Q[(int64_t)j * n + r], v);
    w[r] = v;
    nn = fmaf(v, v, nn);
  }
  nn = mgp_wave_sum(nn);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nn;
  __syncthreads();
  if (threadIdx.x == 0) norm_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// beta = sqrt(sum norm_partial); qnext = w / beta
__global__ __launch_bounds__(kBlock) void lz_normalize_kernel(const float* __restrict__ w, float* __restrict__ qnext,
                                                              int64_t n, const float* __restrict__ norm_partial, int nblk,
                                                              float* __restrict__ beta_slot) {
  __shared__ float sh_beta;
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int bI = 0; bI < nblk; ++bI) s += norm_partial[bI];
    sh_beta = sqrtf(s);
    if (blockIdx.x == 0) *beta_slot = sh_beta;
  }
  __syncthreads();
  const float inv = sh_beta > 0.f ? 1.0f / sh_beta : 0.f;
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
    qnext[r] = w[r] * inv;
}

}  // namespace

// ---------------------------------------------------------------- block of independent Lanczos runs
// P start vectors at once (the probes of the stochastic log-determinant): every vector is an [n, P]
// row-major block, every coefficient a P-vector, the operator apply is ONE P-column SpMM chain.  The runs
// stay independent (no coupling between columns) -- this is batching, not block Lanczos: it divides the
// launch count by P, and the single-vector version is launch-bound (11 launches per step).
namespace {

constexpr int kBlzMaxNq = 48;    // basis vectors kept for re-orthogonalisation (steps + 1 <= this; 48 KB of LDS)
constexpr int kBlzMaxP = 16;

// partial[blk][i][p] = sum_{r in chunk} W[r,p] * Q_i[r,p],  i < nq
__global__ __launch_bounds__(kBlock) void blz_dots_kernel(const float* __restrict__ W, const float* __restrict__ Q,
                                                          int64_t n, int P, int nq, int64_t rows_per_block,
                                                          float* __restrict__ partial) {
  extern __shared__ float sh[];                 // [nq][RL][P]
  const int RL = kBlock / P;
  const int p = threadIdx.x % P, rl = threadIdx.x / P;
  const bool on = rl < RL;
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const int64_t stride = n * P;
  for (int i0 = 0; i0 < nq; i0 += 8) {          // 8 basis vectors per sweep over the chunk (registers)
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
      // two rows x (1 + 8) loads in flight per pass (clamped rows, masked products): with one row per pass the chunk was
      // a chain of dependent round trips (34 us per launch at 60k x 12; the data is 3-60 MB)
      const int nb = nq - i0 < 8 ? nq - i0 : 8;
      for (int64_t rb = r0 + rl; rb < r1; rb += 2 * RL) {
        float w[2], q[2][8];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int64_t r = rb + v * RL < r1 ? rb + v * RL : rb;
          w[v] = W[r * P + p];
#pragma unroll
          for (int u = 0; u < 8; ++u) q[v][u] = Q[(int64_t)(i0 + (u < nb ? u : 0)) * stride + r * P + p];
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const float wv = rb + v * RL < r1 ? w[v] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u < nb) acc[u] = fmaf(wv, q[v][u], acc[u]);
        }
      }
    }
    if (on)
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < nq) sh[((i0 + u) * RL + rl) * P + p] = acc[u];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nq * P; e += kBlock) {
    const int i = e / P, pp = e % P;
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += sh[(i * RL + k) * P + pp];
    partial[((int64_t)blockIdx.x * nq + i) * P + pp] = t;
  }
}

// out[e] = sum_blk partial[blk][e], e < count: four lanes per element (blocks part, part + 4, ...; eight loads in flight
// each), quad xor-sum -- a fixed order.  Every workgroup of blz_update / blz_normalize used to re-reduce all nblk partials
// itself, one serial chain of nblk loads per thread: 71 / 34 us per launch at 60k x 12.
__global__ __launch_bounds__(kBlock) void blz_reduce_kernel(const float* __restrict__ partial, int nblk, int count,
                                                            float* __restrict__ out) {
  const int e = blockIdx.x * (kBlock / 4) + (threadIdx.x >> 2), part = threadIdx.x & 3;
  const int ec = e < count ? e : count - 1;
  float t = 0.f;
  for (int b0 = part; b0 < nblk; b0 += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + 4 * k;
      v[k] = partial[(int64_t)(b < nblk ? b : nblk - 1) * count + ec];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) t += (b0 + 4 * k < nblk) ? v[k] : 0.f;
  }
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  if (part == 0 && e < count) out[e] = t;
}

// h[i][p] (reduced by blz_reduce_kernel); W -= sum_i h[i][p] Q_i; alpha[p] (+)= h[nq-1][p]; norm partials of the new W
__global__ __launch_bounds__(kBlock) void blz_update_kernel(float* __restrict__ W, const float* __restrict__ Q, int64_t n,
                                                            int P, int nq, int64_t rows_per_block,
                                                            const float* __restrict__ hsum,
                                                            float* __restrict__ alpha_row, int accumulate,
                                                            float* __restrict__ norm_partial) {
  extern __shared__ float sh[];                 // h [nq][P], then reduction scratch [RL][P]
  float* h = sh;
  float* red = sh + nq * P;
  const int RL = kBlock / P;
  for (int e = threadIdx.x; e < nq * P; e += kBlock) h[e] = hsum[e];
  __syncthreads();
  if (blockIdx.x == 0 && (int)threadIdx.x < P) {
    const float a = h[(nq - 1) * P + threadIdx.x];
    alpha_row[threadIdx.x] = accumulate ? alpha_row[threadIdx.x] + a : a;
  }
  const int p = threadIdx.x % P, rl = threadIdx.x / P;
  const bool on = rl < RL;
  const int64_t r0 = blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > n) r1 = n;
  const int64_t stride = n * P;
  float nn = 0.f;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      float v = W[r * P + p];
      // eight basis vectors' loads in flight per batch (one at a time: nq dependent round trips per row)
      for (int i0 = 0; i0 < nq; i0 += 8) {
        float q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = Q[(int64_t)(i0 + u < nq ? i0 + u : i0) * stride + r * P + p];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i0 + u < nq) v = fmaf(-h[(i0 + u) * P + p], q[u], v);
      }
      W[r * P + p] = v;
      nn = fmaf(v, v, nn);
    }
  if (on) red[rl * P + p] = nn;
  __syncthreads();
  if ((int)threadIdx.x < P) {
    float t = 0.f;
    for (int k = 0; k < RL; ++k) t += red[k * P + threadIdx.x];
    norm_partial[(int64_t)blockIdx.x * P + threadIdx.x] = t;
  }
}

// beta[p] = sqrt(sum_blk norm_partial[blk][p]); Qnext = W / beta  (a zero column stays zero)
__global__ __launch_bounds__(kBlock) void blz_normalize_kernel(const float* __restrict__ W, float* __restrict__ Qnext,
                                                               int64_t n, int P, const float* __restrict__ norm_sum,
                                                               float* __restrict__ beta_row) {
  __shared__ float inv[kBlzMaxP];
  if ((int)threadIdx.x < P) {
    const float t = norm_sum[threadIdx.x];
    const float b = sqrtf(t);
    inv[threadIdx.x] = b > 0.f ? 1.0f / b : 0.f;
    if (blockIdx.x == 0) beta_row[threadIdx.x] = b;
  }
  __syncthreads();
  const int64_t total = n * P;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
    Qnext[e] = W[e] * inv[e % P];
}

}  // namespace

namespace {

// Buffers of P independent Lanczos runs of `steps` steps on vectors of length n, laid out in one workspace the same way at
// every call (the external-operator form below is stateless: begin / step / end recompute this layout).
struct Blz {
  int64_t n, blockf, nblk, rpb;
  int P, steps, RL, egrid;
  float *Q, *W, *dpart, *npart, *hsum, *nsum, *d_alpha, *d_beta;
  bool ok;
};

size_t blz_bytes(int64_t n, int P, int steps) {
  size_t s = mgp_align((size_t)(steps + 1) * n * P * sizeof(float));   // Q
  s += mgp_align((size_t)n * P * sizeof(float));                       // W
  s += mgp_align((size_t)256 * (steps + 1) * P * sizeof(float));        // dot partials
  s += mgp_align((size_t)256 * P * sizeof(float));                      // norm partials
  s += 2 * mgp_align((size_t)(steps + 1) * P * sizeof(float));          // alpha, beta
  s += mgp_align((size_t)(steps + 1) * P * sizeof(float)) + mgp_align(kBlzMaxP * sizeof(float));   // reduced dots / norms
  return s + 2048;
}

Blz blz_layout(MgpArena& ar, int64_t n, int P, int steps) {
  Blz b;
  b.n = n; b.P = P; b.steps = steps;
  b.Q = ar.take<float>((size_t)(steps + 1) * n * P);
  b.W = ar.take<float>((size_t)n * P);
  b.dpart = ar.take<float>((size_t)256 * (steps + 1) * P);
  b.npart = ar.take<float>((size_t)256 * P);
  b.hsum = ar.take<float>((size_t)(steps + 1) * P);
  b.nsum = ar.take<float>((size_t)kBlzMaxP);
  b.d_alpha = ar.take<float>((size_t)(steps + 1) * P);
  b.d_beta = ar.take<float>((size_t)(steps + 1) * P);
  b.ok = ar.ok();
  b.RL = kBlock / P;
  b.nblk = std::min<int64_t>(256, mgp_cdiv(n, 4 * b.RL));
  if (b.nblk < 1) b.nblk = 1;
  b.rpb = mgp_cdiv(n, b.nblk);
  b.nblk = mgp_cdiv(n, b.rpb);
  b.egrid = (int)std::min<int64_t>(2048, mgp_cdiv(n * P, kBlock));
  b.blockf = n * P;
  return b;
}

void blz_reduce(const Blz& b, const float* part, int count, float* out, hipStream_t st) {
  hipLaunchKernelGGL(blz_reduce_kernel, dim3((unsigned)mgp_cdiv(count, kBlock / 4)), dim3(kBlock), 0, st, part, (int)b.nblk, count, out);
}

// q_0 = Q0 / column norms: one update launch with a zero coefficient gives the norm partials
int blz_begin(const Blz& b, const float* Q0, hipStream_t st) {
  const int P = b.P, steps = b.steps;
  MGP_HIP_TRY(hipMemcpyAsync(b.W, Q0, b.blockf * sizeof(float), hipMemcpyDeviceToDevice, st));
  MGP_HIP_TRY(hipMemcpyAsync(b.Q, Q0, b.blockf * sizeof(float), hipMemcpyDeviceToDevice, st));
  MGP_HIP_TRY(hipMemsetAsync(b.dpart, 0, (size_t)256 * P * sizeof(float), st));
  MGP_HIP_TRY(hipMemsetAsync(b.d_alpha, 0, (size_t)(steps + 1) * P * sizeof(float), st));
  MGP_HIP_TRY(hipMemsetAsync(b.hsum, 0, (size_t)P * sizeof(float), st));
  hipLaunchKernelGGL(blz_update_kernel, dim3((int)b.nblk), dim3(kBlock), (size_t)(P + b.RL * P) * sizeof(float), st, b.W, b.Q, b.n, P, 1,
                     b.rpb, b.hsum, b.d_alpha + (size_t)steps * P, 0, b.npart);
  MGP_LAUNCH_CHECK();
  blz_reduce(b, b.npart, P, b.nsum, st);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(blz_normalize_kernel, dim3(b.egrid), dim3(kBlock), 0, st, b.W, b.Q, b.n, P, b.nsum, b.d_beta + (size_t)steps * P);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// step j on W = A q_j (overwritten): classical Gram-Schmidt against q_0..q_j, twice; alpha_j, beta_j; q_{j+1}
int blz_step(const Blz& b, float* W, int j, hipStream_t st) {
  const int P = b.P, nq = j + 1;
  for (int pass = 0; pass < 2; ++pass) {
    hipLaunchKernelGGL(blz_dots_kernel, dim3((int)b.nblk), dim3(kBlock), (size_t)nq * b.RL * P * sizeof(float), st, W, b.Q, b.n, P,
                       nq, b.rpb, b.dpart);
    MGP_LAUNCH_CHECK();
    blz_reduce(b, b.dpart, nq * P, b.hsum, st);
    MGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(blz_update_kernel, dim3((int)b.nblk), dim3(kBlock), (size_t)(nq * P + b.RL * P) * sizeof(float), st, W,
                       b.Q, b.n, P, nq, b.rpb, b.hsum, b.d_alpha + (size_t)j * P, pass, b.npart);
    MGP_LAUNCH_CHECK();
  }
  blz_reduce(b, b.npart, P, b.nsum, st);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL(blz_normalize_kernel, dim3(b.egrid), dim3(kBlock), 0, st, W, b.Q + (int64_t)(j + 1) * b.blockf, b.n, P, b.nsum,
                     b.d_beta + (size_t)j * P);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

int blz_end(const Blz& b, float* alpha, float* beta, hipStream_t st) {
  MGP_HIP_TRY(hipMemcpyAsync(alpha, b.d_alpha, (size_t)b.steps * b.P * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipMemcpyAsync(beta, b.d_beta, (size_t)b.steps * b.P * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  return MGP_OK;
}

bool blz_shape_ok(int64_t n, int P, int steps) {
  return n > 0 && steps > 0 && steps + 1 <= kBlzMaxNq && P > 0 && P <= kBlzMaxP;
}

}  // namespace

extern "C" size_t mgp_lanczos_tridiag_block_workspace_bytes(const mgp_operator_t* op, int P, int steps) {
  if (!op || !blz_shape_ok(op->L.n, P, steps)) return 0;
  return blz_bytes(op->L.n, P, steps) + mgp_operator_workspace_bytes(op, P) + 2048;
}

// Q0 [n, P] start vectors (columns need not be normalised).  alpha / beta: host [steps][P].
extern "C" int mgp_lanczos_tridiag_block(const mgp_operator_t* op, const float* Q0, int P, int steps, float* alpha,
                                         float* beta, void* work, size_t work_bytes, void* stream) {
  if (!op || !Q0 || !alpha || !beta || !work || steps <= 0 || P <= 0) return MGP_ERR_ARG;
  if (P > kBlzMaxP || steps + 1 > kBlzMaxNq) return MGP_ERR_UNSUPPORTED;
  if (work_bytes < mgp_lanczos_tridiag_block_workspace_bytes(op, P, steps)) return MGP_ERR_WORKSPACE;
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  const Blz b = blz_layout(ar, op->L.n, P, steps);
  const size_t owb = mgp_operator_workspace_bytes(op, P);
  void* ow = ar.take<char>(owb);
  if (!b.ok || !ar.ok()) return MGP_ERR_WORKSPACE;
  MGP_TRY(blz_begin(b, Q0, st));
  for (int j = 0; j < steps; ++j) {
    MGP_TRY(mgp_operator_apply_ex(op, b.Q + (int64_t)j * b.blockf, P, b.W, nullptr, nullptr, nullptr, nullptr, ow, owb, stream));
    MGP_TRY(blz_step(b, b.W, j, st));
  }
  return blz_end(b, alpha, beta, st);
}

// ---- the same P Lanczos runs for an operator the CALLER applies (a wrapper around a Schur complement: every product is a
// CG solve of its own): begin normalises the start block, the caller reads q_j (mgp_blz_q), applies its operator and hands
// W = A q_j to mgp_blz_step (W is overwritten), mgp_blz_end copies alpha / beta [steps][P] to the host and synchronises.
// Nothing in between synchronises or reads anything back.  The workspace is laid out identically at every call.
extern "C" size_t mgp_blz_workspace_bytes(int64_t n, int P, int steps) {
  return blz_shape_ok(n, P, steps) ? blz_bytes(n, P, steps) : 0;
}

extern "C" int mgp_blz_begin(const float* Q0, int64_t n, int P, int steps, void* work, size_t work_bytes, void* stream) {
  if (!Q0 || !work) return MGP_ERR_ARG;
  if (!blz_shape_ok(n, P, steps)) return MGP_ERR_UNSUPPORTED;
  if (work_bytes < blz_bytes(n, P, steps)) return MGP_ERR_WORKSPACE;
  MgpArena ar(work, work_bytes);
  const Blz b = blz_layout(ar, n, P, steps);
  if (!b.ok) return MGP_ERR_WORKSPACE;
  return blz_begin(b, Q0, mgp_stream(stream));
}

extern "C" float* mgp_blz_q(int64_t n, int P, int steps, int j, void* work, size_t work_bytes) {
  if (!work || !blz_shape_ok(n, P, steps) || j < 0 || j > steps || work_bytes < blz_bytes(n, P, steps)) return nullptr;
  MgpArena ar(work, work_bytes);
  const Blz b = blz_layout(ar, n, P, steps);
  return b.ok ? b.Q + (int64_t)j * b.blockf : nullptr;
}

extern "C" int mgp_blz_step(float* W, int64_t n, int P, int steps, int j, void* work, size_t work_bytes, void* stream) {
  if (!W || !work || j < 0 || j >= steps) return MGP_ERR_ARG;
  if (!blz_shape_ok(n, P, steps)) return MGP_ERR_UNSUPPORTED;
  if (work_bytes < blz_bytes(n, P, steps)) return MGP_ERR_WORKSPACE;
  MgpArena ar(work, work_bytes);
  const Blz b = blz_layout(ar, n, P, steps);
  if (!b.ok) return MGP_ERR_WORKSPACE;
  return blz_step(b, W, j, mgp_stream(stream));
}

extern "C" int mgp_blz_end(int64_t n, int P, int steps, float* alpha, float* beta, void* work, size_t work_bytes, void* stream) {
  if (!alpha || !beta || !work) return MGP_ERR_ARG;
  if (!blz_shape_ok(n, P, steps)) return MGP_ERR_UNSUPPORTED;
  if (work_bytes < blz_bytes(n, P, steps)) return MGP_ERR_WORKSPACE;
  MgpArena ar(work, work_bytes);
  const Blz b = blz_layout(ar, n, P, steps);
  if (!b.ok) return MGP_ERR_WORKSPACE;
  return blz_end(b, alpha, beta, mgp_stream(stream));
}

extern "C" size_t mgp_lanczos_tridiag_workspace_bytes(const mgp_operator_t* op, int steps) {
  if (!op || steps <= 0 || op->L.n <= 0) return 0;
  const int64_t n = op->L.n;
  size_t s = mgp_align((size_t)(steps + 1) * n * sizeof(float));   // Q
  s += mgp_align((size_t)n * sizeof(float));                      // w
  s += mgp_operator_workspace_bytes(op, 1);
  s += mgp_align((size_t)512 * (steps + 1) * sizeof(float));       // dot partials
  s += mgp_align(512 * sizeof(float));                            // norm partials
  s += 2 * mgp_align((size_t)(steps + 1) * sizeof(float));         // alpha, beta
  return s + 4096;
}

// q0 [n] start vector (need not be normalised).  alpha[steps], beta[steps] on the host;
// Q_out (nullable, device [steps, n]) receives the orthonormal Lanczos vectors (row j = q_j).
extern "C" int mgp_lanczos_tridiag(const mgp_operator_t* op, const float* q0, int steps, float* alpha, float* beta,
                                   float* Q_out, void* work, size_t work_bytes, void* stream) {
  if (!op || !q0 || !alpha || !beta || !work || steps <= 0) return MGP_ERR_ARG;
  if (work_bytes < mgp_lanczos_tridiag_workspace_bytes(op, steps)) return MGP_ERR_WORKSPACE;
  const int64_t n = op->L.n;
  hipStream_t st = mgp_stream(stream);
  MgpArena ar(work, work_bytes);
  float* Q = ar.take<float>((size_t)(steps + 1) * n);
  float* w = ar.take<float>(n);
  const size_t owb = mgp_operator_workspace_bytes(op, 1);
  void* ow = ar.take<char>(owb);
  float* dpart = ar.take<float>((size_t)512 * (steps + 1));
  float* npart = ar.take<float>(512);
  float* d_alpha = ar.take<float>(steps + 1);
  float* d_beta = ar.take<float>(steps + 1);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  int64_t nblk = std::min<int64_t>(512, mgp_cdiv(n, 1024));
  if (nblk < 1) nblk = 1;
  const int64_t rpb = mgp_cdiv(n, nblk);
  nblk = mgp_cdiv(n, rpb);
  const int egrid = (int)std::min<int64_t>(2048, mgp_cdiv(n, kBlock));

  // q_0 = q0 / ||q0||: reuse the update kernel with nq = 0 to get the norm partials
  MGP_HIP_TRY(hipMemcpyAsync(w, q0, n * sizeof(float), hipMemcpyDeviceToDevice, st));
  MGP_HIP_TRY(hipMemsetAsync(d_alpha, 0, (steps + 1) * sizeof(float), st));
  {
    // norm of w without subtraction: nq = 1 with a zero coefficient is simpler than a new kernel
    MGP_HIP_TRY(hipMemsetAsync(dpart, 0, (size_t)512 * sizeof(float), st));
    MGP_HIP_TRY(hipMemcpyAsync(Q, w, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lz_update_kernel, dim3((int)nblk), dim3(kBlock), (1 + 4) * sizeof(float), st, w, Q, n, 1, rpb,
                       dpart, (int)nblk, d_alpha + steps, 0, npart);
    MGP_LAUNCH_CHECK();
    hipLaunchKernelGGL(lz_normalize_kernel, dim3(egrid), dim3(kBlock), 0, st, w, Q, n, npart, (int)nblk, d_beta + steps);
    MGP_LAUNCH_CHECK();
  }
  for (int j = 0; j < steps; ++j) {
    float* qj = Q + (int64_t)j * n;
    MGP_TRY(mgp_operator_apply_ex(op, qj, 1, w, nullptr, nullptr, nullptr, nullptr, ow, owb, stream));
    for (int pass = 0; pass < 2; ++pass) {   // classical Gram-Schmidt against q_0..q_j, twice
      hipLaunchKernelGGL(lz_dots_kernel, dim3((int)nblk), dim3(kBlock), 0, st, w, Q, n, j + 1, rpb, dpart);
      MGP_LAUNCH_CHECK();
      hipLaunchKernelGGL(lz_update_kernel, dim3((int)nblk), dim3(kBlock), (j + 1 + 4) * sizeof(float), st, w, Q, n,
                         j + 1, rpb, dpart, (int)nblk, d_alpha + j, pass, npart);
      MGP_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(lz_normalize_kernel, dim3(egrid), dim3(kBlock), 0, st, w, Q + (int64_t)(j + 1) * n, n, npart,
                       (int)nblk, d_beta + j);
    MGP_LAUNCH_CHECK();
  }
  MGP_HIP_TRY(hipMemcpyAsync(alpha, d_alpha, steps * sizeof(float), hipMemcpyDeviceToHost, st));
  MGP_HIP_TRY(hipMemcpyAsync(beta, d_beta, steps * sizeof(float), hipMemcpyDeviceToHost, st));
  if (Q_out) MGP_HIP_TRY(hipMemcpyAsync(Q_out, Q, (size_t)steps * n * sizeof(float), hipMemcpyDeviceToDevice, st));
  MGP_HIP_TRY(hipStreamSynchronize(st));
  return MGP_OK;
}
