// Diffusion-maps Laplacian build over the padded symmetric CSR: three gather-only row passes.
//
// Replaces the cached-property chain of manifold_gp/operators/graph_laplacian_operator.py:52-106
// (exp, 4 atomic scatter_add_ passes over int64 COO indices, 4 gathers, sqrt/div/pow passes):
//   pass 1  W_ij = exp(-d2_ij / (4 eps^2));  D~_i = [self_loops] + sum_j W_ij            (:54-69)
//   pass 2  A_ij = W_ij / (D~_i D~_j);       D_i  = [self_loops] D~_i^-2 + sum_j A_ij    (:73-88)
//   pass 3  S_ij = A_ij / (sqrt(D_i) sqrt(D_j)) / eps^2;  diag_i = (1 - D~_i^-2 / D_i) / eps^2
//           (or 1/eps^2 without self loops)                                              (:92-106)
// W is recomputed from d2 in every pass instead of being stored (exp is cheaper than 4 B/entry
// of HBM traffic); row sums are sequential per 16-lane group + shuffle tree, so the result is
// deterministic (the reference's atomic scatter order is not).
// Padding entries carry d2 = +inf -> W = 0 -> contribute nothing.
#include <math.h>
#include "mgp_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int G = 16;  // lanes per row

__device__ __forceinline__ float wexp(float d2, float neg_4eps2) {
  // x.div(-4 eps^2).exp() exactly as graph_laplacian_operator.py:56 (true division, accurate expf)
  return expf(d2 / neg_4eps2);
}

template <int PASS>
__global__ __launch_bounds__(kBlock) void lap_pass(int64_t n, const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col,
                                                   const float* __restrict__ d2, float eps, int self_loops,
                                                   float* __restrict__ dtil, float* __restrict__ deg,
                                                   float* __restrict__ diag, float* __restrict__ dsqrt,
                                                   float* __restrict__ dinvsqrt, float* __restrict__ vals) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) / G;
  const float eps2 = eps * eps;
  const float nq = -4.0f * eps2;
  for (int64_t r = g; r < n; r += ng) {
    const int s = rowptr[r], e = rowptr[r + 1];
    float acc = 0.f;
    float dt_r = 0.f, sq_r = 0.f;
    if (PASS >= 2) dt_r = dtil[r];
    if (PASS == 3) sq_r = dsqrt[r];
    for (int i = s + 4 * lane; i < e; i += 4 * G) {
      const float4 dd = *reinterpret_cast<const float4*>(d2 + i);
      float w[4] = {wexp(dd.x, nq), wexp(dd.y, nq), wexp(dd.z, nq), wexp(dd.w, nq)};
      if (PASS == 1) {
        acc += (w[0] + w[1]) + (w[2] + w[3]);
      } else {
        const int4 c = *reinterpret_cast<const int4*>(col + i);
        const int cc[4] = {c.x, c.y, c.z, c.w};
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = w[t] / (dt_r * dtil[cc[t]]);
        if (PASS == 2) {
          acc += (a[0] + a[1]) + (a[2] + a[3]);
        } else {
          float4 sv;
          sv.x = a[0] / (sq_r * dsqrt[cc[0]]) / eps2;
          sv.y = a[1] / (sq_r * dsqrt[cc[1]]) / eps2;
          sv.z = a[2] / (sq_r * dsqrt[cc[2]]) / eps2;
          sv.w = a[3] / (sq_r * dsqrt[cc[3]]) / eps2;
          *reinterpret_cast<float4*>(vals + i) = sv;
        }
      }
    }
    if (PASS != 3) {
      acc = mgp_group_sum<G>(acc);
      if (lane == 0) {
        if (PASS == 1) {
          dtil[r] = (self_loops ? 1.0f : 0.0f) + acc;
        } else {
          const float base = self_loops ? 1.0f / (dt_r * dt_r) : 0.0f;
          const float d = base + acc;
          deg[r] = d;
          const float sq = sqrtf(d);
          dsqrt[r] = sq;
          dinvsqrt[r] = 1.0f / sq;
          diag[r] = self_loops ? (1.0f - (1.0f / (dt_r * dt_r)) * (1.0f / d)) / eps2 : 1.0f / eps2;
        }
      }
    }
  }
}

__global__ void edge_values_kernel(const int32_t* __restrict__ tr, const int32_t* __restrict__ tc,
                                   const float* __restrict__ tv, int64_t M, const float* __restrict__ dtil,
                                   const float* __restrict__ deg, float eps, int which, float* __restrict__ out) {
  const float eps2 = eps * eps;
  const float nq = -4.0f * eps2;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < M;
       e += (int64_t)gridDim.x * blockDim.x) {
    float v = wexp(tv[e], nq);
    if (which >= 1) v = v / (dtil[tr[e]] * dtil[tc[e]]);
    if (which >= 2) v = v / (sqrtf(deg[tr[e]]) * sqrtf(deg[tc[e]])) / eps2;
    out[e] = v;
  }
}

// ---------------------------------------------------------------- forward-mode tangent wrt eps
// d/d eps of every quantity mgp_laplacian_build produces, by the same three gather-only row passes:
//   dW  = W d2 / (2 eps^3)
//   dD~ = sum_j dW                                   (pass 1)
//   dA  = dW / (D~_i D~_j) - A (dD~_i / D~_i + dD~_j / D~_j)
//   dD  = [self_loops] (-2 D~^-3 dD~) + sum_j dA      (pass 2)
//   dS  = (dA - A (dD_i / (2 D_i) + dD_j / (2 D_j))) / (sqrt(D_i) sqrt(D_j) eps^2) - 2 S / eps   (pass 3)
//   ddiag = [self_loops] (2 D~^-3 dD~ / D + D~^-2 dD / D^2) / eps^2 - 2 diag / eps
// The gradient of any scalar loss wrt the graph bandwidth is then <dl/dvals, dvals> + <dl/ddiag, ddiag>
// + ... ; in particular d(u^T L v)/d eps = u^T L' v with L' the CSR (dvals, ddiag) -- one more SpMV
// (manifold_gp/operators/graph_laplacian_operator.py relies on autograd through its torch ops for
// this, pinned by test/_test_functions.py:59-74 `test_grad`).
template <int PASS>
__global__ __launch_bounds__(kBlock) void lap_tangent_pass(int64_t n, const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, const float* __restrict__ d2,
                                                           float eps, int self_loops, const float* __restrict__ dtil,
                                                           const float* __restrict__ deg, const float* __restrict__ diag,
                                                           float* __restrict__ ddtil, float* __restrict__ ddeg,
                                                           float* __restrict__ ddiag, float* __restrict__ ddsqrt,
                                                           float* __restrict__ ddinvsqrt, float* __restrict__ dvals) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t g = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / G;
  const int64_t ng = ((int64_t)gridDim.x * blockDim.x) / G;
  const float eps2 = eps * eps;
  const float nq = -4.0f * eps2;
  const float dwf = 1.0f / (2.0f * eps2 * eps);
  for (int64_t r = g; r < n; r += ng) {
    const int s = rowptr[r], e = rowptr[r + 1];
    float acc = 0.f;
    const float dt_r = dtil[r];
    float ddt_r = 0.f, d_r = 0.f, dd_r = 0.f;
    if (PASS >= 2) ddt_r = ddtil[r];
    if (PASS == 3) { d_r = deg[r]; dd_r = ddeg[r]; }
    for (int i = s + lane; i < e; i += G) {
      const float dd = d2[i];
      const float w = wexp(dd, nq);
      const float dw = isinf(dd) ? 0.f : w * dd * dwf;
      if (PASS == 1) {
        acc += dw;
      } else {
        const int c = col[i];
        const float dt_c = dtil[c], ddt_c = ddtil[c];
        const float a = w / (dt_r * dt_c);
        const float da = dw / (dt_r * dt_c) - a * (ddt_r / dt_r + ddt_c / dt_c);
        if (PASS == 2) {
          acc += da;
        } else {
          const float d_c = deg[c], dd_c = ddeg[c];
          const float sq = sqrtf(d_r) * sqrtf(d_c);
          const float sv = a / sq / eps2;
          dvals[i] = (da - a * (0.5f * dd_r / d_r + 0.5f * dd_c / d_c)) / sq / eps2 - 2.0f * sv / eps;
        }
      }
    }
    if (PASS != 3) {
      acc = mgp_group_sum<G>(acc);
      if (lane == 0) {
        if (PASS == 1) {
          ddtil[r] = acc;
        } else {
          const float i2 = 1.0f / (dt_r * dt_r);
          const float dbase = self_loops ? -2.0f * i2 / dt_r * ddt_r : 0.0f;
          const float d = deg[r];
          const float ddv = dbase + acc;
          ddeg[r] = ddv;
          const float sq = sqrtf(d);
          ddsqrt[r] = 0.5f * ddv / sq;
          ddinvsqrt[r] = -0.5f * ddv / (d * sq);
          ddiag[r] = self_loops ? (2.0f * i2 / dt_r * ddt_r / d + i2 * ddv / (d * d)) / eps2 - 2.0f * diag[r] / eps
                                : -2.0f / (eps2 * eps);
        }
      }
    }
  }
}

}  // namespace

extern "C" int mgp_laplacian_tangent(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2, float eps,
                                     int self_loops, const float* degree_unnorm, const float* degree,
                                     const float* diag, float* d_degree_unnorm, float* d_degree, float* d_diag,
                                     float* d_dsqrt, float* d_dinvsqrt, float* d_vals, void* stream) {
  if (!rowptr || !col || !d2 || !degree_unnorm || !degree || !diag || !d_degree_unnorm || !d_degree || !d_diag ||
      !d_dsqrt || !d_dinvsqrt || !d_vals)
    return MGP_ERR_ARG;
  if (n <= 0 || !(eps > 0.f)) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  int64_t grid = mgp_cdiv(n, kBlock / G);
  if (grid > 4096) grid = 4096;
#define MGP_TAN(PASS)                                                                                              \
  hipLaunchKernelGGL((lap_tangent_pass<PASS>), dim3((int)grid), dim3(kBlock), 0, st, n, rowptr, col, d2, eps,      \
                     self_loops, degree_unnorm, degree, diag, d_degree_unnorm, d_degree, d_diag, d_dsqrt,          \
                     d_dinvsqrt, d_vals);                                                                          \
  MGP_LAUNCH_CHECK();
  MGP_TAN(1)
  MGP_TAN(2)
  MGP_TAN(3)
#undef MGP_TAN
  return MGP_OK;
}

extern "C" int mgp_laplacian_build(int64_t n, const int32_t* rowptr, const int32_t* col, const float* d2,
                                   float eps, int self_loops, float* degree_unnorm, float* degree,
                                   float* diag, float* dsqrt, float* dinvsqrt, float* vals, void* stream) {
  if (!rowptr || !col || !d2 || !degree_unnorm || !degree || !diag || !dsqrt || !dinvsqrt || !vals)
    return MGP_ERR_ARG;
  if (n <= 0 || !(eps > 0.f)) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  int64_t groups_per_block = kBlock / G;
  int64_t grid = mgp_cdiv(n, groups_per_block);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL((lap_pass<1>), dim3((int)grid), dim3(kBlock), 0, st, n, rowptr, col, d2, eps, self_loops,
                     degree_unnorm, degree, diag, dsqrt, dinvsqrt, vals);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((lap_pass<2>), dim3((int)grid), dim3(kBlock), 0, st, n, rowptr, col, d2, eps, self_loops,
                     degree_unnorm, degree, diag, dsqrt, dinvsqrt, vals);
  MGP_LAUNCH_CHECK();
  hipLaunchKernelGGL((lap_pass<3>), dim3((int)grid), dim3(kBlock), 0, st, n, rowptr, col, d2, eps, self_loops,
                     degree_unnorm, degree, diag, dsqrt, dinvsqrt, vals);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// ---- the reductions of the fused SpMM's backward pass (autograd._FusedSpmm.backward) in one launch.  With h = cov post (.) g the
// upstream gradient and xs = pre (.) X the scaled input, the parameter gradients are inner products of [n, C] blocks:
//     d/d eps  = b <h, L' xs>      d/d a = <h, xs>      d/d pre[r] = sum_c gxs[r, c] X[r, c]      d/d post[r] = cov sum_c g[r, c] (a xs + b L xs)[r, c]
// -- as torch ops a product, a sum and a scalar fix-up each (a dozen small launches per differentiable SpMM, 12 of those per
// supervised epoch, every one issued by a host that is the bottleneck of that epoch).  A thread owns a row; the two scalars are
// block-reduced in a fixed order into partial[block][2] (the caller sums the few hundred partials).  Null inputs switch the
// corresponding output off.
namespace {

__global__ __launch_bounds__(kBlock) void backward_sums_kernel(int64_t n, int C, const float* __restrict__ h, const float* __restrict__ dlx,
                                                               const float* __restrict__ xs, const float* __restrict__ gxs,
                                                               const float* __restrict__ X, const float* __restrict__ g,
                                                               const float* __restrict__ lx, float av, float bv, float cov,
                                                               float* __restrict__ partial, float* __restrict__ gpre,
                                                               float* __restrict__ gpost) {
  __shared__ float sh[2][kBlock / 64];
  float s0 = 0.f, s1 = 0.f;
  for (int64_t r = blockIdx.x * (int64_t)kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
    const int64_t o = r * C;
    float p0 = 0.f, p1 = 0.f;
    for (int c = 0; c < C; ++c) {
      const float hv = h ? h[o + c] : 0.f;
      const float xv = xs ? xs[o + c] : 0.f;
      if (dlx) s0 = fmaf(hv, dlx[o + c], s0);
      if (h && xs) s1 = fmaf(hv, xv, s1);
      if (gpre) p0 = fmaf(gxs[o + c], X[o + c], p0);
      if (gpost) p1 = fmaf(g[o + c], fmaf(av, xv, bv * lx[o + c]), p1);
    }
    if (gpre) gpre[r] = p0;
    if (gpost) gpost[r] = cov * p1;
  }
  for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s1 += __shfl_xor(s1, off, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s0; sh[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    partial[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

}  // namespace

extern "C" int mgp_spmm_backward_blocks(int64_t n) {
  if (n <= 0) return 0;
  const int64_t b = mgp_cdiv(n, kBlock);
  return (int)(b < 512 ? b : 512);
}

// see include/mgp_hip.h
extern "C" int mgp_spmm_backward_sums(int64_t n, int C, const float* h, const float* dlx, const float* xs, const float* gxs,
                                      const float* X, const float* g, const float* lx, float av, float bv, float cov,
                                      float* partial, float* gpre, float* gpost, void* stream) {
  if (n <= 0 || C <= 0 || !partial) return MGP_ERR_ARG;
  if (gpre && (!gxs || !X)) return MGP_ERR_ARG;
  if (gpost && (!g || !xs || !lx)) return MGP_ERR_ARG;
  if (dlx && !h) return MGP_ERR_ARG;
  hipLaunchKernelGGL(backward_sums_kernel, dim3(mgp_spmm_backward_blocks(n)), dim3(kBlock), 0, mgp_stream(stream), n, C, h, dlx, xs, gxs, X,
                     g, lx, av, bv, cov, partial, gpre, gpost);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" int mgp_edge_values(const int32_t* tri_row, const int32_t* tri_col, const float* tri_val, int64_t M,
                               const float* degree_unnorm, const float* degree, float eps, int which,
                               float* out, void* stream) {
  if (!tri_row || !tri_col || !tri_val || !out || which < 0 || which > 2) return MGP_ERR_ARG;
  if (which >= 1 && !degree_unnorm) return MGP_ERR_ARG;
  if (which >= 2 && !degree) return MGP_ERR_ARG;
  if (M <= 0) return M == 0 ? MGP_OK : MGP_ERR_ARG;
  int64_t grid = mgp_cdiv(M, kBlock);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(edge_values_kernel, dim3((int)grid), dim3(kBlock), 0, mgp_stream(stream), tri_row, tri_col,
                     tri_val, M, degree_unnorm, degree, eps, which, out);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}
