// Precision-side operator family as chains of the fused SpMM (see include/mgp_hip.h).
//
//   Q2 = scale * diag(post) (tau I + L_sym)^nu diag(pre)     precision_matern_operator.py:26-37,
//                                                            scale_wrapper_operator.py:27
//   form 0: A = Q2
//   form 1: A = Q2 (I - s Q2 (I - s Q2))                     noise_wrapper_operator.py:22
//   form 2: A = I + s Q2                                     (K + s I) system, K = Q2^-1
//
// The reference runs nu x (2 spmm + ~5 elementwise) launches per Q application and allocates a
// fresh [N,C] tensor for each; here one Q application is nu launches that ping-pong between two
// workspace buffers, the axpy of the wrappers rides in the epilogue of the last launch
// (base / cb / co), and the dot product CG needs rides along as per-workgroup partials.
#include <rccl/rccl.h>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

struct Hooks {
  const float* dotw;
  float* dot_partials;
  const int* skip;
  int* tick;
  // init-free CG solve: launch 0 copies its raw input rows to copy_x and leaves its launch record; the last launch
  // writes the sum dotw^2 partials and resets the iteration state (all nullable / 0)
  float* copy_x;
  void* record;
  float* dot2_partials;
  int tick_reset;
};

}  // namespace

// in-place all-gather of `count_per_rank` floats per rank (slice p at buf + p * count_per_rank)
int mgp_dist_allgather_f32(const MgpDist* d, float* buf, int64_t count_per_rank, void* stream) {
  ncclResult_t r = ncclAllGather(buf + (int64_t)d->rank * count_per_rank, buf, (size_t)count_per_rank, ncclFloat,
                                 static_cast<ncclComm_t>(d->comm), mgp_stream(stream));
  return r == ncclSuccess ? MGP_OK : 1000 + (int)r;
}

namespace {

// Y = cb * base + co * Q2 X   (base nullable).  t0/t1: [n*C] scratch, distinct from X and Y.
// Xs (nullable): diag(pre) X already formed by the producer of X (saves the second gather per entry)
// d (nullable): row partition -- each launch computes the local rows, then the slices are gathered
int q2_chain(const mgp_operator_t* op, const MgpDist* d, int nb_loc, const float* X, const float* Xs, int C,
             float* Y, const float* base, float cb, float co, float* t0, float* t1, const Hooks* hk, void* stream) {
  const float tau = 2.0f * (float)op->nu / (op->kappa * op->kappa);
  const float* in = Xs ? Xs : X;
  for (int s = 0; s < op->nu; ++s) {
    const bool first = (s == 0), last = (s == op->nu - 1);
    float* out = last ? Y : ((s & 1) ? t1 : t0);
    // (x + delta L x) / delta == tau x + L x  with delta = 1/tau (precision_matern_operator.py:31-33)
    float* dp = (last && hk) ? hk->dot_partials : nullptr;
    if (dp && d) dp += (int64_t)d->rank * nb_loc * C;      // this rank's segment of the gathered partials
    MgpFirst fst{(first && hk) ? hk->copy_x : nullptr, (last && hk) ? hk->dot2_partials : nullptr,
                 (last && hk) ? hk->tick_reset : 0, (first && hk) ? hk->record : nullptr};
    const bool use_fst = fst.copy_x || fst.dot2_partials || fst.tick_reset || fst.record;
    MGP_TRY(mgp_spmm_fused_first(&op->L, d ? d->row_offset : 0, in, C, out, tau, 1.0f,
                                 (first && !Xs) ? op->pre : nullptr, last ? op->post : nullptr,
                                 last ? base : nullptr, cb, last ? co * op->scale : 1.0f,
                                 (last && hk) ? hk->dotw : nullptr, dp, hk ? hk->skip : nullptr,
                                 (last && hk) ? hk->tick : nullptr, use_fst ? &fst : nullptr, stream));
    if (d) {
      // the collectives run unconditionally (also after convergence) so that every rank issues
      // the same sequence; a skipped launch leaves stale but finite data behind them
      // an open group is always closed before an error is returned: a rank that left it open would hang or
      // mis-order its next collective
      int rc = MGP_OK;
      if (dp) {
        const ncclResult_t gs = ncclGroupStart();
        if (gs != ncclSuccess) return 1000 + (int)gs;
      }
      rc = mgp_dist_allgather_f32(d, out, d->n_loc * C, stream);
      if (dp) {
        if (rc == MGP_OK) rc = mgp_dist_allgather_f32(d, hk->dot_partials, (int64_t)nb_loc * C, stream);
        const ncclResult_t ge = ncclGroupEnd();
        if (rc == MGP_OK && ge != ncclSuccess) rc = 1000 + (int)ge;
      }
      MGP_TRY(rc);
    }
    in = out;
  }
  return MGP_OK;
}

int check_op(const mgp_operator_t* op) {
  if (!op || !op->L.rowptr || !op->L.col || !op->L.vals || !op->L.diag) return MGP_ERR_ARG;
  if (op->L.n <= 0 || op->nu < 1 || op->nu > 16 || !(op->kappa > 0.f)) return MGP_ERR_ARG;
  if (op->form < 0 || op->form > 2) return MGP_ERR_ARG;
  return MGP_OK;
}

}  // namespace

extern "C" size_t mgp_operator_workspace_bytes(const mgp_operator_t* op, int C) {
  if (check_op(op) != MGP_OK || C <= 0) return 0;
  return 4 * mgp_align((size_t)op->L.n * C * sizeof(float)) + 256;
}

int mgp_operator_apply_ex(const mgp_operator_t* op, const float* X, int C, float* Y, const float* dotw,
                          float* dot_partials, const int* skip, int* tick, void* work, size_t work_bytes,
                          void* stream) {
  return mgp_operator_apply_ex2(op, X, nullptr, C, Y, dotw, dot_partials, skip, tick, work, work_bytes, stream);
}

int mgp_operator_apply_ex2(const mgp_operator_t* op, const float* X, const float* Xs, int C, float* Y,
                           const float* dotw, float* dot_partials, const int* skip, int* tick, void* work,
                           size_t work_bytes, void* stream) {
  return mgp_operator_apply_dist(op, nullptr, X, Xs, C, Y, dotw, dot_partials, 0, skip, tick, work, work_bytes,
                                 stream);
}

// work must hold 4 buffers of the GLOBAL vector length (world * n_loc * C floats) when d != NULL
int mgp_operator_apply_dist(const mgp_operator_t* op, const MgpDist* d, const float* X, const float* Xs, int C,
                            float* Y, const float* dotw, float* dot_partials, int nb_loc, const int* skip, int* tick,
                            void* work, size_t work_bytes, void* stream) {
  MGP_TRY(check_op(op));
  if (!X || !Y || X == Y || C <= 0) return MGP_ERR_ARG;
  const int64_t nrows = d ? d->n_loc * d->world : op->L.n;
  if (!work || work_bytes < 4 * mgp_align((size_t)nrows * C * sizeof(float))) return MGP_ERR_WORKSPACE;
  MgpArena ar(work, work_bytes);
  const size_t nc = (size_t)nrows * C;
  float* t0 = ar.take<float>(nc);
  float* t1 = ar.take<float>(nc);
  float* ua = ar.take<float>(nc);
  float* ub = ar.take<float>(nc);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  Hooks hk{dotw, dot_partials, skip, tick, nullptr, nullptr, nullptr, 0};
  Hooks hk_mid{nullptr, nullptr, skip, nullptr, nullptr, nullptr, nullptr, 0};
  switch (op->form) {
    case 0:
      return q2_chain(op, d, nb_loc, X, Xs, C, Y, nullptr, 0.f, 1.f, t0, t1, &hk, stream);
    case 2:
      return q2_chain(op, d, nb_loc, X, Xs, C, Y, X, 1.f, op->noise, t0, t1, &hk, stream);
    case 1:
      // Q(v - s Q(v - s Q v))
      MGP_TRY(q2_chain(op, d, nb_loc, X, Xs, C, ua, X, 1.f, -op->noise, t0, t1, &hk_mid, stream));
      MGP_TRY(q2_chain(op, d, nb_loc, ua, nullptr, C, ub, X, 1.f, -op->noise, t0, t1, &hk_mid, stream));
      return q2_chain(op, d, nb_loc, ub, nullptr, C, Y, nullptr, 0.f, 1.f, t0, t1, &hk, stream);
  }
  return MGP_ERR_ARG;
}

// First operator apply of an init-free CG solve (cg.hip), C == 1, tile kernel, single-chain forms (0 and 2):
// w = A rhs without a preceding cg_init launch.  Launch 0 reads the caller's right-hand side (scaled by op->pre
// inside the kernel) and copies the raw rows to r_copy; the last launch takes r_copy as its base / dot weight -- or
// rhs itself when the chain has a single launch, where the copy is written by the same kernel -- and leaves the
// partials of r . A r and of ||r||^2 and the iteration state {1, 0, 0}.  No skip flag: this apply opens the solve.
// `record`: launch arguments of launch 0 (mgp_spmm_patch_node re-points them at the next solve's rhs).
int mgp_operator_apply_first(const mgp_operator_t* op, const float* rhs, float* r_copy, float* Y, float* dot_partials,
                             float* dot2_partials, int* state, void* record, void* work, size_t work_bytes, void* stream) {
  MGP_TRY(check_op(op));
  if (!rhs || !r_copy || !Y || !dot_partials || !dot2_partials || !state || rhs == Y) return MGP_ERR_ARG;
  if (op->form != 0 && op->form != 2) return MGP_ERR_UNSUPPORTED;
  if (!work || work_bytes < 4 * mgp_align((size_t)op->L.n * sizeof(float))) return MGP_ERR_WORKSPACE;
  MgpArena ar(work, work_bytes);
  const size_t nc = (size_t)op->L.n;
  float* t0 = ar.take<float>(nc);
  float* t1 = ar.take<float>(nc);
  if (!ar.ok()) return MGP_ERR_WORKSPACE;
  const float* rv = op->nu == 1 ? rhs : r_copy;      // what the last launch reads as r
  Hooks hk{rv, dot_partials, nullptr, state, r_copy, record, dot2_partials, 1};
  if (op->form == 0) return q2_chain(op, nullptr, 0, rhs, nullptr, 1, Y, nullptr, 0.f, 1.f, t0, t1, &hk, stream);
  return q2_chain(op, nullptr, 0, rhs, nullptr, 1, Y, rv, 1.f, op->noise, t0, t1, &hk, stream);
}

// ---------------------------------------------------------------- fp64 apply (iterative refinement only)
// Y = cb base + co post (.) (a xs + b (diag (.) xs - S xs)), xs = pre (.) X, everything accumulated in
// fp64 from the fp32 matrix: used once per refinement round of the CG to form the TRUE residual, so that
// the refined solution is accurate to fp32 round-off in the FORWARD error even when cond(A) eps32 is
// large (N = 1M swiss roll: 1e-3).  One lane per (row, column); not a hot kernel.
namespace {

__global__ void spmm_f64_kernel(int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                const float* __restrict__ vals, const float* __restrict__ diag,
                                const double* __restrict__ X, double* __restrict__ Y, int C, double a, double b,
                                const float* __restrict__ pre, const float* __restrict__ post,
                                const double* __restrict__ base, double cb, double co) {
  const int64_t total = n * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i % C);
    double acc = 0.0;
    for (int e = rowptr[r], e1 = rowptr[r + 1]; e < e1; ++e) {
      const int j = col[e];
      double xj = X[(int64_t)j * C + c];
      if (pre) xj *= (double)pre[j];
      acc += (double)vals[e] * xj;
    }
    double xs = X[i];
    if (pre) xs *= (double)pre[r];
    double t = a * xs + b * ((double)diag[r] * xs - acc);
    if (post) t *= (double)post[r];
    Y[i] = (base ? cb * base[i] : 0.0) + co * t;
  }
}

// C == 1: G lanes per row (entries strided over the lanes, coalesced col / value loads, G gathers of X in flight per
// row), xor-shuffle tree in fp64.  The one-lane-per-row kernel above walks a row serially: 2.4 ms per launch on the
// 1M-node k = 64 graph, 8 launches per refined solve = 14 % of the 136 ms S5 solve (profiles/r02_s5_kernel_stats.csv).
template <int G>
__global__ __launch_bounds__(256) void spmv_f64_kernel(int64_t n, const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, const float* __restrict__ vals,
                                                       const float* __restrict__ diag, const double* __restrict__ X,
                                                       double* __restrict__ Y, double a, double b,
                                                       const float* __restrict__ pre, const float* __restrict__ post,
                                                       const double* __restrict__ base, double cb, double co) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t rows_per_pass = (int64_t)gridDim.x * (256 / G);
  for (int64_t r = (int64_t)blockIdx.x * (256 / G) + threadIdx.x / G; r < n; r += rows_per_pass) {
    const int e0 = rowptr[r], e1 = rowptr[r + 1];
    double acc = 0.0;
    for (int e = e0 + lane; e < e1; e += G) {
      const int j = col[e];
      double xj = X[j];
      if (pre) xj *= (double)pre[j];
      acc += (double)vals[e] * xj;
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
    if (lane == 0) {
      double xs = X[r];
      if (pre) xs *= (double)pre[r];
      double t = a * xs + b * ((double)diag[r] * xs - acc);
      if (post) t *= (double)post[r];
      Y[r] = (base ? cb * base[r] : 0.0) + co * t;
    }
  }
}

int q2_chain_f64(const mgp_operator_t* op, const double* X, int C, double* Y, const double* base, double cb, double co,
                 double* t0, double* t1, hipStream_t st) {
  const double tau = 2.0 * (double)op->nu / ((double)op->kappa * (double)op->kappa);
  const int64_t n = op->L.n;
  int64_t grid = mgp_cdiv(n * C, 256);
  if (grid > 65535 * 4) grid = 65535 * 4;
  const double* in = X;
  for (int s = 0; s < op->nu; ++s) {
    const bool first = (s == 0), last = (s == op->nu - 1);
    double* out = last ? Y : ((s & 1) ? t1 : t0);
    if (C == 1) {
      int64_t g1 = mgp_cdiv(n, 16);
      if (g1 > 65535 * 4) g1 = 65535 * 4;
      hipLaunchKernelGGL((spmv_f64_kernel<16>), dim3((unsigned)g1), dim3(256), 0, st, n, op->L.rowptr, op->L.col,
                         op->L.vals, op->L.diag, in, out, tau, 1.0, first ? op->pre : nullptr, last ? op->post : nullptr,
                         last ? base : nullptr, cb, last ? co * (double)op->scale : 1.0);
    } else {
      hipLaunchKernelGGL(spmm_f64_kernel, dim3((unsigned)grid), dim3(256), 0, st, n, op->L.rowptr, op->L.col, op->L.vals,
                         op->L.diag, in, out, C, tau, 1.0, first ? op->pre : nullptr, last ? op->post : nullptr,
                         last ? base : nullptr, cb, last ? co * (double)op->scale : 1.0);
    }
    MGP_LAUNCH_CHECK();
    in = out;
  }
  return MGP_OK;
}

}  // namespace

// work64: 4 buffers of n * C doubles
int mgp_operator_apply_f64(const mgp_operator_t* op, const double* X, int C, double* Y, double* work64, void* stream) {
  MGP_TRY(check_op(op));
  if (!X || !Y || !work64 || C <= 0) return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  const size_t nc = (size_t)op->L.n * C;
  double *t0 = work64, *t1 = work64 + nc, *ua = work64 + 2 * nc, *ub = work64 + 3 * nc;
  const double noise = (double)op->noise;
  switch (op->form) {
    case 0: return q2_chain_f64(op, X, C, Y, nullptr, 0.0, 1.0, t0, t1, st);
    case 2: return q2_chain_f64(op, X, C, Y, X, 1.0, noise, t0, t1, st);
    case 1:
      MGP_TRY(q2_chain_f64(op, X, C, ua, X, 1.0, -noise, t0, t1, st));
      MGP_TRY(q2_chain_f64(op, ua, C, ub, X, 1.0, -noise, t0, t1, st));
      return q2_chain_f64(op, ub, C, Y, nullptr, 0.0, 1.0, t0, t1, st);
  }
  return MGP_ERR_ARG;
}

extern "C" int mgp_operator_apply(const mgp_operator_t* op, const float* X, int C, float* Y, void* work,
                                  size_t work_bytes, void* stream) {
  return mgp_operator_apply_ex(op, X, C, Y, nullptr, nullptr, nullptr, nullptr, work, work_bytes, stream);
}

extern "C" int mgp_operator_apply_dot(const mgp_operator_t* op, const float* X, int C, float* Y,
                                      const float* dotw, float* dot_partials, void* work, size_t work_bytes,
                                      void* stream) {
  if (!dotw || !dot_partials) return MGP_ERR_ARG;
  return mgp_operator_apply_ex(op, X, C, Y, dotw, dot_partials, nullptr, nullptr, work, work_bytes, stream);
}

// Jacobi: diag(A).  For the chain B = tau I + L_sym: diag(B) = tau + diag_i; diag(B^2)_i =
// (tau + diag_i)^2 + sum_j S_ij^2 (exact); for nu > 2 the pure diagonal power is used.
__global__ void jacobi_kernel(int64_t n, const int32_t* __restrict__ rowptr, const float* __restrict__ vals,
                              const float* __restrict__ diag, const float* __restrict__ pre,
                              const float* __restrict__ post, int nu, float tau, float scale, int form,
                              float noise, float* __restrict__ minv) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    const float b = tau + diag[r];
    float q;
    if (nu == 1) {
      q = b;
    } else if (nu == 2) {
      float s2 = 0.f;
      for (int i = rowptr[r]; i < rowptr[r + 1]; ++i) s2 = fmaf(vals[i], vals[i], s2);
      q = b * b + s2;
    } else {
      q = powf(b, (float)nu);
    }
    q *= scale;
    if (pre) q *= pre[r];
    if (post) q *= post[r];
    float a;
    if (form == 0) a = q;
    else if (form == 2) a = 1.0f + noise * q;
    else a = q * (1.0f - noise * q * (1.0f - noise * q));
    minv[r] = (a > 0.f && isfinite(a)) ? 1.0f / a : 1.0f;
  }
}

extern "C" int mgp_operator_jacobi(const mgp_operator_t* op, float* minv, void* stream) {
  MGP_TRY(check_op(op));
  if (!minv) return MGP_ERR_ARG;
  const float tau = 2.0f * (float)op->nu / (op->kappa * op->kappa);
  int64_t grid = mgp_cdiv(op->L.n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(jacobi_kernel, dim3((int)grid), dim3(256), 0, mgp_stream(stream), op->L.n, op->L.rowptr,
                     op->L.vals, op->L.diag, op->pre, op->post, op->nu, tau, op->scale, op->form, op->noise,
                     minv);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}
