// (Preconditioned) conjugate gradients for the precision-side operators, multi right-hand side.
//
// Replaces linear_operator.utils.linear_cg as reached from the reference at
// precision_matern_operator.py:53, schur_complement_operator.py:28 and train_model.py:68
// (per iteration there: one operator apply = s x (2 spmm + elementwise) launches, then ~15 tiny
// reduction / elementwise launches, host-side convergence test every iteration).
//
// MI355X design
//   * single-reduction (Chronopoulos-Gear) recurrence: per iteration ONE fused vector kernel
//     (p,s,x,r,u updates + the two dot products of the new residual) and the s SpMM launches of
//     the operator chain, whose last launch carries the third dot product (u . A u);
//   * dot products are per-workgroup partials reduced again, in a fixed order, by every
//     workgroup of the consuming kernel: no atomics, no separate reduction launch, bitwise
//     reproducible, and a kernel boundary (~1.5 us) is the only synchronisation;
//   * the iteration index and the convergence flag live in device memory, so whole solves are
//     hipGraphs: cg_init + k x (apply, update) is ONE launch (rhs pointer patched into the graph, k
//     follows the previous solves), longer solves continue with `check_every`-step graphs; launches
//     behind the stopping decision return after their first loads; graphs are captured at a plan's
//     second solve; the host polls the host-mapped decision word instead of sleeping on the stream;
//   * column freezing / stopping follow linear_cg (stop_mode 0) or a per-column relative
//     residual (stop_mode 1).
#include <math.h>
#include <chrono>
#include <mutex>
#include <new>
#include <vector>
#include <string.h>
#include <atomic>
#include "mgp_common.h"
#include "mgp_internal.h"

namespace {

constexpr int kBlock = 256;
#ifndef MGP_LAB_UPD
#define MGP_LAB_UPD 0      // lab builds (tools/lab/upd_bounds.sh): 4 = never converge, alpha = beta = 0; | 1 skip the partial reads; | 2 skip the vector pass
#endif
constexpr int kMaxC = 256;
constexpr int kMaxGridVec = 512;
constexpr int kMaxPartials = 4096;   // capacity of the gamma / rr partial arrays

typedef float mgp_cg_v4f __attribute__((ext_vector_type(4)));

struct CgArgs {
  int64_t n;
  int C, TC, TS;
  float *x, *r, *u, *w, *p, *s;
  float* us;          // nullable: pre (.) u, the pre-scaled SpMM input (op->pre != NULL)
  const float* pre;   // op->pre
  const float* minv;
  float* pd_gamma;  // [2][nbv][C]
  float* pd_rr;     // [2][nbv][C]
  int nbv;
  const float* pd_delta;  // [nbs][C]
  int nbs;
  float* gamma_old;  // [2][C]
  float* alpha_old;  // [2][C]
  float* bb;         // [C]  ||b||^2
  float* resid;      // [C]  relative residual norm
  int* state;        // [0] iteration (1-based), [1] done, [2] status
  int* host_state;   // host-mapped mirror of state[0..3], written when the solve ends
  float* host_resid; // host-mapped [C]
  float tol;
  int max_iter, min_iter, stop_mode;
  int64_t rows_per_block;
  // init-free solve (C == 1, tile SpMV, no preconditioner): there is no cg_init launch -- the first apply read the
  // right-hand side itself, copied it to r and left the partials of ||b||^2 here (one slot per SpMV workgroup);
  // at iteration 1 the update takes gamma = ||r||^2 from them and treats p, s, x as zero.  NULL: classic start.
  const float* pd_bb;   // [nbs]
  // many columns (C > 16): cg_reduce_kernel has summed the partials of this step -> [3][C] gamma, ||r||^2, delta;
  // the update kernel reads 3 C floats instead of re-reducing (2 nbv + nbs) C of them in every workgroup.  NULL: off.
  float* tot;
  // quad form of the update (cg_update_q_kernel, C % 4 == 0): CQ = C / 4 column quads, TSQ row slices per workgroup (CQ * TSQ active threads)
  int CQ, TSQ;
  int* arrive;   // [9][32] arrival counters of the deciding update launch (8 groups + top, a 128-byte line each), zero between launches
};

// sh[k][sl * TC + cc] holds the partial of slice sl for column cc; result in sh[k][cc] for cc < TC.
// TC <= 64: the slices of a column sit TC lanes apart inside a wave -> xor-shuffle tree in registers,
// then one LDS hop across the four waves (1 barrier).  TC > 64: LDS tree over the remaining 1-2 slices.
// Fixed order in both cases (bitwise reproducible).  A single thread summing 256 LDS words serially
// cost ~2.5 us per call and an 8-barrier LDS tree ~1 us (profiled).
template <int K>
__device__ __forceinline__ void reduce_slices(float (*sh)[kBlock], int TC, int TS, int sl, int cc) {
  const int tid = threadIdx.x;
  if (TC <= 64) {
    float v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sh[k][tid];
    for (int o = TC; o < 64; o <<= 1) {
#pragma unroll
      for (int k = 0; k < K; ++k) v[k] += __shfl_xor(v[k], o, 64);
    }
    __syncthreads();                       // everyone has read its own slot
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < TC) {
#pragma unroll
      for (int k = 0; k < K; ++k) sh[k][wave * 64 + lane] = v[k];
    }
    __syncthreads();
    if (tid < TC) {
#pragma unroll
      for (int k = 0; k < K; ++k) sh[k][tid] = (sh[k][tid] + sh[k][64 + tid]) + (sh[k][128 + tid] + sh[k][192 + tid]);
    }
    __syncthreads();
    return;
  }
  for (int stride = TS >> 1; stride > 0; stride >>= 1) {
    __syncthreads();
    if (sl < stride) {
#pragma unroll
      for (int k = 0; k < K; ++k) sh[k][sl * TC + cc] += sh[k][(sl + stride) * TC + cc];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void cg_init_kernel(CgArgs a, const float* __restrict__ B) {
  __shared__ float sh[2][kBlock];
  const int tid = threadIdx.x, cc = tid % a.TC, sl = tid / a.TC;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)lb * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  if (blockIdx.x == 0 && tid == 0) { a.state[0] = 0; a.state[1] = 0; a.state[2] = 0; }   // the first apply ticks it to 1
  float g = 0.f, rr = 0.f;
  if (cc < a.C) {
    for (int64_t r = r0 + sl; r < r1; r += a.TS) {
      const int64_t i = r * a.C + cc;
      const float b = B[i];
      const float u = a.minv ? a.minv[r] * b : b;
      a.x[i] = 0.f; a.p[i] = 0.f; a.s[i] = 0.f;
      a.r[i] = b;
      if (a.minv) a.u[i] = u;
      if (a.us) a.us[i] = a.pre[r] * u;
      g = fmaf(b, u, g);
      rr = fmaf(b, b, rr);
    }
  }
  sh[0][tid] = g; sh[1][tid] = rr;
  reduce_slices<2>(sh, a.TC, a.TS, sl, cc);
  if (tid < a.TC && tid < a.C) {
    a.pd_gamma[(int64_t)blockIdx.x * a.C + tid] = sh[0][tid];   // parity slot 0 = "previous" of iteration 1
    a.pd_rr[(int64_t)blockIdx.x * a.C + tid] = sh[1][tid];
  }
}

// ---- Partial sums of one step, many columns.  cg_update_kernel lets every workgroup re-reduce all partials, which is
// the cheapest scheme while they are few: at C = 100 on the 60k graph it is 256 workgroups x (2 x 2 x 256 + 3750
// SpMM blocks) x 100 columns = 490 MB of L2 reads per launch, more than the vectors themselves (240 MB) -- profiled:
// 103 us per update against 89 us per SpMM.  Here each (array, 4-column group) is summed ONCE by one workgroup --
// 64 slices x 4 columns, slice sl takes blocks sl, sl + 64, ... in batches of 16 independent loads, xor tree inside
// the wave, four waves through LDS: a fixed order -- and the update reads the 3 C totals.
__global__ __launch_bounds__(kBlock) void cg_reduce_kernel(CgArgs a) {
  __shared__ float sh[kBlock / 64][4];
  const int st_it = a.state[0], st_done = a.state[1];
  if (st_done) return;
  const int prev = (st_it & 1) ^ 1;
  const int tid = threadIdx.x, cc = tid & 3, sl = tid >> 2;
  const int c = blockIdx.x * 4 + cc;
  const int cl = c < a.C ? c : a.C - 1;
  const int k = blockIdx.y;
  const int nb = (k == 2) ? a.nbs : a.nbv;
  const float* __restrict__ src = (k == 0) ? a.pd_gamma + (int64_t)prev * a.nbv * a.C
                                : (k == 1) ? a.pd_rr + (int64_t)prev * a.nbv * a.C : a.pd_delta;
  float t = 0.f;
  constexpr int U = 16;
  for (int b0 = sl; b0 < nb; b0 += U * 64) {
    float v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int b = b0 + q * 64;
      v[q] = src[(int64_t)(b < nb ? b : nb - 1) * a.C + cl];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) t += (b0 + q * 64 < nb) ? v[q] : 0.f;
  }
  for (int o = 4; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
  if ((tid & 63) < 4) sh[tid >> 6][cc] = t;
  __syncthreads();
  if (tid < 4 && c < a.C) a.tot[(int64_t)k * a.C + c] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}

__global__ __launch_bounds__(kBlock) void cg_update_kernel(CgArgs a) {
  __shared__ float sh[3][kBlock];
  __shared__ float sh_alpha[kMaxC], sh_beta[kMaxC], sh_rel[kMaxC];
  __shared__ int sh_done;
  __shared__ int sh_state[2];
  const int tid = threadIdx.x, cc = tid % a.TC, sl = tid / a.TC;
  const int C = a.C;
  // Every first touch of a line in a kernel is served from beyond L2 (~0.7 us), so the prologue is ONE
  // round trip: iteration counter + done flag, the dot partials of BOTH parities and gamma_old / alpha_old
  // of both parities are all requested before the first wait; the parity (a function of the iteration
  // counter) only selects among values already in registers.
  const int st_it = a.state[0], st_done = a.state[1];

  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  const int64_t rf = r0 + sl;
  // ---- every workgroup reduces the partials in the same fixed order: thread (sl, cc) sums
  // partials sl, sl+TS, ... of column cc with independent loads, LDS combines the TS slices
  float g2[2] = {0.f, 0.f}, rr2[2] = {0.f, 0.f}, d = 0.f;
  float go2[2] = {0.f, 0.f}, ao2[2] = {0.f, 0.f}, bb_old = 0.f;
  if (a.tot) {
    // totals of this step from cg_reduce_kernel: slice 0 carries them through the slice reduction below
    if (cc < C && sl == 0) {
      g2[0] = g2[1] = a.tot[cc];
      rr2[0] = rr2[1] = a.tot[C + cc];
      d = a.tot[2 * C + cc];
    }
  } else if (cc < C && !(MGP_LAB_UPD & 1)) {
    // batches of 8 / 32 loads on clamped indices, masked afterwards: all in flight together (one round trip per
    // batch: with 256 workgroups and TS = 16 slices the gamma / rr partials are two batches, the ~940 delta
    // partials of a 12-column SpMM two as well)
    constexpr int UG = 8, UD = 32;
    // the first two batches of each array are requested TOGETHER (128 loads per lane in flight: the whole reduction of
    // a 12-column solve on the 60k graph, 256 + 938 partial rows, in one round trip instead of four); sums in the same
    // order as before
    float gv0[2][2][UG], rv0[2][2][UG], dv0[2][UD];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b0 = sl + t * UG * a.TS;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* pg = a.pd_gamma + (int64_t)h * a.nbv * C;
        const float* pr = a.pd_rr + (int64_t)h * a.nbv * C;
#pragma unroll
        for (int q = 0; q < UG; ++q) {
          const int b = b0 + q * a.TS;
          const int bc = b < a.nbv ? b : a.nbv - 1;
          gv0[t][h][q] = pg[(int64_t)bc * C + cc];
          rv0[t][h][q] = pr[(int64_t)bc * C + cc];
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b0 = sl + t * UD * a.TS;
#pragma unroll
      for (int q = 0; q < UD; ++q) {
        const int b = b0 + q * a.TS;
        const int bc = b < a.nbs ? b : a.nbs - 1;
        dv0[t][q] = a.pd_delta[(int64_t)bc * C + cc];
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b0 = sl + t * UG * a.TS;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < UG; ++q) {
          const bool on = b0 + q * a.TS < a.nbv;
          g2[h] += on ? gv0[t][h][q] : 0.f;
          rr2[h] += on ? rv0[t][h][q] : 0.f;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int b0 = sl + t * UD * a.TS;
#pragma unroll
      for (int q = 0; q < UD; ++q) d += (b0 + q * a.TS < a.nbs) ? dv0[t][q] : 0.f;
    }
    for (int b0 = sl + 2 * UG * a.TS; b0 < a.nbv; b0 += UG * a.TS) {
      float gv[2][UG], rv[2][UG];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float* pg = a.pd_gamma + (int64_t)h * a.nbv * C;
        const float* pr = a.pd_rr + (int64_t)h * a.nbv * C;
#pragma unroll
        for (int q = 0; q < UG; ++q) {
          const int b = b0 + q * a.TS;
          const int bc = b < a.nbv ? b : a.nbv - 1;
          gv[h][q] = pg[(int64_t)bc * C + cc];
          rv[h][q] = pr[(int64_t)bc * C + cc];
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int q = 0; q < UG; ++q) {
          const bool on = b0 + q * a.TS < a.nbv;
          g2[h] += on ? gv[h][q] : 0.f;
          rr2[h] += on ? rv[h][q] : 0.f;
        }
      }
    }
    for (int b0 = sl + 2 * UD * a.TS; b0 < a.nbs; b0 += UD * a.TS) {
      float dv[UD];
#pragma unroll
      for (int q = 0; q < UD; ++q) {
        const int b = b0 + q * a.TS;
        const int bc = b < a.nbs ? b : a.nbs - 1;
        dv[q] = a.pd_delta[(int64_t)bc * C + cc];
      }
#pragma unroll
      for (int q = 0; q < UD; ++q) d += (b0 + q * a.TS < a.nbs) ? dv[q] : 0.f;
    }
  }
  if (tid < C) {
    go2[0] = a.gamma_old[tid]; go2[1] = a.gamma_old[C + tid];
    ao2[0] = a.alpha_old[tid]; ao2[1] = a.alpha_old[C + tid];
    bb_old = a.bb[tid];
  }
  // workgroup 0 may raise the done flag while this launch runs: one thread's view is published to
  // the whole workgroup so that all its waves take the same branch
  if (tid == 0) { sh_state[0] = st_it; sh_state[1] = st_done; }
  __syncthreads();
  if (sh_state[1]) return;
  const int it = sh_state[0];
  const int par = it & 1, prev = par ^ 1;
  const float g = prev ? g2[1] : g2[0], rr = prev ? rr2[1] : rr2[0];
  sh[0][tid] = g; sh[1][tid] = rr; sh[2][tid] = d;
  reduce_slices<3>(sh, a.TC, a.TS, sl, cc);
  if (tid < C) {
    const float gamma = sh[0][tid], rr2 = sh[1][tid], delta = sh[2][tid];
    const float bb = (it == 1) ? rr2 : bb_old;
    const float rel = (bb > 0.f) ? sqrtf(rr2 / bb) : 0.f;
    sh_rel[tid] = rel;
    const bool frozen = (a.stop_mode == 0) ? (rel < 1e-10f) : (rel <= a.tol);
    float alpha = 0.f, beta = 0.f;
    if (!frozen) {
      if (it == 1) {
        alpha = (delta != 0.f) ? gamma / delta : 0.f;
      } else {
        const float go = prev ? go2[1] : go2[0], ao = prev ? ao2[1] : ao2[0];
        beta = (go != 0.f) ? gamma / go : 0.f;
        const float den = delta - ((ao != 0.f) ? beta * gamma / ao : 0.f);
        alpha = (den != 0.f) ? gamma / den : 0.f;
      }
      if (!isfinite(alpha) || !isfinite(beta)) { alpha = 0.f; beta = 0.f; }
    }
    if (MGP_LAB_UPD) { alpha = 0.f; beta = 0.f; }
    sh_alpha[tid] = alpha;
    sh_beta[tid] = beta;
    if (blockIdx.x == 0) {
      a.gamma_old[par * C + tid] = gamma;
      a.alpha_old[par * C + tid] = alpha;
      if (it == 1) a.bb[tid] = bb;
      a.resid[tid] = rel;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int done = 0, status = 0;
    if (a.stop_mode == 0) {
      float m = 0.f;
      for (int c = 0; c < C; ++c) m += sh_rel[c];
      m /= (float)C;
      if (it > a.min_iter && m < a.tol) { done = 1; status = 1; }   // >= min_iter iterations done
    } else {
      int all = 1;
      for (int c = 0; c < C; ++c) all &= (sh_rel[c] <= a.tol) ? 1 : 0;
      if (all) { done = 1; status = 1; }
    }
    for (int c = 0; c < C; ++c) if (!isfinite(sh_rel[c])) { done = 1; status = 3; }
    if (MGP_LAB_UPD) { done = 0; status = 0; }
    if (!done && it > a.max_iter) { done = 1; status = 2; }
    sh_done = done;
    if (done && blockIdx.x == 0) {
      a.state[2] = status; a.state[1] = 1;
      // zero-copy results for the host: no blit kernels behind the solve
      for (int c = 0; c < C; ++c) a.host_resid[c] = sh_rel[c];
      a.host_state[0] = it; a.host_state[2] = status;
      __threadfence_system();
      a.host_state[1] = 1;
    }
  }
  __syncthreads();
  if (sh_done) return;

  // ---- fused vector update over this workgroup's contiguous rows
  float ng = 0.f, nrr = 0.f;
  if (cc < C && !(MGP_LAB_UPD & 2)) {
    const float alpha = sh_alpha[cc], beta = sh_beta[cc];
    // eight row passes per batch, all their loads (clamped rows, masked afterwards) in flight before the
    // first use: with one pass at a time a workgroup's 30 passes were 30 dependent round trips (27 us per
    // launch at 60k x 12); the per-element arithmetic and the order of the two running sums are unchanged
    constexpr int U = 8;
    for (int64_t rb = rf; rb < r1; rb += (int64_t)U * a.TS) {
      float un[U], po[U], so[U], wo[U], xo[U], ro[U], mo[U], pr[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t r = rb + (int64_t)k * a.TS;
        const int64_t rc = r < r1 ? r : rf;
        const int64_t i = rc * C + cc;
        un[k] = a.u[i]; po[k] = a.p[i]; so[k] = a.s[i]; wo[k] = a.w[i]; xo[k] = a.x[i]; ro[k] = a.r[i];
        mo[k] = a.minv ? a.minv[rc] : 1.f;
        pr[k] = a.us ? a.pre[rc] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t r = rb + (int64_t)k * a.TS;
        if (r < r1) {
          const int64_t i = r * C + cc;
          const float p = fmaf(beta, po[k], un[k]);
          const float s = fmaf(beta, so[k], wo[k]);
          a.p[i] = p;
          a.s[i] = s;
          a.x[i] = fmaf(alpha, p, xo[k]);
          const float rn = fmaf(-alpha, s, ro[k]);
          a.r[i] = rn;
          float u2 = rn;
          if (a.minv) { u2 = mo[k] * rn; a.u[i] = u2; }
          if (a.us) a.us[i] = pr[k] * u2;
          ng = fmaf(rn, u2, ng);
          nrr = fmaf(rn, rn, nrr);
        }
      }
    }
  }
  __syncthreads();
  sh[0][tid] = ng; sh[1][tid] = nrr;
  reduce_slices<2>(sh, a.TC, a.TS, sl, cc);
  if (tid < a.TC && tid < C) {
    a.pd_gamma[((int64_t)par * a.nbv + blockIdx.x) * C + tid] = sh[0][tid];
    a.pd_rr[((int64_t)par * a.nbv + blockIdx.x) * C + tid] = sh[1][tid];
  }
}

// ---- Quad form of the update for C % 4 == 0 (the column counts of training: 12 probes, 32, 100 one-hot columns).
// cg_update_kernel gives a lane one (row, column) ELEMENT: dword loads and stores, tile_cols(C) - C lanes idle (12 of 16, 100 of 128),
// a wave's stores covering 48-byte rows.  Here the [n, C] arrays are streams of float4: thread (slq, cq) owns column quad cq of the
// rows slq, slq + TSQ, ... of its workgroup's contiguous row range -- CQ * TSQ of BLOCK threads active (255 of 256 at C = 12, 250
// at C = 100), every load / store a dwordx4, a wave's accesses 1 KB contiguous.  Same arithmetic per element, same
// single-reduction recurrence, same decision code as cg_update_kernel; the per-column sums run in a fixed order (rows of a slice in
// order, slices in groups, groups left to right): bitwise reproducible, but not the order of cg_update_kernel.
// Launched behind cg_reduce_kernel only (C > 16: the step's totals are 3 C floats).  Measured on the 60k graph, C = 100, Jacobi +
// pre-scaling on (12 arrays, tools/lab/cg12.py): 57.7 -> 50.3 us per launch.  For 2 <= C <= 16, where every workgroup re-reduces
// the partials itself, the same form was slower than cg_update_kernel (12.2 us: 15.4 in 1024-thread workgroups, 17.4 in 512) and
// is not used.
template <int K, int BLOCK>
__device__ __forceinline__ void reduce_quads(mgp_cg_v4f (*shq)[BLOCK], float (*sh2)[BLOCK], float (*out)[kMaxC], const mgp_cg_v4f* v,
                                             bool act, int C, int TSQ) {
  // shq[k] as floats is P[slq][c] (flat slq * C + c): phase A every active thread stores its quad; phase B thread (g, c), g < G,
  // sums the rows g, g + G, ... of column c; phase C thread c sums the G group values
  const int tid = threadIdx.x;
  if (act) {
#pragma unroll
    for (int k = 0; k < K; ++k) shq[k][tid] = v[k];
  }
  __syncthreads();
  int G = BLOCK / C;
  if (G > 32) G = 32;
  if (G > TSQ) G = TSQ;
  const int g = tid / C, c = tid - g * C;
  if (g < G) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float* P = reinterpret_cast<const float*>(shq[k]);
      float t = 0.f;
      for (int r = g; r < TSQ; r += G) t += P[r * C + c];
      sh2[k][tid] = t;
    }
  }
  __syncthreads();
  if (tid < C) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += sh2[k][q * C + tid];
      out[k][tid] = t;
    }
  }
  __syncthreads();
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void cg_update_q_kernel(CgArgs a) {
  __shared__ mgp_cg_v4f shq[2][BLOCK];
  __shared__ float sh2[2][BLOCK];
  __shared__ float sh_tot[2][kMaxC];
  __shared__ __attribute__((aligned(16))) float sh_alpha[kMaxC];
  __shared__ __attribute__((aligned(16))) float sh_beta[kMaxC];
  __shared__ float sh_rel[kMaxC];
  __shared__ int sh_done;
  __shared__ int sh_state[2];
  const int tid = threadIdx.x;
  const int C = a.C, CQ = a.CQ, TSQ = a.TSQ;
  const bool act = tid < CQ * TSQ;
  const int slq = tid / CQ, cq = tid - slq * CQ;
  // one round trip for everything the prologue needs (first touches of a line come from beyond L2): state, the partials of BOTH
  // parities (or the step's totals), gamma_old / alpha_old of both parities -- the parity only selects among registers
  const int st_it = a.state[0], st_done = a.state[1];
  const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  const int64_t rf = r0 + slq;
  const mgp_cg_v4f z4 = {0.f, 0.f, 0.f, 0.f};
  float go2[2] = {0.f, 0.f}, ao2[2] = {0.f, 0.f}, bb_old = 0.f, tg = 0.f, trr = 0.f, td = 0.f;
  // (the step's totals come from cg_reduce_kernel: this form is only launched behind it)
  if (tid < C) { tg = a.tot[tid]; trr = a.tot[C + tid]; td = a.tot[2 * C + tid]; }
  if (tid < C) {
    go2[0] = a.gamma_old[tid]; go2[1] = a.gamma_old[C + tid];
    ao2[0] = a.alpha_old[tid]; ao2[1] = a.alpha_old[C + tid];
    bb_old = a.bb[tid];
  }
  if (tid == 0) { sh_state[0] = st_it; sh_state[1] = st_done; }
  __syncthreads();
  if (sh_state[1]) return;
  const int it = sh_state[0];
  const int par = it & 1, prev = par ^ 1;
  if (tid < C) {
    const float gamma = tg, rrn = trr, delta = td;
    const float bb = (it == 1) ? rrn : bb_old;
    const float rel = (bb > 0.f) ? sqrtf(rrn / bb) : 0.f;
    sh_rel[tid] = rel;
    const bool frozen = (a.stop_mode == 0) ? (rel < 1e-10f) : (rel <= a.tol);
    float alpha = 0.f, beta = 0.f;
    if (!frozen) {
      if (it == 1) {
        alpha = (delta != 0.f) ? gamma / delta : 0.f;
      } else {
        const float go = prev ? go2[1] : go2[0], ao = prev ? ao2[1] : ao2[0];
        beta = (go != 0.f) ? gamma / go : 0.f;
        const float den = delta - ((ao != 0.f) ? beta * gamma / ao : 0.f);
        alpha = (den != 0.f) ? gamma / den : 0.f;
      }
      if (!isfinite(alpha) || !isfinite(beta)) { alpha = 0.f; beta = 0.f; }
    }
    sh_alpha[tid] = alpha;
    sh_beta[tid] = beta;
    if (blockIdx.x == 0) {
      a.gamma_old[par * C + tid] = gamma;
      a.alpha_old[par * C + tid] = alpha;
      if (it == 1) a.bb[tid] = bb;
      a.resid[tid] = rel;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int done = 0, status = 0;
    if (a.stop_mode == 0) {
      float m = 0.f;
      for (int c = 0; c < C; ++c) m += sh_rel[c];
      m /= (float)C;
      if (it > a.min_iter && m < a.tol) { done = 1; status = 1; }   // >= min_iter iterations done
    } else {
      int all = 1;
      for (int c = 0; c < C; ++c) all &= (sh_rel[c] <= a.tol) ? 1 : 0;
      if (all) { done = 1; status = 1; }
    }
    for (int c = 0; c < C; ++c) if (!isfinite(sh_rel[c])) { done = 1; status = 3; }
    if (!done && it > a.max_iter) { done = 1; status = 2; }
    sh_done = done;
    if (done && blockIdx.x == 0) {
      a.state[2] = status; a.state[1] = 1;
      for (int c = 0; c < C; ++c) a.host_resid[c] = sh_rel[c];
      a.host_state[0] = it; a.host_state[2] = status;
      __threadfence_system();
      a.host_state[1] = 1;
    }
  }
  __syncthreads();
  if (sh_done) return;

  // ---- fused vector update: U row passes per batch, every load of a batch in flight before the first use
  mgp_cg_v4f ng = z4, nrr = z4;
  if (act) {
    const mgp_cg_v4f alpha = *reinterpret_cast<const mgp_cg_v4f*>(&sh_alpha[4 * cq]);
    const mgp_cg_v4f beta = *reinterpret_cast<const mgp_cg_v4f*>(&sh_beta[4 * cq]);
    const mgp_cg_v4f *U4 = reinterpret_cast<const mgp_cg_v4f*>(a.u), *W4 = reinterpret_cast<const mgp_cg_v4f*>(a.w);
    mgp_cg_v4f *P4 = reinterpret_cast<mgp_cg_v4f*>(a.p), *S4 = reinterpret_cast<mgp_cg_v4f*>(a.s), *X4 = reinterpret_cast<mgp_cg_v4f*>(a.x),
               *R4 = reinterpret_cast<mgp_cg_v4f*>(a.r);
    constexpr int U = 2;
    for (int64_t rb = rf; rb < r1; rb += (int64_t)U * TSQ) {
      mgp_cg_v4f un[U], po[U], so[U], wo[U], xo[U], ro[U];
      float mo[U], pr[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t r = rb + (int64_t)k * TSQ;
        const int64_t rc = r < r1 ? r : rf;
        const int64_t i = rc * CQ + cq;
        un[k] = U4[i]; po[k] = P4[i]; so[k] = S4[i]; wo[k] = W4[i]; xo[k] = X4[i]; ro[k] = R4[i];
        mo[k] = a.minv ? a.minv[rc] : 1.f;
        pr[k] = a.us ? a.pre[rc] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int64_t r = rb + (int64_t)k * TSQ;
        if (r < r1) {
          const int64_t i = r * CQ + cq;
          mgp_cg_v4f p, sv, xn, rn, u2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            p[j] = fmaf(beta[j], po[k][j], un[k][j]);
            sv[j] = fmaf(beta[j], so[k][j], wo[k][j]);
            xn[j] = fmaf(alpha[j], p[j], xo[k][j]);
            rn[j] = fmaf(-alpha[j], sv[j], ro[k][j]);
            u2[j] = a.minv ? mo[k] * rn[j] : rn[j];
            ng[j] = fmaf(rn[j], u2[j], ng[j]);
            nrr[j] = fmaf(rn[j], rn[j], nrr[j]);
          }
          P4[i] = p; S4[i] = sv; X4[i] = xn; R4[i] = rn;
          if (a.minv) reinterpret_cast<mgp_cg_v4f*>(a.u)[i] = u2;
          if (a.us) {
            mgp_cg_v4f us;
#pragma unroll
            for (int j = 0; j < 4; ++j) us[j] = pr[k] * u2[j];
            reinterpret_cast<mgp_cg_v4f*>(a.us)[i] = us;
          }
        }
      }
    }
  }
  {
    const mgp_cg_v4f v[2] = {ng, nrr};
    reduce_quads<2, BLOCK>(shq, sh2, sh_tot, v, act, C, TSQ);
  }
  if (tid < C) {
    a.pd_gamma[((int64_t)par * a.nbv + blockIdx.x) * C + tid] = sh_tot[0][tid];
    a.pd_rr[((int64_t)par * a.nbv + blockIdx.x) * C + tid] = sh_tot[1][tid];
  }
}

// ---- C == 1 specialisation of cg_update_kernel (the GP-mean solve).  Same arithmetic per element,
// but built for latency: at N = 60k the kernel is a chain of dependent waits, not bandwidth.
//   * ONE memory round trip: state, both parities of the gamma / rr partials (<= 2 per lane), the
//     delta partials (<= 16 per lane), gamma_old / alpha_old of both parities and the lane's first
//     vector element are all requested before the first wait;
//   * TWO barriers: the wave sums of the five partial totals and the state words cross the
//     workgroup through LDS once, then EVERY lane derives alpha, beta and the stopping decision
//     redundantly (no broadcast round), and the new partials take the second barrier.
// Summation order is fixed (lane slots in order, xor tree, (w0 + w1) + (w2 + w3)).
// C == 1 layout of the scalar block (plan_create_impl): fetched as one s_load_dwordx8.  Separate scalar
// loads are not batched by hipcc (each is followed by lgkmcnt(0)): five of them cost five round trips.
struct alignas(32) CgScalars {
  float go0, go1, ao0, ao1, bb, resid;
  int it, done;
};

constexpr int kC1GammaSlots = 2;    // nbv <= kMaxGridVec = 2 * 256
constexpr int kC1DeltaSlots = 16;   // nbs <= 4096

// DECIDE (the LAST update of a plan's first graph): the workgroup whose partials arrive last also takes the stopping decision
// of the NEXT step -- ||r_k||^2 summed over the partials this very launch wrote, in cg_decide_c1_kernel's order, the same rule,
// the same flags -- and leaves the end-of-graph mark.  The single-workgroup decision launch + the marker launch (4.5 + 4.0 us
// and two kernel boundaries of a ~57 us solve at N = 60k) go away.  Hand-off inside the launch: lane 0 of every workgroup
// stores its ||r||^2 partial write-through (sc1), drains its stores (s_waitcnt vmcnt(0)), makes one returning agent-scope
// atomic add on its group's arrival counter (CgArgs::arrive); the lane that completes the count joins the workgroup barrier, then all lanes of that workgroup
// read the partials with sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility, table of sc1 hand-offs: "one lane of
// each storing workgroup ... the workgroup whose add came last, told by the value its add returned").  The counter needs no reset between solves:
// in a launch either every workgroup arrives or none does (the early exits below are taken by all of them or by none), and
// the last arriver puts it back to zero.
__device__ __forceinline__ void cg_mark_end_of_graph(const CgArgs& a) {
  const int c = a.state[5] + 1;
  a.state[5] = c;
  __threadfence_system();
  a.host_state[4] = c;
}

// DS: slots of 256 SpMV-workgroup partials a lane sums (4 covers nbs <= 1024, i.e. graphs up to 65 536 nodes: each slot is two
// loads with their address arithmetic -- 24 fewer loads in front of the kernel's one wait than the general 16)
template <bool DECIDE, int DS>
__global__ __launch_bounds__(kBlock) void cg_update_c1_kernel(CgArgs a) {
  __shared__ float sh_w[kBlock / 64][6];
  __shared__ float sh_o[kBlock / 64][2];
  __shared__ int sh_state[2];
  __shared__ int sh_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const CgScalars sc = *reinterpret_cast<const CgScalars*>(a.gamma_old);
  const int st_it = sc.it, st_done = sc.done;
#ifdef MGP_STAMP
  if (blockIdx.x == 0 && tid == 0) {
    const int si = atomicAdd(a.state + 8, 1);
    reinterpret_cast<unsigned long long*>(a.state + 16)[si & 255] = wall_clock64() * 8 + 2;
  }
#endif
  // same block -> XCD -> row-range mapping as the SpMV kernels (mgp_xcd_block): the vector slices this
  // workgroup writes are the ones the SpMV workgroups of the same XCD read next, and vice versa
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)lb * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  const int64_t rf = r0 + tid;
  const int64_t rs = rf < r1 ? rf : 0;
  const float f_u = a.u[rs], f_p = a.p[rs], f_s = a.s[rs], f_w = a.w[rs], f_x = a.x[rs], f_r = a.r[rs];
  const float l_m = (a.minv ? a.minv : a.x)[rs], l_pre = (a.us ? a.pre : a.x)[rs];
  const float f_m = a.minv ? l_m : 1.f, f_pre = a.us ? l_pre : 1.f;
  float gv[2][kC1GammaSlots], rv[2][kC1GammaSlots], dv[DS];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < kC1GammaSlots; ++q) {
      const int b = tid + q * kBlock;
      const int bc = b < a.nbv ? b : a.nbv - 1;
      gv[h][q] = a.pd_gamma[(int64_t)h * a.nbv + bc];
      rv[h][q] = a.pd_rr[(int64_t)h * a.nbv + bc];
    }
  }
  float bv[DS];
  const float* __restrict__ pbb = a.pd_bb ? a.pd_bb : a.pd_delta;     // stand-in: unconditional loads
#pragma unroll
  for (int q = 0; q < DS; ++q) {
    const int b = tid + q * kBlock;
    const int bc = b < a.nbs ? b : a.nbs - 1;
    dv[q] = a.pd_delta[bc];
    bv[q] = pbb[bc];
  }
  const float go0 = sc.go0, go1 = sc.go1, ao0 = sc.ao0, ao1 = sc.ao1;
  const float bb_old = sc.bb;
  float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < kC1GammaSlots; ++q) {
    const bool on = tid + q * kBlock < a.nbv;
    t[0] += on ? gv[0][q] : 0.f; t[1] += on ? gv[1][q] : 0.f;
    t[2] += on ? rv[0][q] : 0.f; t[3] += on ? rv[1][q] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < DS; ++q) {
    const bool on = tid + q * kBlock < a.nbs;
    t[4] += on ? dv[q] : 0.f;
    t[5] += on ? bv[q] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) t[k] = mgp_wave_sum(t[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) sh_w[wave][k] = t[k];
  }
  // workgroup 0 may raise the done flag while this launch runs: one lane's view of the state is
  // published so that all waves of a workgroup take the same branch
  if (tid == 0) { sh_state[0] = st_it; sh_state[1] = st_done; }
  __syncthreads();
  const int it = sh_state[0];
  if (sh_state[1]) {
    // init-free solve, iteration 1, flag already up: the first apply reset it, so workgroup 0 raised it in THIS launch
    // (b = 0 or not finite) before this workgroup started -- the solve ends here and x, which nobody has
    // initialised, must still come out zero (every workgroup that gets to the decision itself does the same below)
    if (a.pd_bb != nullptr && it == 1) for (int64_t r = rf; r < r1; r += kBlock) a.x[r] = 0.f;
    if (DECIDE && blockIdx.x == 0 && tid == 0) cg_mark_end_of_graph(a);     // decided earlier: nobody arrives below
    return;
  }
  const int par = it & 1, prev = par ^ 1;
#pragma unroll
  for (int k = 0; k < 6; ++k) t[k] = (sh_w[0][k] + sh_w[1][k]) + (sh_w[2][k] + sh_w[3][k]);
  const bool fresh = a.pd_bb != nullptr && it == 1;       // init-free solve, first update: r = b, p = s = x = 0
  const float gamma = fresh ? t[5] : (prev ? t[1] : t[0]), rr2 = fresh ? t[5] : (prev ? t[3] : t[2]), delta = t[4];
  const float bb = (it == 1) ? rr2 : bb_old;
  const float rel = (bb > 0.f) ? sqrtf(rr2 / bb) : 0.f;
  const bool frozen = (a.stop_mode == 0) ? (rel < 1e-10f) : (rel <= a.tol);
  float alpha = 0.f, beta = 0.f;
  if (!frozen) {
    if (it == 1) {
      alpha = (delta != 0.f) ? gamma / delta : 0.f;
    } else {
      const float go = prev ? go1 : go0, ao = prev ? ao1 : ao0;
      beta = (go != 0.f) ? gamma / go : 0.f;
      const float den = delta - ((ao != 0.f) ? beta * gamma / ao : 0.f);
      alpha = (den != 0.f) ? gamma / den : 0.f;
    }
    if (!isfinite(alpha) || !isfinite(beta)) { alpha = 0.f; beta = 0.f; }
  }
  int done = 0, status = 0;
  if (a.stop_mode == 0) {
    if (it > a.min_iter && rel < a.tol) { done = 1; status = 1; }
  } else if (rel <= a.tol) { done = 1; status = 1; }
  if (!isfinite(rel)) { done = 1; status = 3; }
  if (!done && it > a.max_iter) { done = 1; status = 2; }
  if (blockIdx.x == 0 && tid == 0) {
    a.gamma_old[par] = gamma;
    a.alpha_old[par] = alpha;
    if (it == 1) a.bb[0] = bb;
    // DECIDE and not done: the last arriver of this launch writes the residual of step it + 1 to the same word, possibly
    // through another XCD's L2 -- one writer per launch, so that the device value is defined
    if (!DECIDE || done) a.resid[0] = rel;
    if (done) {
      a.state[2] = status; a.state[1] = 1;
      a.host_resid[0] = rel;
      a.host_state[0] = it; a.host_state[2] = status;
      __threadfence_system();
      a.host_state[1] = 1;
    }
  }
  if (done) {
    // an init-free solve that ends before its first update (b = 0): nobody has zeroed x
    if (fresh) for (int64_t r = rf; r < r1; r += kBlock) a.x[r] = 0.f;
    if (DECIDE && blockIdx.x == 0 && tid == 0) cg_mark_end_of_graph(a);     // every workgroup takes this exit: nobody arrives
    return;
  }

  float ng = 0.f, nrr = 0.f;
  // first element: prefetched in the prologue (the only one at N = 60k); the rest of a long row range in
  // batches of four with all their loads in flight (at N = 1M a lane walks 8 elements: one round trip each
  // made the kernel run at ~1 TB/s)
  auto step = [&](int64_t r, float un, float po, float so, float wo, float xo, float ro, float mo, float pr) {
    if (fresh) { po = 0.f; so = 0.f; xo = 0.f; }      // whatever the previous solve left there (possibly NaN)
    const float p = fmaf(beta, po, un);
    const float s = fmaf(beta, so, wo);
    a.p[r] = p;
    a.s[r] = s;
    a.x[r] = fmaf(alpha, p, xo);
    const float rn = fmaf(-alpha, s, ro);
    a.r[r] = rn;
    float u2 = rn;
    if (a.minv) { u2 = mo * rn; a.u[r] = u2; }
    if (a.us) a.us[r] = pr * u2;
    ng = fmaf(rn, u2, ng);
    nrr = fmaf(rn, rn, nrr);
  };
  if (rf < r1) step(rf, f_u, f_p, f_s, f_w, f_x, f_r, f_m, f_pre);
  constexpr int U = 4;
  for (int64_t rb = rf + kBlock; rb < r1; rb += (int64_t)U * kBlock) {
    float un[U], po[U], so[U], wo[U], xo[U], ro[U], mo[U], pr[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t r = rb + (int64_t)k * kBlock;
      const int64_t rc = r < r1 ? r : rf;
      un[k] = a.u[rc]; po[k] = a.p[rc]; so[k] = a.s[rc]; wo[k] = a.w[rc]; xo[k] = a.x[rc]; ro[k] = a.r[rc];
      mo[k] = a.minv ? a.minv[rc] : 1.f;
      pr[k] = a.us ? a.pre[rc] : 1.f;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t r = rb + (int64_t)k * kBlock;
      if (r < r1) step(r, un[k], po[k], so[k], wo[k], xo[k], ro[k], mo[k], pr[k]);
    }
  }
  ng = mgp_wave_sum(ng);
  nrr = mgp_wave_sum(nrr);
  if (lane == 0) { sh_o[wave][0] = ng; sh_o[wave][1] = nrr; }
  __syncthreads();
  if (tid == 0) {
    const float o_g = (sh_o[0][0] + sh_o[1][0]) + (sh_o[2][0] + sh_o[3][0]);
    const float o_r = (sh_o[0][1] + sh_o[1][1]) + (sh_o[2][1] + sh_o[3][1]);
    if (!DECIDE) {
      a.pd_gamma[(int64_t)par * a.nbv + lb] = o_g;
      a.pd_rr[(int64_t)par * a.nbv + lb] = o_r;
    } else {
      // write-through (sc1) stores, drained, then the arrive; the last arriver reads the partials with sc1 loads below.
      // (An agent-scope release fence here writes back every dirty L2 line of the vectors this launch has just stored:
      // measured +3.4 us per solve against the separate decision launch it was meant to save.)
      a.pd_gamma[(int64_t)par * a.nbv + lb] = o_g;                     // read by the next launch only
      __hip_atomic_store(&a.pd_rr[(int64_t)par * a.nbv + lb], o_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // two levels (one word takes ~88 arrivals per us: 235 workgroups on it were 3 us of the launch): the workgroups
      // with equal blockIdx % 8 -- one XCD under round-robin placement, which only speed depends on -- count on a line of
      // their own, the last of each group counts on the top word
      const int grp = blockIdx.x & 7, members = ((int)gridDim.x - grp + 7) >> 3, groups = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
      int last = 0;
      if (__hip_atomic_fetch_add(a.arrive + 32 * grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
        __hip_atomic_store(a.arrive + 32 * grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(a.arrive + 32 * 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1) {
          __hip_atomic_store(a.arrive + 32 * 8, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          last = 1;
        }
      }
      sh_last = last;
    }
  }
#ifdef MGP_STAMP
  if (blockIdx.x == 0 && tid == 0) {
    const int si = atomicAdd(a.state + 8, 1);
    reinterpret_cast<unsigned long long*>(a.state + 16)[si & 255] = wall_clock64() * 8 + 3;
  }
#endif
  if (!DECIDE) return;
  __syncthreads();
  if (!sh_last) return;
  // ---- the last arriver: stopping decision of step it + 1 (cg_decide_c1_kernel, same sums in the same order)
  {
    const int itn = it + 1;
    float t2 = 0.f;
#pragma unroll
    for (int q = 0; q < kC1GammaSlots; ++q) {
      const int b = tid + q * kBlock;
      const float v = __hip_atomic_load(&a.pd_rr[(int64_t)par * a.nbv + (b < a.nbv ? b : a.nbv - 1)], __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT);
      t2 += (b < a.nbv) ? v : 0.f;
    }
    t2 = mgp_wave_sum(t2);
    __syncthreads();                          // sh_o is reused below: everyone has read sh_last / the first use is over
    if (lane == 0) sh_o[wave][0] = t2;
    __syncthreads();
    if (tid != 0) return;
    const float rr2n = (sh_o[0][0] + sh_o[1][0]) + (sh_o[2][0] + sh_o[3][0]);
    const float reln = (bb > 0.f) ? sqrtf(rr2n / bb) : 0.f;
    int dn = 0, stn = 0;
    if (a.stop_mode == 0) {
      if (itn > a.min_iter && reln < a.tol) { dn = 1; stn = 1; }
    } else if (reln <= a.tol) { dn = 1; stn = 1; }
    if (!isfinite(reln)) { dn = 1; stn = 3; }
    if (!dn && itn > a.max_iter) { dn = 1; stn = 2; }
    // The host reads nothing but host-mapped words (the solution stays in stream order).  The decision travels as ONE
    // naturally aligned 8-byte record {residual bits, step << 8 | status << 4 | 3} at host_state[8..9], written by one store
    // instruction (one PCIe write, observed whole by the host's 8-byte read): no drain between "the details" and "the
    // flag" -- that wait for a host-memory write to be acknowledged was ~1 us of every solve -- and no
    // __threadfence_system(), which would also write back every dirty L2 line of the vectors this launch has just stored.
    // run_cg clears the record before every solve and unpacks it into the words the other deciding kernels write.
    const int c = a.state[5] + 1;
    a.state[5] = c;
    a.resid[0] = reln;                        // the only writer of this word in a deciding launch that goes on (see above)
    if (dn) {
      a.state[2] = stn; a.state[1] = 1;
      const unsigned long long rec = (unsigned long long)__builtin_bit_cast(unsigned, reln) |
                                     ((unsigned long long)(unsigned)((itn << 8) | (stn << 4) | 3) << 32);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.host_state + 8), rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __hip_atomic_store(a.host_state + 4, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // end-of-graph mark
  }
}

// ---- Stopping decision alone (C == 1).  The update kernel of step k+1 is where ||r_k|| <= tol is noticed, after
// apply k+1 has already run for nothing: at 3 iterations per solve that is 2 of 8 SpMVs + a launch, ~14 us of
// ~76.  The first graph of a plan is captured for the step count the previous solves needed, so its last
// (apply, update) pair -- the one that only detects -- is replaced by this single-workgroup launch: the same
// partial sums in the same order, the same rule, the same flags as cg_update_c1_kernel would write at
// it + 1.  Not converged: it writes nothing and the continuation graph carries on as before.
__global__ __launch_bounds__(kBlock) void cg_decide_c1_kernel(CgArgs a) {
  __shared__ float sh_w[kBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const CgScalars sc = *reinterpret_cast<const CgScalars*>(a.gamma_old);
#ifdef MGP_STAMP
  if (tid == 0) {
    const int si = atomicAdd(a.state + 8, 1);
    reinterpret_cast<unsigned long long*>(a.state + 16)[si & 255] = wall_clock64() * 8 + 4;
  }
#endif
  if (sc.done) return;
  const int it = sc.it + 1;                 // the step whose update would take this decision
  const int prev = (it & 1) ^ 1;            // slot the last update wrote
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < kC1GammaSlots; ++q) {
    const int b = tid + q * kBlock;
    const float v = a.pd_rr[(int64_t)prev * a.nbv + (b < a.nbv ? b : a.nbv - 1)];
    t += (b < a.nbv) ? v : 0.f;
  }
  t = mgp_wave_sum(t);
  if (lane == 0) sh_w[wave] = t;
  __syncthreads();
  if (tid != 0) return;
  const float rr2 = (sh_w[0] + sh_w[1]) + (sh_w[2] + sh_w[3]);
  const float bb = sc.bb;                   // it >= 2 here: written by the first update
  const float rel = (bb > 0.f) ? sqrtf(rr2 / bb) : 0.f;
  int done = 0, status = 0;
  if (a.stop_mode == 0) {
    if (it > a.min_iter && rel < a.tol) { done = 1; status = 1; }
  } else if (rel <= a.tol) { done = 1; status = 1; }
  if (!isfinite(rel)) { done = 1; status = 3; }
  if (!done && it > a.max_iter) { done = 1; status = 2; }
  if (!done) return;
  a.resid[0] = rel;
  a.state[2] = status; a.state[1] = 1;
  a.host_resid[0] = rel;
  a.host_state[0] = it; a.host_state[2] = status;
  a.host_state[3] = 1;                      // decided without running apply `it`
  __threadfence_system();
  a.host_state[1] = 1;
}

// ---- End-of-graph marker.  The host reads the stopping flag alone while the first graph of a solve runs (a stream query
// during a ~55 us solve costs ~5 us of it); a first graph that ends UNdecided (the solve needs more steps than the graph
// holds) must still be noticed at once: its last node counts the graphs that have run to their end in a host-mapped word
// the host reads next to the flag.  When the decision has been taken the host has long seen the flag: this launch is
// behind the critical path.
__global__ void cg_marker_kernel(int* state, int* host_state) {
  const int c = state[5] + 1;
  state[5] = c;
  __threadfence_system();
  host_state[4] = c;
}

// ---- Complex-shift solve of the (K + s I) system in precision form, symmetric normalisation, nu = 2 (round 5).
// A = I + c B^2 with B = tau I + L_sym and c = noise * scale factorises over the complex numbers: 1 + c b^2 =
// (1 + i sigma b)(1 - i sigma b), sigma = sqrt(c), and 1 / (1 + c b^2) = Re[1 / (1 + i sigma b)], so
//     x = Re[(I + i sigma B)^-1 y].
// M = I + i sigma B is complex SYMMETRIC (not Hermitian) with its spectrum on the segment {1 + i sigma b}: its condition is
// ~sqrt(cond(A)), and COCG -- the CG recurrences with the unconjugated bilinear form z . w = sum z_j w_j -- needs about the
// square root of CG's iterations on A.  Measured on the 1M-node swiss roll (cond(A) = 1.4e4): 58 iterations of ONE product with
// B against 688 iterations of two for CG on A, the same solution to 2e-7 (tools/lab/cocg_s5.py).  The reference's call site is the
// unpreconditioned linear_cg of precision_matern_operator.py:53; this is the north star's "preconditioned CG" taken to its
// end for the systems that factorise: an exact algebraic split instead of an approximate inverse.
//   * vectors z, r, p, s are [n] complex (float2); the product B u runs on the 4-column tile SpMM (spmm_tile_q_kernel) over
//     u4 = (u_re, u_im, u_re, u_im) with dot weights w4 = (u_re, u_im, u_im, u_re): its per-workgroup partials are exactly the
//     four real sums of u . B u = (d0 - d1) + i (d2 + d3); w = M u = u + i sigma B u is formed in the update;
//   * single-reduction (Chronopoulos-Gear) form as cg_update_kernel: gamma = r . r, delta = u . M u = gamma + i sigma u . B u
//     (u = r: no preconditioner), beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old), all complex;
//   * stop: ||r||_2 <= tol ||b||_2 on the COMPLEX residual (an upper bound for the residual of the real system's solution
//     Re z up to the factor |I - i sigma B|; callers that need a certified true residual use the refinement rounds, which
//     evaluate b - A x in fp64 on the original operator);
//   * state words, host flags, skip / tick and the chunked hipGraph replay are those of the real solver.
constexpr int kCxDeltaSlots = 16;    // nbs4 <= 4096

struct CxArgs {
  float2 *z, *r, *p, *s;     // [n] complex
  mgp_cg_v4f *u4, *w4;       // SpMM input (u_re, u_im, u_re, u_im), dot weights (u_re, u_im, u_im, u_re)
  const mgp_cg_v4f* y4;      // B u4
  const float* pd4;          // [nbs4][4] partials of w4 . y4 per column
  int nbs4;
  float sigma;
  float* pd_g;               // [2][nbv][4]: gamma_re, gamma_im, ||r||^2, 0
  float* sc;                 // [2][4] {gamma_old re, im, alpha_old re, im} per parity, then [8] = ||b||^2
};

__device__ __forceinline__ float2 cx_mul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cx_div(float2 a, float2 b) {
  const float d = b.x * b.x + b.y * b.y;
  if (!(d > 0.f)) return make_float2(0.f, 0.f);
  return make_float2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

__global__ __launch_bounds__(kBlock) void cx_init_kernel(CgArgs a, CxArgs c, const float* __restrict__ B) {
  __shared__ float sh_o[kBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)lb * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  if (blockIdx.x == 0 && tid == 0) { a.state[0] = 0; a.state[1] = 0; a.state[2] = 0; }   // the first product ticks it to 1
  float g = 0.f;
  for (int64_t r = r0 + tid; r < r1; r += kBlock) {
    const float b = B[r];
    c.z[r] = make_float2(0.f, 0.f);
    c.r[r] = make_float2(b, 0.f);
    c.p[r] = make_float2(0.f, 0.f);
    c.s[r] = make_float2(0.f, 0.f);
    c.u4[r] = mgp_cg_v4f{b, 0.f, b, 0.f};
    c.w4[r] = mgp_cg_v4f{b, 0.f, 0.f, b};
    a.x[r] = 0.f;
    g = fmaf(b, b, g);
  }
  g = mgp_wave_sum(g);
  if (lane == 0) sh_o[wave] = g;
  __syncthreads();
  if (tid == 0) {
    const float t = (sh_o[0] + sh_o[1]) + (sh_o[2] + sh_o[3]);
    float* dst = c.pd_g + 4 * (int64_t)lb;            // parity slot 0 = "previous" of iteration 1
    dst[0] = t; dst[1] = 0.f; dst[2] = t; dst[3] = 0.f;
  }
}

__global__ __launch_bounds__(kBlock) void cx_update_kernel(CgArgs a, CxArgs c) {
  __shared__ float sh_w[kBlock / 64][7];
  __shared__ float sh_o[kBlock / 64][3];
  __shared__ int sh_state[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int st_it = a.state[0], st_done = a.state[1];
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)lb * a.rows_per_block;
  int64_t r1 = r0 + a.rows_per_block;
  if (r1 > a.n) r1 = a.n;
  // ---- one round trip: the partials of both parities, the four-column partials of u . B u, the scalars
  float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // gamma re/im, rr (parity 0), gamma re/im, rr (parity 1) -> selected below
  mgp_cg_v4f gv[2][kC1GammaSlots];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int q = 0; q < kC1GammaSlots; ++q) {
      const int b = tid + q * kBlock;
      const int bc = b < a.nbv ? b : a.nbv - 1;
      gv[h][q] = *reinterpret_cast<const mgp_cg_v4f*>(c.pd_g + 4 * ((int64_t)h * a.nbv + bc));
    }
  }
  mgp_cg_v4f dv[kCxDeltaSlots];
#pragma unroll
  for (int q = 0; q < kCxDeltaSlots; ++q) {
    const int b = tid + q * kBlock;
    const int bc = b < c.nbs4 ? b : c.nbs4 - 1;
    dv[q] = *reinterpret_cast<const mgp_cg_v4f*>(c.pd4 + 4 * (int64_t)bc);
  }
  const float go_r0 = c.sc[0], go_i0 = c.sc[1], ao_r0 = c.sc[2], ao_i0 = c.sc[3];
  const float go_r1 = c.sc[4], go_i1 = c.sc[5], ao_r1 = c.sc[6], ao_i1 = c.sc[7];
  const float bb_old = c.sc[8];
  float g0[3] = {0.f, 0.f, 0.f}, g1[3] = {0.f, 0.f, 0.f}, d4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < kC1GammaSlots; ++q) {
    const bool on = tid + q * kBlock < a.nbv;
    g0[0] += on ? gv[0][q].x : 0.f; g0[1] += on ? gv[0][q].y : 0.f; g0[2] += on ? gv[0][q].z : 0.f;
    g1[0] += on ? gv[1][q].x : 0.f; g1[1] += on ? gv[1][q].y : 0.f; g1[2] += on ? gv[1][q].z : 0.f;
  }
#pragma unroll
  for (int q = 0; q < kCxDeltaSlots; ++q) {
    const bool on = tid + q * kBlock < c.nbs4;
    d4[0] += on ? dv[q].x : 0.f; d4[1] += on ? dv[q].y : 0.f; d4[2] += on ? dv[q].z : 0.f; d4[3] += on ? dv[q].w : 0.f;
  }
  if (tid == 0) { sh_state[0] = st_it; sh_state[1] = st_done; }
  __syncthreads();
  if (sh_state[1]) return;
  const int it = sh_state[0];
  const int par = it & 1, prev = par ^ 1;
  t[0] = prev ? g1[0] : g0[0]; t[1] = prev ? g1[1] : g0[1]; t[2] = prev ? g1[2] : g0[2];
  t[3] = d4[0]; t[4] = d4[1]; t[5] = d4[2]; t[6] = d4[3];
#pragma unroll
  for (int k = 0; k < 7; ++k) t[k] = mgp_wave_sum(t[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) sh_w[wave][k] = t[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 7; ++k) t[k] = (sh_w[0][k] + sh_w[1][k]) + (sh_w[2][k] + sh_w[3][k]);
  const float2 gamma = make_float2(t[0], t[1]);
  const float rr2 = t[2];
  const float2 uBu = make_float2(t[3] - t[4], t[5] + t[6]);
  const float2 delta = make_float2(gamma.x - c.sigma * uBu.y, gamma.y + c.sigma * uBu.x);     // u . (u + i sigma B u)
  const float bb = (it == 1) ? rr2 : bb_old;
  const float rel = (bb > 0.f) ? sqrtf(rr2 / bb) : 0.f;
  float2 alpha = make_float2(0.f, 0.f), beta = make_float2(0.f, 0.f);
  if (rel > a.tol) {
    if (it == 1) {
      alpha = cx_div(gamma, delta);
    } else {
      const float2 go = prev ? make_float2(go_r1, go_i1) : make_float2(go_r0, go_i0);
      const float2 ao = prev ? make_float2(ao_r1, ao_i1) : make_float2(ao_r0, ao_i0);
      beta = cx_div(gamma, go);
      const float2 corr = cx_div(cx_mul(beta, gamma), ao);
      alpha = cx_div(gamma, make_float2(delta.x - corr.x, delta.y - corr.y));
    }
    if (!isfinite(alpha.x) || !isfinite(alpha.y) || !isfinite(beta.x) || !isfinite(beta.y)) {
      alpha = make_float2(0.f, 0.f); beta = make_float2(0.f, 0.f);
    }
  }
  int done = 0, status = 0;
  if (rel <= a.tol) { done = 1; status = 1; }
  if (!isfinite(rel)) { done = 1; status = 3; }
  if (!done && it > a.max_iter) { done = 1; status = 2; }
  if (blockIdx.x == 0 && tid == 0) {
    c.sc[4 * par + 0] = gamma.x; c.sc[4 * par + 1] = gamma.y;
    c.sc[4 * par + 2] = alpha.x; c.sc[4 * par + 3] = alpha.y;
    if (it == 1) c.sc[8] = bb;
    a.resid[0] = rel;
    if (done) {
      a.state[2] = status; a.state[1] = 1;
      a.host_resid[0] = rel;
      a.host_state[0] = it; a.host_state[2] = status;
      __threadfence_system();
      a.host_state[1] = 1;
    }
  }
  if (done) return;
  // ---- vector update over this workgroup's rows, four rows per lane in flight
  float ngr = 0.f, ngi = 0.f, nrr = 0.f;
  constexpr int U = 4;
  const int64_t rf = r0 + tid;
  for (int64_t rb = rf; rb < r1; rb += (int64_t)U * kBlock) {
    float2 uo[U], po[U], so[U], zo[U];
    mgp_cg_v4f yo[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t r = rb + (int64_t)k * kBlock;
      const int64_t rc = r < r1 ? r : rf;
      uo[k] = c.r[rc]; po[k] = c.p[rc]; so[k] = c.s[rc]; zo[k] = c.z[rc]; yo[k] = c.y4[rc];
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t r = rb + (int64_t)k * kBlock;
      if (r < r1) {
        const float2 u = uo[k];
        const float2 w = make_float2(u.x - c.sigma * yo[k].y, u.y + c.sigma * yo[k].x);       // M u = u + i sigma B u
        const float2 bp = cx_mul(beta, po[k]), bs = cx_mul(beta, so[k]);
        const float2 pn = make_float2(u.x + bp.x, u.y + bp.y);
        const float2 sn = make_float2(w.x + bs.x, w.y + bs.y);
        const float2 ap = cx_mul(alpha, pn), as = cx_mul(alpha, sn);
        const float2 zn = make_float2(zo[k].x + ap.x, zo[k].y + ap.y);
        const float2 rn = make_float2(u.x - as.x, u.y - as.y);
        c.p[r] = pn; c.s[r] = sn; c.z[r] = zn; c.r[r] = rn;
        a.x[r] = zn.x;                                             // the real system's solution: Re z
        c.u4[r] = mgp_cg_v4f{rn.x, rn.y, rn.x, rn.y};
        c.w4[r] = mgp_cg_v4f{rn.x, rn.y, rn.y, rn.x};
        ngr += rn.x * rn.x - rn.y * rn.y;
        ngi += 2.f * rn.x * rn.y;
        nrr += rn.x * rn.x + rn.y * rn.y;
      }
    }
  }
  ngr = mgp_wave_sum(ngr); ngi = mgp_wave_sum(ngi); nrr = mgp_wave_sum(nrr);
  if (lane == 0) { sh_o[wave][0] = ngr; sh_o[wave][1] = ngi; sh_o[wave][2] = nrr; }
  __syncthreads();
  if (tid == 0) {
    float* dst = c.pd_g + 4 * ((int64_t)par * a.nbv + lb);
    dst[0] = (sh_o[0][0] + sh_o[1][0]) + (sh_o[2][0] + sh_o[3][0]);
    dst[1] = (sh_o[0][1] + sh_o[1][1]) + (sh_o[2][1] + sh_o[3][1]);
    dst[2] = (sh_o[0][2] + sh_o[1][2]) + (sh_o[2][2] + sh_o[3][2]);
    dst[3] = 0.f;
  }
}

// ---- iterative refinement (stop_mode 1, max_refine > 0): the recurrence residual of a single-
// reduction CG drifts from the true residual on ill-conditioned systems in fp32; the true residual
// R = B - A x is formed explicitly and, if it misses the tolerance, A d = R is solved and x += d.
__global__ __launch_bounds__(kBlock) void refine_residual_kernel(const float* __restrict__ B, const float* __restrict__ T,
                                                                 float* __restrict__ R, int64_t n, int C,
                                                                 float* __restrict__ partial /*[grid][C][2]*/) {
  __shared__ float sh[2][kBlock];
  int TC = 1;
  while (TC < C) TC <<= 1;
  const int TS = kBlock / TC;
  const int tid = threadIdx.x, cc = tid % TC, sl = tid / TC;
  float rr = 0.f, bb = 0.f;
  if (cc < C) {
    for (int64_t r = (int64_t)blockIdx.x * TS + sl; r < n; r += (int64_t)gridDim.x * TS) {
      const int64_t i = r * C + cc;
      const float b = B[i], d = b - T[i];
      R[i] = d;
      rr = fmaf(d, d, rr);
      bb = fmaf(b, b, bb);
    }
  }
  sh[0][tid] = rr; sh[1][tid] = bb;
  reduce_slices<2>(sh, TC, TS, sl, cc);
  if (tid < TC && tid < C) {
    partial[((int64_t)blockIdx.x * C + tid) * 2 + 0] = sh[0][tid];
    partial[((int64_t)blockIdx.x * C + tid) * 2 + 1] = sh[1][tid];
  }
}

__global__ void refine_finalize_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ host_rel) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float rr = 0.f, bb = 0.f;
    for (int b = 0; b < nblk; ++b) { rr += partial[((int64_t)b * C + c) * 2]; bb += partial[((int64_t)b * C + c) * 2 + 1]; }
    host_rel[c] = bb > 0.f ? sqrtf(rr / bb) : 0.f;
  }
}

__global__ void refine_accumulate_kernel(float* __restrict__ xacc, const float* __restrict__ x, int64_t total, int first) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    xacc[i] = first ? x[i] : xacc[i] + x[i];
}

// ---- fp64 refinement (single GPU): the accumulated solution and the true residual live in fp64
__global__ void refine_accumulate64_kernel(double* __restrict__ xacc, const float* __restrict__ x, int64_t total, int first) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    xacc[i] = first ? (double)x[i] : xacc[i] + (double)x[i];
}

// R = B - T (fp64) -> next right-hand side in fp32; partial[blk][c] = {sum R^2, sum B^2} in fp64
__global__ __launch_bounds__(kBlock) void refine_residual64_kernel(const float* __restrict__ B, const double* __restrict__ T,
                                                                   float* __restrict__ R, int64_t n, int C,
                                                                   double* __restrict__ partial) {
  __shared__ double sh[2][kBlock];
  int TC = 1;
  while (TC < C) TC <<= 1;
  const int TS = kBlock / TC;
  const int tid = threadIdx.x, cc = tid % TC, sl = tid / TC;
  double rr = 0.0, bb = 0.0;
  if (cc < C) {
    for (int64_t r = (int64_t)blockIdx.x * TS + sl; r < n; r += (int64_t)gridDim.x * TS) {
      const int64_t i = r * C + cc;
      const double b = (double)B[i], d = b - T[i];
      R[i] = (float)d;
      rr += d * d;
      bb += b * b;
    }
  }
  sh[0][tid] = rr; sh[1][tid] = bb;
  __syncthreads();
  if (tid < TC && tid < C) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < TS; ++k) { s0 += sh[0][k * TC + tid]; s1 += sh[1][k * TC + tid]; }
    partial[((int64_t)blockIdx.x * C + tid) * 2 + 0] = s0;
    partial[((int64_t)blockIdx.x * C + tid) * 2 + 1] = s1;
  }
}

__global__ void refine_finalize64_kernel(const double* __restrict__ partial, int nblk, int C, float* __restrict__ host_rel) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double rr = 0.0, bb = 0.0;
    for (int b = 0; b < nblk; ++b) { rr += partial[((int64_t)b * C + c) * 2]; bb += partial[((int64_t)b * C + c) * 2 + 1]; }
    host_rel[c] = bb > 0.0 ? (float)sqrt(rr / bb) : 0.f;
  }
}

__global__ void refine_publish64_kernel(const double* __restrict__ xacc, float* __restrict__ x, float* __restrict__ X2,
                                        int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)xacc[i];
    x[i] = v;
    if (X2) X2[i] = v;
  }
}

// Host-mapped blocks of the plans (solve state, residuals), pooled: hipHostMalloc costs ~0.1 ms and hipHostFree waits for the
// device to go idle, and a training epoch creates and retires a dozen plans (every hyper-parameter step changes the operator).
// A retired plan's block goes back behind an event recorded on the plan's stream (its last graph's end-of-graph marker may
// still be on its way); it is handed out again only once that event has completed.
struct HostBlock {
  void* p = nullptr;
  size_t bytes = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
};
std::mutex g_host_pool_mu;
std::vector<HostBlock> g_host_pool;

bool host_block_acquire(size_t bytes, HostBlock* out) {
  {
    std::lock_guard<std::mutex> lk(g_host_pool_mu);
    for (size_t i = 0; i < g_host_pool.size(); ++i) {
      HostBlock& b = g_host_pool[i];
      if (b.bytes < bytes || b.bytes > 4 * bytes + 4096) continue;
      if (b.pending && hipEventQuery(b.ev) != hipSuccess) continue;
      *out = b;
      out->pending = false;
      g_host_pool.erase(g_host_pool.begin() + (long)i);
      return true;
    }
  }
  HostBlock b;
  b.bytes = (bytes + 4095) & ~size_t(4095);
  if (hipHostMalloc(&b.p, b.bytes, hipHostMallocMapped) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&b.ev, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(b.p); return false; }
  *out = b;
  return true;
}

void host_block_release(HostBlock b, hipStream_t stream) {
  if (!b.p) return;
  if (hipEventRecord(b.ev, stream) != hipSuccess) {          // cannot tell when its last writer ends: do not reuse it
    (void)hipEventDestroy(b.ev);
    (void)hipHostFree(b.p);
    return;
  }
  b.pending = true;
  std::lock_guard<std::mutex> lk(g_host_pool_mu);
  if (g_host_pool.size() < 64) { g_host_pool.push_back(b); return; }
  (void)hipEventDestroy(b.ev);
  (void)hipHostFree(b.p);
}

struct CgPlan {
  bool poisoned = false;     // a solve ended in MGP_ERR_TIMEOUT: destroy leaks instead of waiting (mgp_cg_plan_solve)
  HostBlock host_block;
  mgp_operator_t op;
  MgpDist dist;             // row partition (is_dist): op.L holds the local rows only
  bool is_dist;
  int nb_loc;               // SpMM workgroups per rank that write dot partials
  int C;
  mgp_cg_params_t prm;
  CgArgs args;
  void* op_work;
  size_t op_work_bytes;
  float* pd_delta;
  hipStream_t stream;       // caller's stream: all work is enqueued here
  hipStream_t cap_stream;   // private stream used only to capture the iteration graph
  hipGraphExec_t exec;        // `chunk` x (apply, update): continuation replays
  hipGraph_t graph_first;     // cg_init + `len_first` x (apply, update): a short solve is ONE graph launch
  hipGraphExec_t exec_first;
  hipGraphNode_t init_node;   // the cg_init node of graph_first (its rhs argument is patched per solve)
  bool has_graph, has_first;
  int chunk, len_first;
  int last_need;              // (apply, update) pairs the previous solve needed: len_first follows it
  const float* patched_rhs;   // rhs the cg_init node currently points at
  int solves;                 // run_cg calls so far (graphs are captured at the second one)
  int64_t last_solve_ns;      // host time of the previous solve when it ended inside its first chunk (0: it did not)
  int marker_seq;             // first graphs launched so far = what cg_marker_kernel will have counted when the newest ends
  bool graphs_tried;
  bool init_free;             // no cg_init launch: the first apply reads the rhs itself (CgArgs::pd_bb)
  bool rebound;               // mgp_cg_plan_rebind since the last solve: the graphs are refreshed before they are launched again
  int upd_quads;              // 0: cg_update_kernel; else the workgroup size of cg_update_q_kernel (C % 4 == 0)
  bool decide_in_update;      // the first graph's last update decides + marks (mgp_cg_set_decide_in_update at plan creation)
  bool cx;                    // complex-shift solve (cx_update_kernel): form 2, nu = 2, symmetric normalisation, C = 1
  CxArgs cxa;
  mgp_operator_t opB;         // B = tau I + L_sym (one launch of the 4-column SpMM per iteration)
  float* pd4;                 // [nb4][4] partials of u . B u
  int nb4;
  void* op_work4;
  size_t op_work4_bytes;
  float* pd_bb;               // [nbs] partials of ||b||^2 written by the first apply
  char first_record[MGP_SPMM_RECORD_BYTES];   // launch arguments of the first graph's root SpMV (rhs patched per solve)
  int32_t* host_state;      // pinned
  float* host_resid;        // pinned
  float *xacc, *rbuf, *tbuf, *rpart;   // refinement: accumulated solution, residual rhs, A x, partials
  double *xacc64, *t64, *work64, *rpart64;   // single GPU: the same in fp64 (true residual from an fp64 apply)
  float* host_true_rel;     // host-mapped [C]: true relative residuals
  float* dev_true_rel;
};

int tile_cols(int C) {
  int t = 1;
  while (t < C) t <<= 1;
  return t;
}

size_t cg_bytes(const mgp_operator_t* op, int C, int world = 1) {
  const size_t nc = mgp_align((size_t)op->L.n * world * C * sizeof(float));
  const int nbs = mgp_spmm_dot_blocks_for(&op->L, C) * world;
  size_t b = 10 * nc;                                  // x r u w p s us + refinement xacc rbuf tbuf
  b += 4 * nc + 256;                                   // operator chain scratch (global length)
  b += 4 * mgp_align((size_t)kMaxPartials * C * sizeof(float));  // pd_gamma[2], pd_rr[2]
  b += 2 * mgp_align((size_t)nbs * C * sizeof(float));          // pd_delta, pd_bb
  b += mgp_align((6 * (size_t)C + 16 + 1024 + 8192) * sizeof(float));   // gamma_old[2] alpha_old[2] bb resid state (+ lab stamps)
  b += mgp_align(3 * (size_t)C * sizeof(float));                 // tot (cg_reduce_kernel)
  b += mgp_align(9 * 32 * sizeof(int));                         // arrival counters (cg_update_c1_kernel<true>)
  b += mgp_align((size_t)256 * C * 2 * sizeof(float));          // refinement partials
  b += 6 * 2 * nc + mgp_align((size_t)256 * C * 2 * sizeof(double));   // fp64 refinement: xacc, A x, 4 chain buffers
  if (C == 1 && world == 1) {
    // complex-shift solve: z r p s (float2), u4 w4 y4 (float4), the 4-column chain scratch, partials, scalars
    const size_t n = (size_t)op->L.n;
    b += 4 * mgp_align(n * 8) + 3 * mgp_align(n * 16) + 4 * mgp_align(n * 16) + 256;
    b += mgp_align((size_t)kCxDeltaSlots * kBlock * 4 * sizeof(float)) + mgp_align((size_t)2 * kMaxGridVec * 4 * sizeof(float)) + 256;
  }
  return b + 1024;
}

constexpr int kReduceOnceAbove = 16;
std::atomic<int> g_cg_complex_shift{1};   // form 2, nu = 2, symmetric normalisation, C = 1: the complex-shift solve (mgp_cg_set_complex_shift(0): CG on A)
std::atomic<int> g_cg_reduce_once{1};   // mgp_cg_set_reduce_once(0): every update workgroup re-reduces the partials at any C (A/B, tests)
std::atomic<int> g_cg_update_quads{1};   // C % 4 == 0 plans update through cg_update_q_kernel (mgp_cg_set_update_quads(0): the element form at every C)
std::atomic<int> g_cg_poll_spin{64};    // flag reads between two looks at the clock in the flag-only poll window; 0: no such window (mgp_cg_set_poll_spin)
std::atomic<int> g_cg_init_free{1};   // C == 1 plans start without a cg_init launch (mgp_cg_set_init_free(0): classic start)
std::atomic<int> g_cg_decide_in_update{1};   // the first graph's last update decides + marks (mgp_cg_set_decide_in_update(0): separate launches)

// one CG step = operator apply (w = A u, partials of u . w, ticks the iteration counter; skipped once
// converged) followed by the fused update kernel, which also takes the stopping decision: every
// graph therefore ends right behind a decision and a solve that needs k steps runs exactly k bodies.
void launch_update_c1(CgPlan* pl, hipStream_t st, bool decide_last) {
  const dim3 grid(pl->args.nbv), block(kBlock);
  if (pl->args.nbs <= 4 * kBlock) {
    if (decide_last) hipLaunchKernelGGL((cg_update_c1_kernel<true, 4>), grid, block, 0, st, pl->args);
    else hipLaunchKernelGGL((cg_update_c1_kernel<false, 4>), grid, block, 0, st, pl->args);
  } else {
    if (decide_last) hipLaunchKernelGGL((cg_update_c1_kernel<true, kC1DeltaSlots>), grid, block, 0, st, pl->args);
    else hipLaunchKernelGGL((cg_update_c1_kernel<false, kC1DeltaSlots>), grid, block, 0, st, pl->args);
  }
}

int enqueue_body(CgPlan* pl, hipStream_t st, bool decide_last = false) {
  if (pl->cx) {
    // B u on the 4-column tile SpMM (partials of u . B u ride along; skipped once decided; ticks the iteration), then the
    // complex update
    MGP_TRY(mgp_operator_apply_dist(&pl->opB, nullptr, reinterpret_cast<const float*>(pl->cxa.u4), nullptr, 4,
                                    const_cast<float*>(reinterpret_cast<const float*>(pl->cxa.y4)),
                                    reinterpret_cast<const float*>(pl->cxa.w4), pl->pd4, pl->nb4, pl->args.state + 1,
                                    pl->args.state, pl->op_work4, pl->op_work4_bytes, st));
    hipLaunchKernelGGL(cx_update_kernel, dim3(pl->args.nbv), dim3(kBlock), 0, st, pl->args, pl->cxa);
    MGP_LAUNCH_CHECK();
    return MGP_OK;
  }
  MGP_TRY(mgp_operator_apply_dist(&pl->op, pl->is_dist ? &pl->dist : nullptr, pl->args.u, pl->args.us, pl->C,
                                  pl->args.w, pl->args.u, pl->pd_delta, pl->nb_loc, pl->args.state + 1,
                                  pl->args.state, pl->op_work, pl->op_work_bytes, st));
  if (pl->C == 1 && pl->args.nbv <= kC1GammaSlots * kBlock && pl->args.nbs <= kC1DeltaSlots * kBlock) {
    launch_update_c1(pl, st, decide_last);
  } else {
    if (pl->args.tot) {
      hipLaunchKernelGGL(cg_reduce_kernel, dim3((unsigned)mgp_cdiv(pl->C, 4), 3), dim3(kBlock), 0, st, pl->args);
      MGP_LAUNCH_CHECK();
    }
    if (pl->upd_quads) hipLaunchKernelGGL(cg_update_q_kernel<kBlock>, dim3(pl->args.nbv), dim3(kBlock), 0, st, pl->args);
    else hipLaunchKernelGGL(cg_update_kernel, dim3(pl->args.nbv), dim3(kBlock), 0, st, pl->args);
  }
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// Init-free solve: the first (apply, update) pair.  The apply reads `rhs` directly (launch 0 scales it by op->pre,
// copies it to r), leaves the partials of r . A r and ||r||^2 and resets the iteration state; the update then runs
// as iteration 1 with p = s = x = 0.  One launch (cg_init, ~3.8 us at N = 60k) less per solve.
int enqueue_first_body(CgPlan* pl, hipStream_t st, const float* rhs, bool record, bool decide_last = false) {
  MGP_TRY(mgp_operator_apply_first(&pl->op, rhs, pl->args.r, pl->args.w, pl->pd_delta, pl->pd_bb, pl->args.state,
                                   record ? pl->first_record : nullptr, pl->op_work, pl->op_work_bytes, st));
  launch_update_c1(pl, st, decide_last);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// stream-capture the first graph (cg_init / init-free first apply + len bodies) into *out; false on any failure
bool record_first(CgPlan* pl, int len, hipGraph_t* out) {
  *out = nullptr;
  if (!pl->cap_stream || len < 1) return false;
  if (hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  int rc = MGP_OK;
  // `len` = steps until the stopping rule fires: the last of them only detects (see cg_decide_c1_kernel)
  const bool decide = len >= 2 && !pl->is_dist && pl->C == 1 &&
                      pl->args.nbv <= kC1GammaSlots * kBlock && pl->args.nbs <= kC1DeltaSlots * kBlock;
  const int bodies = decide ? len - 1 : len;
  int done_bodies = 0;
  // decide: the graph's LAST update also takes the next step's decision and leaves the end-of-graph mark (g_cg_decide_in_update;
  // 0: the separate cg_decide_c1_kernel + cg_marker_kernel launches of rounds 1-3)
  const bool in_update = decide && pl->decide_in_update;
  if (pl->init_free) {
    // root node = launch 0 of the first apply, reading a placeholder rhs that is patched before every launch
    rc = enqueue_first_body(pl, pl->cap_stream, (const float*)pl->args.x, true, in_update && bodies == 1);
    done_bodies = 1;
  } else {
    hipLaunchKernelGGL(cg_init_kernel, dim3(pl->args.nbv), dim3(kBlock), 0, pl->cap_stream, pl->args,
                       (const float*)pl->args.x);   // placeholder rhs, patched before every launch
    rc = hipGetLastError() == hipSuccess ? MGP_OK : 1;
  }
  for (int i = done_bodies; i < bodies && rc == MGP_OK; ++i) rc = enqueue_body(pl, pl->cap_stream, in_update && i == bodies - 1);
  if (decide && !in_update && rc == MGP_OK) {
    hipLaunchKernelGGL(cg_decide_c1_kernel, dim3(1), dim3(kBlock), 0, pl->cap_stream, pl->args);
    rc = hipGetLastError() == hipSuccess ? MGP_OK : 1;
  }
  if (!in_update && rc == MGP_OK) {
    hipLaunchKernelGGL(cg_marker_kernel, dim3(1), dim3(1), 0, pl->cap_stream, pl->args.state, pl->args.host_state);
    rc = hipGetLastError() == hipSuccess ? MGP_OK : 1;
  }
  hipGraph_t graph = nullptr;
  const hipError_t e2 = hipStreamEndCapture(pl->cap_stream, &graph);
  if (rc == MGP_OK && e2 == hipSuccess && graph != nullptr) { *out = graph; return true; }
  if (graph) (void)hipGraphDestroy(graph);
  return false;
}

// (re)build the first graph: cg_init + len bodies.  Leaves has_first = false on any failure (the
// solve then launches cg_init eagerly and replays the continuation graph).
void capture_first(CgPlan* pl, int len) {
  if (pl->exec_first) { (void)hipGraphExecDestroy(pl->exec_first); pl->exec_first = nullptr; }
  if (pl->graph_first) { (void)hipGraphDestroy(pl->graph_first); pl->graph_first = nullptr; }
  pl->has_first = false;
  pl->patched_rhs = nullptr;
  pl->len_first = len;
  bool ok = record_first(pl, len, &pl->graph_first);
  if (ok) {
    size_t nroot = 1;
    hipGraphNode_t root = nullptr;
    ok = hipGraphGetRootNodes(pl->graph_first, &root, &nroot) == hipSuccess && nroot == 1 && root != nullptr;
    hipGraphNodeType ty;
    if (ok) ok = hipGraphNodeGetType(root, &ty) == hipSuccess && ty == hipGraphNodeTypeKernel;
    pl->init_node = root;
  }
  if (ok) ok = hipGraphInstantiate(&pl->exec_first, pl->graph_first, nullptr, nullptr, 0) == hipSuccess;
  (void)hipGetLastError();
  pl->has_first = ok;
}

// point the cg_init node of the first graph at this solve's right-hand side
bool patch_first_rhs(CgPlan* pl, const float* rhs) {
  if (rhs == pl->patched_rhs) return true;
  if (pl->init_free) {
    if (mgp_spmm_patch_node(pl->exec_first, pl->init_node, pl->first_record, (const float*)pl->args.x, rhs) != MGP_OK) {
      (void)hipGetLastError();
      return false;
    }
    pl->patched_rhs = rhs;
    return true;
  }
  hipKernelNodeParams np;
  memset(&np, 0, sizeof(np));
  if (hipGraphKernelNodeGetParams(pl->init_node, &np) != hipSuccess) return false;
  void* kp[2] = {(void*)&pl->args, (void*)&rhs};
  np.kernelParams = kp;
  np.extra = nullptr;
  if (hipGraphExecKernelNodeSetParams(pl->exec_first, pl->init_node, &np) != hipSuccess) return false;
  pl->patched_rhs = rhs;
  return true;
}

}  // namespace

extern "C" size_t mgp_cg_workspace_bytes(const mgp_operator_t* op, int C) {
  if (!op || C <= 0 || C > kMaxC || op->L.n <= 0) return 0;
  return cg_bytes(op, C);
}

// stream-capture `chunk` bodies (the continuation graph) into *out
static bool record_chunk(CgPlan* pl, hipGraph_t* out) {
  *out = nullptr;
  if (hipStreamBeginCapture(pl->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
  int rc = MGP_OK;
  for (int i = 0; i < pl->chunk && rc == MGP_OK; ++i) rc = enqueue_body(pl, pl->cap_stream);
  hipGraph_t graph = nullptr;
  const hipError_t e2 = hipStreamEndCapture(pl->cap_stream, &graph);
  if (rc == MGP_OK && e2 == hipSuccess && graph != nullptr) { *out = graph; return true; }
  if (graph) (void)hipGraphDestroy(graph);
  return false;
}

// After mgp_cg_plan_rebind: the captured graphs still hold the OLD operator's pointers and scalars.  Record the same launches
// again with the new ones and update the executable graphs in place (hipGraphExecUpdate: same topology, same kernels, new
// arguments -- tools/lab/graph_update.hip: record 15 us + update 20 us for 30 nodes, against instantiate 40-50 + the 210 us a
// hipGraphExecDestroy costs); an update the runtime refuses falls back to destroy + instantiate.  The first graph keeps its ORIGINAL
// hipGraph_t: the root node patched per solve (patch_first_rhs) is addressed through it, and the launch record it is patched from
// was rewritten by the recording.
static void refresh_graphs(CgPlan* pl) {
  pl->rebound = false;
  if (!pl->graphs_tried) return;          // nothing captured yet: the first capture will see the new operator
  (void)hipGetLastError();
  if (pl->has_graph) {
    hipGraph_t g = nullptr;
    bool ok = record_chunk(pl, &g);
    if (ok) {
      hipGraphNode_t err_node = nullptr;
      hipGraphExecUpdateResult res;
      if (hipGraphExecUpdate(pl->exec, g, &err_node, &res) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphExecDestroy(pl->exec);
        pl->exec = nullptr;
        ok = hipGraphInstantiate(&pl->exec, g, nullptr, nullptr, 0) == hipSuccess;
      }
    }
    if (g) (void)hipGraphDestroy(g);
    if (!ok && pl->exec) { (void)hipGraphExecDestroy(pl->exec); pl->exec = nullptr; }
    pl->has_graph = ok;
    (void)hipGetLastError();
  }
  if (pl->has_first) {
    hipGraph_t g = nullptr;
    bool ok = record_first(pl, pl->len_first, &g);
    if (ok) {
      hipGraphNode_t err_node = nullptr;
      hipGraphExecUpdateResult res;
      ok = hipGraphExecUpdate(pl->exec_first, g, &err_node, &res) == hipSuccess;
    }
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    pl->patched_rhs = nullptr;            // the update reset the root node to the placeholder right-hand side
    if (!ok) capture_first(pl, pl->len_first);
  }
}

// Graphs are captured at the SECOND solve of a plan: capture + instantiate cost ~0.3 ms (7 ms the first time in a
// process; tools/lab/time_capture.py), as much as a short solve, and plans built for a one-off solve never earn it
// back.  The first solve runs the same launches eagerly.
static void capture_graphs(CgPlan* pl) {
  pl->graphs_tried = true;
  if (!pl->prm.use_graph || pl->is_dist) return;   // collectives are enqueued eagerly (no capture)
  hipError_t e = hipStreamCreateWithFlags(&pl->cap_stream, hipStreamNonBlocking);
  bool ok = (e == hipSuccess);
  if (ok) {
    hipGraph_t graph = nullptr;
    ok = record_chunk(pl, &graph);
    if (ok) ok = hipGraphInstantiate(&pl->exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
  }
  pl->has_graph = ok;
  (void)hipGetLastError();   // a failed capture falls back to eager launches
  if (ok && !pl->cx) {      // (the complex-shift solve runs tens of iterations: init launch + chunk graphs, no single-graph form)
    int len = pl->last_need >= 1 && pl->last_need <= 64 ? pl->last_need : (pl->chunk < 4 ? pl->chunk : 4);
    capture_first(pl, len);
  }
}

static int plan_create_impl(const mgp_operator_t* op, int C, const float* minv, const mgp_cg_params_t* params,
                            const MgpDist* dist, void* work, size_t work_bytes, void* stream, void** plan_out) {
  if (!op || !params || !work || !plan_out) return MGP_ERR_ARG;
  if (C <= 0 || C > kMaxC) return C > kMaxC ? MGP_ERR_UNSUPPORTED : MGP_ERR_ARG;
  const int world = dist ? dist->world : 1;
  if (dist && (dist->n_loc != op->L.n || dist->world < 1 || dist->rank < 0 || dist->rank >= dist->world))
    return MGP_ERR_ARG;
  if (work_bytes < cg_bytes(op, C, world)) return MGP_ERR_WORKSPACE;
  CgPlan* pl = new (std::nothrow) CgPlan();
  if (!pl) return MGP_ERR_ARG;
  memset(pl, 0, sizeof(*pl));
  pl->op = *op;
  pl->is_dist = dist != nullptr;
  if (dist) pl->dist = *dist;
  pl->C = C;
  pl->prm = *params;
  if (pl->prm.max_iter <= 0) pl->prm.max_iter = 1000;
  if (pl->prm.min_iter < 0) pl->prm.min_iter = 0;
  pl->chunk = pl->prm.check_every > 0 ? pl->prm.check_every : 10;
  pl->stream = mgp_stream(stream);
  const int64_t n = op->L.n * world;        // global vector length
  const size_t nc = (size_t)n * C;
  MgpArena ar(work, work_bytes);
  CgArgs& a = pl->args;
  a.n = n; a.C = C; a.TC = tile_cols(C); a.TS = kBlock / a.TC;
  a.x = ar.take<float>(nc); a.r = ar.take<float>(nc);
  float* ubuf = ar.take<float>(nc);
  a.w = ar.take<float>(nc); a.p = ar.take<float>(nc); a.s = ar.take<float>(nc);
  float* usbuf = ar.take<float>(nc);
  a.minv = minv;
  a.u = minv ? ubuf : a.r;
  a.pre = op->pre;
  a.us = op->pre ? usbuf : nullptr;
  pl->op_work_bytes = 4 * mgp_align(nc * sizeof(float)) + 256;
  pl->op_work = ar.take<char>(pl->op_work_bytes);
  // contiguous row ranges per workgroup, at most kMaxGridVec workgroups.  C > 1: every workgroup of the
  // update kernel re-reduces ALL dot partials of ALL columns (nbv x (2 nbv + nbs) x C loads per launch),
  // so the grid is kept to one workgroup per CU (their loads go out in batches of 8 / 32 per lane)
  // C > 16: the partials are summed once by cg_reduce_kernel, the update grid is free to fill the chip
  const int reduce_mode = g_cg_reduce_once;      // (lab knobs: read once, at plan creation)
  pl->decide_in_update = g_cg_decide_in_update != 0;
  const bool reduce_once = reduce_mode && C > (reduce_mode == 2 ? 1 : kReduceOnceAbove);
  int max_grid_vec = (C == 1) ? kMaxGridVec : (reduce_once ? 2048 : 256);
  int64_t rpb = a.TS;
  // C % 4 == 0 behind cg_reduce_kernel (C > 16): the quad form of the update (cg_update_q_kernel).  Up to 16 columns, where every
  // workgroup re-reduces the partials, the element form stays: measured at 60k x 12 (tools/lab/cg12.py) 12.2 us against 15.4 for the
  // quad form in 1024-thread workgroups and 17.4 in 512-thread ones (the kernel can do it: lab switch below)
  pl->upd_quads = (C % 4 == 0 && reduce_once && g_cg_update_quads != 0) ? kBlock : 0;
  a.CQ = C / 4; a.TSQ = 0;
  if (pl->upd_quads) {
    a.TSQ = pl->upd_quads / a.CQ;
    rpb = a.TSQ;
  }
  int64_t nbv = mgp_cdiv(n, rpb);
  {
    const int64_t step = pl->upd_quads ? a.TSQ : a.TS;
    if (nbv > max_grid_vec) { rpb = mgp_cdiv(mgp_cdiv(n, max_grid_vec), step) * step; nbv = mgp_cdiv(n, rpb); }
  }
  a.rows_per_block = rpb; a.nbv = (int)nbv;
  a.pd_gamma = ar.take<float>(2 * (size_t)kMaxPartials * C);
  a.pd_rr = ar.take<float>(2 * (size_t)kMaxPartials * C);
  pl->nb_loc = mgp_spmm_dot_blocks_for(&op->L, C);
  a.nbs = pl->nb_loc * world;
  pl->pd_delta = ar.take<float>((size_t)a.nbs * C);
  a.pd_delta = pl->pd_delta;
  // one contiguous block: for C == 1 {gamma_old[2], alpha_old[2], bb, resid, state[0], state[1]} are 32
  // consecutive bytes, which the C == 1 kernels fetch with a single scalar load (CgScalars)
#ifdef MGP_STAMP
  float* blk = ar.take<float>(6 * (size_t)C + 16 + 1024 + 8192);
  MGP_HIP_TRY(hipMemsetAsync(blk, 0, (6 * (size_t)C + 16 + 1024 + 8192) * sizeof(float), pl->stream));
#else
  float* blk = ar.take<float>(6 * (size_t)C + 16);
#endif
  a.gamma_old = blk;
  a.alpha_old = blk + 2 * (size_t)C;
  a.bb = blk + 4 * (size_t)C;
  a.resid = blk + 5 * (size_t)C;
  a.state = reinterpret_cast<int*>(blk + 6 * (size_t)C);
  (void)hipMemsetAsync(a.state, 0, 16 * sizeof(int), pl->stream);      // (state[5]: cg_marker_kernel's count)
  {
    float* tot = ar.take<float>(3 * (size_t)C);
    a.tot = reduce_once ? tot : nullptr;
  }
  pl->xacc = ar.take<float>(nc); pl->rbuf = ar.take<float>(nc); pl->tbuf = ar.take<float>(nc);
  pl->rpart = ar.take<float>((size_t)256 * C * 2);
  pl->xacc64 = ar.take<double>(nc); pl->t64 = ar.take<double>(nc); pl->work64 = ar.take<double>(4 * nc);
  pl->rpart64 = ar.take<double>((size_t)256 * C * 2);
  a.tol = pl->prm.tol; a.max_iter = pl->prm.max_iter; a.min_iter = pl->prm.min_iter;
  a.stop_mode = pl->prm.stop_mode;
  pl->cx = false;
  if (C == 1 && !dist) {
    // the complex-shift solve's buffers (taken whenever the shape could use them: cg_bytes counts them)
    const size_t nn = (size_t)n;
    CxArgs& cx = pl->cxa;
    cx.z = ar.take<float2>(nn); cx.r = ar.take<float2>(nn); cx.p = ar.take<float2>(nn); cx.s = ar.take<float2>(nn);
    cx.u4 = ar.take<mgp_cg_v4f>(nn); cx.w4 = ar.take<mgp_cg_v4f>(nn);
    cx.y4 = ar.take<mgp_cg_v4f>(nn);
    pl->op_work4_bytes = 4 * mgp_align(nn * 16) + 256;
    pl->op_work4 = ar.take<char>(pl->op_work4_bytes);
    pl->pd4 = ar.take<float>((size_t)kCxDeltaSlots * kBlock * 4);
    cx.pd_g = ar.take<float>((size_t)2 * kMaxGridVec * 4);
    cx.sc = ar.take<float>(64);
    cx.pd4 = pl->pd4;
    const float cc = op->noise * op->scale;
    if (g_cg_complex_shift && !minv && op->form == 2 && op->nu == 2 && !op->pre && !op->post && cc > 0.f && ar.ok() &&
        a.nbv <= kC1GammaSlots * kBlock && pl->prm.stop_mode == 1) {
      pl->opB = *op;
      pl->opB.nu = 1;
      pl->opB.kappa = op->kappa / sqrtf(2.0f);       // tau_B = 2 / kappa_B^2 = 2 nu / kappa^2
      pl->opB.scale = 1.0f; pl->opB.form = 0; pl->opB.noise = 0.f;
      pl->nb4 = mgp_spmm_dot_blocks_for(&pl->opB.L, 4);
      if (pl->nb4 >= 1 && pl->nb4 <= kCxDeltaSlots * kBlock) {
        cx.nbs4 = pl->nb4;
        cx.sigma = sqrtf(cc);
        MGP_HIP_TRY(hipMemsetAsync(cx.sc, 0, 64 * sizeof(float), pl->stream));
        pl->cx = true;
      }
    }
  }
  pl->pd_bb = ar.take<float>((size_t)a.nbs * C);
  a.pd_bb = nullptr;
  a.arrive = ar.take<int>(9 * 32);
  if (a.arrive) MGP_HIP_TRY(hipMemsetAsync(a.arrive, 0, 9 * 32 * sizeof(int), pl->stream));
  pl->init_free = false;
  if (g_cg_init_free && C == 1 && !dist && !minv && !pl->cx && (op->form == 0 || op->form == 2) &&
      mgp_tile_plan(&op->L, 1, nullptr, nullptr, nullptr) && a.nbv <= kC1GammaSlots * kBlock &&
      a.nbs <= kC1DeltaSlots * kBlock) {
    pl->init_free = true;
    a.pd_bb = pl->pd_bb;
  }
  if (!ar.ok()) { delete pl; return MGP_ERR_WORKSPACE; }
  // one pooled host-mapped block: [16 int32 state | C resid | C true_rel], each part on its own 64-byte line
  const size_t rbytes = (((size_t)C * sizeof(float)) + 63) & ~size_t(63);
  if (!host_block_acquire(64 + 2 * rbytes, &pl->host_block)) { delete pl; return (int)hipErrorOutOfMemory; }
  char* hb = static_cast<char*>(pl->host_block.p);
  pl->host_state = reinterpret_cast<int32_t*>(hb);
  pl->host_resid = reinterpret_cast<float*>(hb + 64);
  pl->host_true_rel = reinterpret_cast<float*>(hb + 64 + rbytes);
  hipError_t e = hipHostGetDevicePointer((void**)&a.host_state, pl->host_state, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&a.host_resid, pl->host_resid, 0);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&pl->dev_true_rel, pl->host_true_rel, 0);
  if (e != hipSuccess) { host_block_release(pl->host_block, pl->stream); delete pl; return (int)e; }
  memset(hb, 0, 64 + 2 * rbytes);

  *plan_out = pl;
  return MGP_OK;
}

extern "C" int mgp_cg_set_init_free(int on) {
  g_cg_init_free = on ? 1 : 0;
  return MGP_OK;
}

extern "C" int mgp_cg_set_poll_spin(int spins) {
  g_cg_poll_spin = spins < 0 ? 0 : spins;
  return MGP_OK;
}

extern "C" int mgp_cg_set_reduce_once(int on) {
  g_cg_reduce_once = on == 2 ? 2 : (on ? 1 : 0);      // 2: from two columns up (A/B runs)
  return MGP_OK;
}

extern "C" int mgp_cg_set_update_quads(int on) {
  g_cg_update_quads = on ? 1 : 0;
  return MGP_OK;
}

extern "C" int mgp_cg_set_complex_shift(int on) {
  const int prev = g_cg_complex_shift;
  g_cg_complex_shift = on ? 1 : 0;
  return prev;
}

extern "C" int mgp_cg_plan_is_complex_shift(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  return pl && pl->cx ? 1 : 0;
}

extern "C" int mgp_cg_set_decide_in_update(int on) {
  const int prev = g_cg_decide_in_update;
  g_cg_decide_in_update = on ? 1 : 0;
  return prev;
}

extern "C" int mgp_cg_plan_create(const mgp_operator_t* op, int C, const float* minv,
                                  const mgp_cg_params_t* params, void* work, size_t work_bytes,
                                  void* stream, void** plan_out) {
  return plan_create_impl(op, C, minv, params, nullptr, work, work_bytes, stream, plan_out);
}

// see include/mgp_hip.h
extern "C" int mgp_cg_plan_rebind(void* plan, const mgp_operator_t* op, const float* minv) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl || !op) return MGP_ERR_ARG;
  if (pl->poisoned || pl->is_dist) return MGP_ERR_UNSUPPORTED;
  const mgp_operator_t& o = pl->op;
  // the same STRUCTURE: everything that chose kernels, grids and buffer sizes at creation
  const bool same = op->L.n == o.L.n && op->L.ncols == o.L.ncols && op->nu == o.nu && op->form == o.form &&
                    (op->pre != nullptr) == (o.pre != nullptr) && (op->post != nullptr) == (o.post != nullptr) &&
                    (minv != nullptr) == (pl->args.minv != nullptr) &&
                    (op->L.tile_ptr != nullptr) == (o.L.tile_ptr != nullptr) && op->L.tile_rows == o.L.tile_rows &&
                    op->L.tile_max_cols == o.L.tile_max_cols && op->L.tile_max_entries == o.L.tile_max_entries &&
                    (op->L.tile_rowptr != nullptr) == (o.L.tile_rowptr != nullptr) &&
                    (op->L.tile_vals != nullptr) == (o.L.tile_vals != nullptr) &&
                    (op->L.tile_rowid != nullptr) == (o.L.tile_rowid != nullptr) &&
                    (op->L.mt_img != nullptr) == (o.L.mt_img != nullptr) && op->L.mt_tiles == o.L.mt_tiles &&
                    op->L.mt_steps == o.L.mt_steps && mgp_spmm_dot_blocks_for(&op->L, pl->C) == pl->nb_loc;
  if (!same) return MGP_ERR_UNSUPPORTED;
  if (pl->cx) {
    const float cc = op->noise * op->scale;
    if (!(cc > 0.f)) return MGP_ERR_UNSUPPORTED;
    pl->opB = *op;
    pl->opB.nu = 1;
    pl->opB.kappa = op->kappa / sqrtf(2.0f);
    pl->opB.scale = 1.0f; pl->opB.form = 0; pl->opB.noise = 0.f;
    if (mgp_spmm_dot_blocks_for(&pl->opB.L, 4) != pl->nb4) return MGP_ERR_UNSUPPORTED;
    pl->cxa.sigma = sqrtf(cc);
  }
  pl->op = *op;
  pl->args.minv = minv;
  pl->args.pre = op->pre;
  pl->rebound = true;
  return MGP_OK;
}

extern "C" size_t mgp_cg_dist_workspace_bytes(const mgp_operator_t* op_local, int C, int world) {
  if (!op_local || C <= 0 || C > kMaxC || op_local->L.n <= 0 || world < 1) return 0;
  return cg_bytes(op_local, C, world);
}

extern "C" int mgp_cg_plan_create_dist(const mgp_operator_t* op_local, int C, const float* minv,
                                       const mgp_cg_params_t* params, void* comm, int rank, int world,
                                       void* work, size_t work_bytes, void* stream, void** plan_out) {
  if (!op_local || !comm) return MGP_ERR_ARG;
  MgpDist d{comm, rank, world, op_local->L.n, (int64_t)rank * op_local->L.n};
  return plan_create_impl(op_local, C, minv, params, &d, work, work_bytes, stream, plan_out);
}

// one CG run on `rhs` into the plan's x buffer; ends with the stream synchronised
static int run_cg(CgPlan* pl, const float* rhs, float* Xcopy) {
  hipStream_t st = pl->stream;
  const size_t nc = (size_t)pl->args.n * pl->C;
  pl->host_state[1] = 0;
  pl->host_state[3] = 0;
  *reinterpret_cast<volatile uint64_t*>(pl->host_state + 8) = 0;     // the deciding update's 8-byte record
  bool first = true;
  const auto t_begin = std::chrono::steady_clock::now();
  if (pl->rebound) refresh_graphs(pl);
  if (pl->solves++ >= 1 && !pl->graphs_tried) capture_graphs(pl);
  if (pl->has_first && !patch_first_rhs(pl, rhs)) pl->has_first = false;
  int eager_done = 0;           // bodies of the first eager chunk already enqueued
  if (!pl->has_first) {
    if (pl->cx) {
      hipLaunchKernelGGL(cx_init_kernel, dim3(pl->args.nbv), dim3(kBlock), 0, st, pl->args, pl->cxa, rhs);
      MGP_LAUNCH_CHECK();
    } else if (pl->init_free) {
      MGP_TRY(enqueue_first_body(pl, st, rhs, false));
      eager_done = 1;
    } else {
      hipLaunchKernelGGL(cg_init_kernel, dim3(pl->args.nbv), dim3(kBlock), 0, st, pl->args, rhs);
      MGP_LAUNCH_CHECK();
    }
  }
  int guard = 0;
  for (;;) {
    const bool launched_first = first && pl->has_first;
    if (launched_first) {
      ++pl->marker_seq;
      // (round 4, measured and removed: the first (apply, update) pair launched eagerly in front of a graph holding steps
      // 2 .. k, to hide the replay's launch latency behind it -- 59.9 us per 60k solve against 55.2 with the whole solve in
      // the graph, tools/lab/ab_decide.py: three eager launches and a graph launch behind them cost more than one replay)
      MGP_HIP_TRY(hipGraphLaunch(pl->exec_first, st));
    } else if (pl->has_graph) {
      MGP_HIP_TRY(hipGraphLaunch(pl->exec, st));
    } else {
      const int len = (first && pl->chunk > 4) ? 4 : pl->chunk;   // eager path: short solves stop early
      for (int i = first ? eager_done : 0; i < len; ++i) MGP_TRY(enqueue_body(pl, st));
    }
    first = false;
    // the solution rides behind every chunk so that one synchronisation ends the solve; the
    // convergence flag / residuals arrive through host-mapped memory written by the update kernel
    if (Xcopy) MGP_HIP_TRY(hipMemcpyAsync(Xcopy, pl->args.x, nc * sizeof(float), hipMemcpyDeviceToDevice, st));
    // The stopping decision arrives in host-mapped memory (written behind a system-scope fence by the
    // update kernel): poll it instead of sleeping in hipStreamSynchronize -- at ~70 us per solve the
    // wake-up latency of a blocking wait is a visible fraction.  Work queued behind the solve on the
    // same stream (the X copy, the caller's kernels) stays ordered; a chunk that ends undecided is
    // detected by the stream going idle.
    // hipStreamQuery is only the guard against a chunk that ends undecided: the first graph of a plan is sized to end
    // in the stopping decision, so for about twice as long as the previous solve took the host reads nothing but the
    // flag (measured: 60.5 -> 58.4 us per 60k solve with no query during the solve -- the queries themselves, a
    // runtime lock each, delay the launch's progress); continuation chunks poll as before.
    volatile int32_t* flag = pl->host_state + 1;
    // the deciding update launch (cg_update_c1_kernel<true>) reports through one 8-byte record instead of the flag words:
    // unpack it into them, so that everything below reads one format
    volatile uint64_t* record = reinterpret_cast<volatile uint64_t*>(pl->host_state + 8);
    auto decided = [&]() -> bool {
      if (*flag) return true;
      const uint64_t rec = *record;
      if (!(rec >> 32 & 1)) return false;
      const uint32_t lo = (uint32_t)rec, hi = (uint32_t)(rec >> 32);
      memcpy(pl->host_resid, &lo, sizeof(float));
      pl->host_state[0] = (int32_t)(hi >> 8);
      pl->host_state[2] = (int32_t)((hi >> 4) & 15);
      pl->host_state[3] = (int32_t)((hi >> 1) & 1);
      pl->host_state[1] = 1;
      return true;
    };
    const int poll_spin = g_cg_poll_spin;      // (lab knob: read once per chunk)
    if (launched_first && poll_spin > 0) {
      // (not even one query every 20 us: two or three of them during a 55 us solve took the whole gain back.)  The
      // graph's last node (cg_marker_kernel) reports a first graph that ran to its end undecided; the time budget --
      // ten times the last decided solve, at least 2 ms -- is only the guard against a marker that never comes.
      const int64_t budget = 10 * pl->last_solve_ns + 2000000;
      const auto t_spin = std::chrono::steady_clock::now();
      volatile int32_t* marker = pl->host_state + 4;
      while (!decided() && *marker != pl->marker_seq) {
        for (int spin = 0; spin < poll_spin && !decided() && *marker != pl->marker_seq; ++spin) __builtin_ia32_pause();
        if (std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_spin).count() > budget) break;
      }
    }
    if (!decided() && pl->is_dist && pl->dist.world > 1) {
      MGP_TRY(mgp_stream_wait_bounded(st));      // collectives on the stream: a dead peer must not hang this rank
    }
    while (!decided()) {
      const hipError_t q = hipStreamQuery(st);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return (int)q;
    }
    if (decided()) break;
    if (++guard > pl->prm.max_iter / (pl->chunk < 4 ? pl->chunk : 4) + 4) break;
  }
  // the first graph follows the workload: when two solves in a row needed the same number of steps and
  // it is not the captured length, re-capture (a few hundred us, once) so that the next solve of
  // this kind is exactly one graph launch with no skipped launches behind the stopping decision
  pl->last_solve_ns = 0;
  if (pl->host_state[1]) {
    if (guard == 0)      // decided inside the first chunk: how long such a solve takes (the next one's flag-only poll window)
      pl->last_solve_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_begin).count();
    const int need = pl->host_state[0];
    // longer than the captured graph: re-capture at once (an undecided first graph costs the flag-only window above
    // and a second launch); shorter: only when two solves in a row agree (the extra bodies of a graph that is one or
    // two steps too long return at their first load)
    if (pl->exec_first && need >= 1 && need <= 64 && (need > pl->len_first || (need < pl->len_first && need == pl->last_need))) {
      MGP_HIP_TRY(hipStreamSynchronize(st));   // the graph being replaced may still be draining
      capture_first(pl, need);
    }
    pl->last_need = need;
  }
  return MGP_OK;
}

static int cg_plan_solve_body(void* plan, const float* B, float* X, int32_t* iters, float* resid,
                                 int32_t* status) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl || !B) return MGP_ERR_ARG;   // X == NULL: leave the solution in the plan (mgp_cg_plan_x)
  hipStream_t st = pl->stream;
  const size_t nc = (size_t)pl->args.n * pl->C;
  const int max_refine = (pl->prm.stop_mode == 1) ? pl->prm.max_refine : 0;
  if (max_refine <= 0) {
    MGP_TRY(run_cg(pl, B, X));
    if (iters) *iters = pl->host_state[0] - 1;
    if (status) *status = pl->host_state[2];
    if (resid) memcpy(resid, pl->host_resid, (size_t)pl->C * sizeof(float));
    return MGP_OK;
  }
  int total_iters = 0, last_status = 0;
  const float* rhs = B;
  const int egrid = (int)(mgp_cdiv((int64_t)nc, kBlock) > 2048 ? 2048 : mgp_cdiv((int64_t)nc, kBlock));
  const bool f64 = !pl->is_dist;     // single GPU: accumulate x and form the true residual in fp64
  const int rgrid = 256;
  for (int ref = 0; ref <= max_refine; ++ref) {
    MGP_TRY(run_cg(pl, rhs, nullptr));
    total_iters += pl->host_state[0] - 1;
    last_status = pl->host_state[2];
    if (f64) {
      hipLaunchKernelGGL(refine_accumulate64_kernel, dim3(egrid), dim3(kBlock), 0, st, pl->xacc64, pl->args.x, (int64_t)nc,
                         ref == 0 ? 1 : 0);
      MGP_LAUNCH_CHECK();
      MGP_TRY(mgp_operator_apply_f64(&pl->op, pl->xacc64, pl->C, pl->t64, pl->work64, st));
      hipLaunchKernelGGL(refine_residual64_kernel, dim3(rgrid), dim3(kBlock), 0, st, B, pl->t64, pl->rbuf, pl->args.n, pl->C,
                         pl->rpart64);
      MGP_LAUNCH_CHECK();
      hipLaunchKernelGGL(refine_finalize64_kernel, dim3(1), dim3(kBlock), 0, st, pl->rpart64, rgrid, pl->C, pl->dev_true_rel);
      MGP_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(refine_accumulate_kernel, dim3(egrid), dim3(kBlock), 0, st, pl->xacc, pl->args.x, (int64_t)nc,
                         ref == 0 ? 1 : 0);
      MGP_LAUNCH_CHECK();
      // true residual R = B - A xacc
      MGP_TRY(mgp_operator_apply_dist(&pl->op, &pl->dist, pl->xacc, nullptr, pl->C, pl->tbuf, nullptr, nullptr, 0, nullptr,
                                      nullptr, pl->op_work, pl->op_work_bytes, st));
      hipLaunchKernelGGL(refine_residual_kernel, dim3(rgrid), dim3(kBlock), 0, st, B, pl->tbuf, pl->rbuf, pl->args.n,
                         pl->C, pl->rpart);
      MGP_LAUNCH_CHECK();
      hipLaunchKernelGGL(refine_finalize_kernel, dim3(1), dim3(kBlock), 0, st, pl->rpart, rgrid, pl->C, pl->dev_true_rel);
      MGP_LAUNCH_CHECK();
    }
    MGP_STREAM_WAIT(st, pl->is_dist && pl->dist.world > 1);
    bool ok = true;
    for (int c = 0; c < pl->C; ++c) ok = ok && (pl->host_true_rel[c] <= 2.0f * pl->prm.tol);
    if (ok || ref == max_refine || last_status == 3) break;
    rhs = pl->rbuf;
  }
  // publish the accumulated solution in the plan buffer (mgp_cg_plan_x) and, if asked, in X
  if (f64) {
    hipLaunchKernelGGL(refine_publish64_kernel, dim3(egrid), dim3(kBlock), 0, st, pl->xacc64, pl->args.x, X, (int64_t)nc);
    MGP_LAUNCH_CHECK();
  } else {
    MGP_HIP_TRY(hipMemcpyAsync(pl->args.x, pl->xacc, nc * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (X) MGP_HIP_TRY(hipMemcpyAsync(X, pl->xacc, nc * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  MGP_STREAM_WAIT(st, pl->is_dist && pl->dist.world > 1);
  if (iters) *iters = total_iters;
  if (status) *status = last_status;
  if (resid) memcpy(resid, pl->host_true_rel, (size_t)pl->C * sizeof(float));   // TRUE relative residuals
  return MGP_OK;
}

// operator applies the last (non-refined) solve actually ran: its step count, minus the detecting step when the
// stopping decision came from the decision-only launch at the end of the first graph
extern "C" int mgp_cg_plan_last_applies(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl) return MGP_ERR_ARG;
  return pl->host_state[0] - (pl->host_state[3] ? 1 : 0);
}

// device pointer of the plan's solution buffer [n, C] (valid until the plan is destroyed)
extern "C" float* mgp_cg_plan_x(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  return pl ? pl->args.x : nullptr;
}

#ifdef MGP_STAMP
// lab build: the stamp ring of the plan (256 x uint64: 8 * wall_clock64 + kind) and how many were written
extern "C" int mgp_cg_plan_debug_block_stamps(void* plan, unsigned long long* out, int nblocks) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl || !out || nblocks > 2048) return MGP_ERR_ARG;
  MGP_HIP_TRY(hipStreamSynchronize(pl->stream));
  MGP_HIP_TRY(hipMemcpy(out, reinterpret_cast<unsigned long long*>(pl->args.state + 16) + 256, (size_t)nblocks * 2 * sizeof(unsigned long long),
                        hipMemcpyDeviceToHost));
  return MGP_OK;
}
extern "C" int mgp_cg_plan_debug_stamps(void* plan, unsigned long long* out256, int* count) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl || !out256 || !count) return MGP_ERR_ARG;
  MGP_HIP_TRY(hipStreamSynchronize(pl->stream));
  MGP_HIP_TRY(hipMemcpy(count, pl->args.state + 8, sizeof(int), hipMemcpyDeviceToHost));
  MGP_HIP_TRY(hipMemcpy(out256, pl->args.state + 16, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return MGP_OK;
}
#endif

extern "C" double* mgp_cg_plan_x64(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  return (pl && !pl->is_dist) ? pl->xacc64 : nullptr;
}

// A solve that ends in MGP_ERR_TIMEOUT leaves kernels / collectives queued on the stream that still reference the plan's
// buffers, and the peer they wait for is gone: the plan is POISONED.  Destroying a poisoned plan frees nothing and
// synchronises nothing (hipFree / hipGraphExecDestroy / hipStreamDestroy would wait on the dead collective, which is the
// hang the timeout exists to avoid): the memory is leaked on purpose and the process is expected to exit non-zero.
extern "C" int mgp_cg_plan_solve(void* plan, const float* B, float* X, int32_t* iters, float* resid, int32_t* status) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (pl && pl->poisoned) return MGP_ERR_TIMEOUT;
  const int rc = cg_plan_solve_body(plan, B, X, iters, resid, status);
  if (rc == MGP_ERR_TIMEOUT && pl) pl->poisoned = true;
  return rc;
}

extern "C" int mgp_cg_plan_poisoned(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  return pl && pl->poisoned ? 1 : 0;
}

extern "C" int mgp_cg_plan_poison(void* plan) {      // a caller that learns by other means that a peer rank is gone
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl) return MGP_ERR_ARG;
  pl->poisoned = true;
  return MGP_OK;
}

extern "C" int mgp_cg_plan_destroy(void* plan) {
  CgPlan* pl = static_cast<CgPlan*>(plan);
  if (!pl) return MGP_ERR_ARG;
  if (pl->poisoned) return MGP_OK;      // leaked on purpose, see above
  if (pl->exec) (void)hipGraphExecDestroy(pl->exec);
  if (pl->exec_first) (void)hipGraphExecDestroy(pl->exec_first);
  if (pl->graph_first) (void)hipGraphDestroy(pl->graph_first);
  if (pl->cap_stream) (void)hipStreamDestroy(pl->cap_stream);
  host_block_release(pl->host_block, pl->stream);
  delete pl;
  return MGP_OK;
}

extern "C" int mgp_cg_solve(const mgp_operator_t* op, const float* B, int C, float* X, const float* minv,
                            const mgp_cg_params_t* params, int32_t* iters, float* resid, void* work,
                            size_t work_bytes, void* stream) {
  void* plan = nullptr;
  MGP_TRY(mgp_cg_plan_create(op, C, minv, params, work, work_bytes, stream, &plan));
  int32_t status = 0;
  int rc = mgp_cg_plan_solve(plan, B, X, iters, resid, &status);
  (void)mgp_cg_plan_destroy(plan);
  if (rc != MGP_OK) return rc;
  return status == 1 ? MGP_OK : MGP_ERR_NOT_CONVERGED;
}
