// Fused CSR SpMV / SpMM with the symmetric graph Laplacian (the hot kernel of the path).
//
//   xs_j = pre ? pre[j] * X[j,:] : X[j,:]
//   lx_i = diag[i] * xs_i - sum_j vals[ij] * xs_j
//   t_i  = (post ? post[i] : 1) * (a * xs_i + b * lx_i)
//   Y[i,:] = (base ? cb * base[i,:] : 0) + co * t_i ;  partial sums of dotw[i,:] * Y[i,:]
//
// Replaces GraphLaplacianOperator._matmul (manifold_gp/operators/graph_laplacian_operator.py:
// 108-124), i.e. 2 x torch_sparse.spmm (index_select + mul + atomic scatter_add) + ~5
// elementwise launches, by ONE launch without atomics over a full symmetric CSR.
//
// gfx950 mapping
//   * rows are padded to 4 entries and 16-byte aligned: every stream load is a 16-byte load;
//   * C == 1, default: spmv_tile_kernel -- 64-row tiles, the distinct columns of a tile staged in LDS
//     once (dictionary built by mgp_graph_tiles), per-entry gathers are ds_reads through 16-bit ids;
//   * C == 1, fallback (no dictionaries): spmv_kernel -- a row is owned by a G-lane sub-wave group
//     (G in 4..64 from the mean row length), gathers go to memory, shuffle reduction in the group;
//   * C  > 1: spmm_kernel -- a row is owned by G = pow2(min(C,64)) lanes laid over the right-hand-
//     side columns; the group loads G (col,val) pairs coalesced and broadcasts them (readlane ->
//     scalar base address when G = 64), 8 entries' X rows in flight per pass, every X row read as one
//     contiguous burst; dot partials go through LDS across the groups of a workgroup;
//   * grids are capped (<= 4096 workgroups, contiguous row ranges per workgroup) and remapped so
//     that each XCD streams one contiguous slice of rows (mgp_xcd_block).
#include <string.h>
#include <atomic>
#include "mgp_common.h"
#include "mgp_internal.h"
#include <hip/hip_ext.h>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 4096;

struct SpmmArgs {
  int64_t n;
  const int32_t* rowptr;
  const int32_t* col;
  const float* vals;
  const float* diag;
  const float* X;
  float* Y;
  int C;
  float a, b;
  const float* pre;
  const float* post;
  const float* base;
  float cb, co;
  const float* dotw;
  float* dot_partials;
  int64_t rows_per_block;
  const int* skip;   // nullable: workgroups return at once when *skip != 0 (CG converged)
  int* tick;         // nullable: workgroup 0 adds 1 when the launch is not skipped
  int64_t goff;      // row partition: CSR rows are local [0, n), vectors are global -> row r is
                     // element r + goff of X / Y / pre / post / base / dotw (0 on one GPU)
  // tile kernel only (init-free CG solve, cg.hip): the first apply of a solve reads the caller's right-hand side
  float* copy_x;          // nullable: the epilogue stores the row's raw input x[row] here (r = b)
  float* dot2_partials;   // nullable: per-workgroup partials of sum dotw[row]^2 (||b||^2)
  int tick_reset;         // tick != NULL: write {1, 0, 0} (iteration 1, not done, no status) instead of adding 1
#ifdef MGP_STAMP
  // lab build only: the stamps go BEHIND the skip / tick words, space that only cg.hip's CgPlan reserves (its state block
  // + 1024 + 8192 floats).  Other callers of the tile kernel with skip / tick (pcg.hip's 16-float state, the Lanczos and
  // dist plans) reserve nothing, so stamping is opt-in per process: tools/lab/stamp_solve.py calls mgp_stamp_enable(1)
  // and runs CgPlan solves only.
  int stamp_on = 0;
#endif
};

typedef int mgp_v4i __attribute__((ext_vector_type(4)));
typedef float mgp_v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int4 load_col4(const int32_t* base, int idx) {
  const mgp_v4i v = *reinterpret_cast<const mgp_v4i*>(base + idx);
  return make_int4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float4 load_val4(const float* base, int idx) {
  const mgp_v4f v = *reinterpret_cast<const mgp_v4f*>(base + idx);
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float epilogue(const SpmmArgs& p, int64_t r, int c, float xs, float acc) {
  const int64_t gr = r + p.goff;
  float lx = p.diag[r] * xs - acc;
  float t = p.a * xs + p.b * lx;
  if (p.post) t *= p.post[gr];
  float y = p.co * t;
  if (p.base) y += p.cb * p.base[gr * p.C + c];
  return y;
}

// ---------------------------------------------------------------- C == 1
// A G-lane group owns R consecutive rows and issues ALL their loads before the first use:
// R+1 row pointers, then R x (16 B of column ids + 16 B of values) per lane, then 4R gathers of x,
// then R shuffle reductions.  Three dependent memory phases per group instead of 3R: at N = 60k
// the kernel is latency-bound (the matrix is cache resident), so bytes in flight per lane is
// what sets the rate.  Rows longer than 4G entries take the (rare) remainder loop.
template <int G, int R, bool PRE>
__global__ __launch_bounds__(kBlock) void spmv_kernel(SpmmArgs p) {
  constexpr int BS = kBlock;
  if (p.skip && *p.skip) return;
  if (p.tick && blockIdx.x == 0 && threadIdx.x == 0) *p.tick += 1;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  constexpr int kGroups = BS / G;
  const int64_t r0 = (int64_t)lb * p.rows_per_block;
  int64_t r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  const float* __restrict__ x = p.X;
  float dsum = 0.f;
  for (int64_t rb = r0 + (int64_t)grp * R; rb < r1; rb += (int64_t)kGroups * R) {
    int s[R + 1];
#pragma unroll
    for (int t = 0; t <= R; ++t) {
      const int64_t rr = rb + t;
      s[t] = p.rowptr[rr < r1 ? rr : r1];
    }
    // epilogue operands of "my" row (lane t finishes row rb + t): in flight with everything else
    // every load below is UNCONDITIONAL on a clamped (always valid) address and masked by a
    // select afterwards: a load under `if` makes hipcc wait vmcnt(0) at each join, which
    // serialises the R rows (seen in the .s: one s_waitcnt vmcnt(0) per 16-B load)
    const int64_t myr = rb + lane;
    const bool mine = lane < R && myr < r1;
    const int64_t er = mine ? myr : r0;
    const int64_t ger = er + p.goff;
    float e_x = x[ger];
    if (PRE) e_x *= p.pre[ger];
    const float e_diag = p.diag[er];
    const float e_post = p.post ? p.post[ger] : 1.f;
    const float e_base = p.base ? p.base[ger] : 0.f;
    const float e_dotw = p.dotw ? p.dotw[ger] : 0.f;
    int4 c[R];
    float4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const int i = s[t] + 4 * lane;
      const bool on = i < s[t + 1];
      const int ii = on ? i : 0;
      c[t] = load_col4(p.col, ii);
      const float4 vv = load_val4(p.vals, ii);
      v[t] = make_float4(on ? vv.x : 0.f, on ? vv.y : 0.f, on ? vv.z : 0.f, on ? vv.w : 0.f);
    }
    // keep the phases apart: all 2R stream loads are in flight before the first gather issues
    __builtin_amdgcn_sched_barrier(0);
    float xg[R][4];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      xg[t][0] = x[c[t].x]; xg[t][1] = x[c[t].y]; xg[t][2] = x[c[t].z]; xg[t][3] = x[c[t].w];
      if (PRE) {
        xg[t][0] *= p.pre[c[t].x]; xg[t][1] *= p.pre[c[t].y]; xg[t][2] *= p.pre[c[t].z]; xg[t][3] *= p.pre[c[t].w];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      float a = v[t].x * xg[t][0];
      a = fmaf(v[t].y, xg[t][1], a);
      a = fmaf(v[t].z, xg[t][2], a);
      a = fmaf(v[t].w, xg[t][3], a);
      acc[t] = a;
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {
      for (int i = s[t] + 4 * lane + 4 * G; i < s[t + 1]; i += 4 * G) {
        const int4 cc = load_col4(p.col, i);
        const float4 vv = load_val4(p.vals, i);
        float x0 = x[cc.x], x1 = x[cc.y], x2 = x[cc.z], x3 = x[cc.w];
        if (PRE) { x0 *= p.pre[cc.x]; x1 *= p.pre[cc.y]; x2 *= p.pre[cc.z]; x3 *= p.pre[cc.w]; }
        acc[t] = fmaf(vv.x, x0, acc[t]);
        acc[t] = fmaf(vv.y, x1, acc[t]);
        acc[t] = fmaf(vv.z, x2, acc[t]);
        acc[t] = fmaf(vv.w, x3, acc[t]);
      }
    }
    float my_acc = 0.f;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const float tot = mgp_group_sum<G>(acc[t]);
      if (lane == t) my_acc = tot;
    }
    if (mine) {
      const float lx = e_diag * e_x - my_acc;
      const float tt = (p.a * e_x + p.b * lx) * e_post;
      const float y = p.co * tt + p.cb * e_base;
      p.Y[myr + p.goff] = y;
      dsum = fmaf(e_dotw, y, dsum);
    }
  }
  if (p.dot_partials) {
    __shared__ float red[BS / MGP_WAVE];
    dsum = mgp_wave_sum(dsum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < BS / MGP_WAVE; ++w) t += red[w];
      p.dot_partials[lb] = t;
    }
  }
}

// ---------------------------------------------------------------- C == 1, row tiles + LDS dictionary
// Counters on the 60k bench graph (profiles/r01_pmc_spmv.txt): the gather kernel above issues 3.9 M
// texture-cache accesses per launch, 3 M of them the per-entry gathers of x -- one tag lookup per
// lane per clock per CU, ~5 us of the 8.5 us launch -- although only 0.5 M requests go on to L2: the
// kernel is bound by the L1 lookup rate, not by bytes.  Here a workgroup owns a tile of BS/4 rows, stages
// the DISTINCT columns the tile references (mgp_graph_tiles: tile_cols) in LDS once -- a k-NN tile's
// rows share most neighbours, so that is a few hundred gathers instead of thousands -- and every
// per-entry gather becomes a ds_read through a 16-bit local id.  Stream: 4 B value + 2 B id per entry,
// fully coalesced (lane q owns quad q of the tile's contiguous entry range, rows are padded to quads).
//   phase 0  issue the stream loads of the first NQ quads per lane, row pointers, epilogue operands
//   phase 1  xl[j] = pre * x[tile_cols[j]]                        -> barrier
//   phase 2  part[q] = sum of the quad's 4 products (ds_read gathers) -> barrier
//   phase 3  4 lanes per row add the row's quads, xor-shuffle, lane 0 runs the epilogue
// The row sum order (quad-wise, then stride-4 lanes, then xor tree) does not depend on the tiling.
struct TileArgs {
  const int32_t* tile_ptr;
  const int32_t* tile_cols;
  const uint16_t* lid;
  int64_t ntiles;
  int tiles_per_block;
  int max_cols;
  // tiles over a row order (all NULL: position p of the tile order is row p of the CSR)
  const int32_t* rowptr_t;
  const float* vals_t;
  const int32_t* rowid;
  int max_entries;   // multi-column tile kernel: entries of the largest tile (LDS carving)
  int part_window;   // multi-column tile kernel: quads of partial sums staged per window (multiple of 256)
};

typedef unsigned short mgp_v4h __attribute__((ext_vector_type(4)));

template <bool PRE, int BS>
__global__ __launch_bounds__(BS) void spmv_tile_kernel(SpmmArgs p, TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  // the CG graph's skip flag / iteration tick: loaded here, consumed only after the first tile's
  // metadata loads have been issued, so that the flag costs no round trip of its own
  // The CG graph's skip flag / iteration tick are fetched through the VECTOR memory path, at the top of the tile loop, and
  // tested when the tile's own loads have all been issued.  As scalar loads at the head of the kernel (rounds 1-3) they cost
  // every launch a round trip beyond L2 (~0.5 us) in front of everything else: scalar loads return out of order, so the
  // first use of ANY later scalar -- the kernel arguments -- waits for lgkmcnt(0), flag included (s_load_dword +
  // s_waitcnt lgkmcnt(0) at the top of the .s).  `lane0` is a zero the compiler cannot see through (mbcnt is a divergent
  // source): a uniform address would be selected back to the scalar cache, and a provably uniform VALUE is moved to an SGPR
  // (v_readfirstlane behind s_waitcnt) in the loop preheader.
  const int lane0 = __builtin_amdgcn_mbcnt_lo(0u, 0u);
  const int* __restrict__ skip_ptr = (p.skip ? p.skip : reinterpret_cast<const int*>(p.rowptr)) + lane0;
  const int* __restrict__ tick_ptr = (p.tick ? p.tick : reinterpret_cast<const int*>(p.rowptr)) + lane0;
  int skipl = 0, tickl = 0;
  constexpr int TR = BS / 4;
  constexpr int NQ = 4;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, sub = tid & 3;
#ifdef MGP_STAMP   // lab build (tools/lab/stamp_solve.py): block 0 leaves its start / end time behind the CG state words
  int* st_base = !p.stamp_on ? nullptr : (p.skip ? const_cast<int*>(p.skip) - 1 : p.tick);
  if (st_base && blockIdx.x == 0 && tid == 0) {
    const int si = atomicAdd(st_base + 8, 1);
    reinterpret_cast<unsigned long long*>(st_base + 16)[si & 255] = wall_clock64() * 8 + 0;
  }
  if (st_base && tid == 0 && blockIdx.x < 2048) reinterpret_cast<unsigned long long*>(st_base + 16)[256 + 2 * blockIdx.x] = wall_clock64();
#endif
  float* __restrict__ xl = tile_lds;
  float* __restrict__ part = tile_lds + t.max_cols;
  // restrict-qualified read-only views: wave-uniform addresses become scalar (s_load) reads
  const float* __restrict__ x = p.X;
  const float* __restrict__ prev = p.pre;
  const int32_t* __restrict__ rowptr = t.rowptr_t ? t.rowptr_t : p.rowptr;   // CSR in tile order
  const float* __restrict__ vals = t.vals_t ? t.vals_t : p.vals;
  const int32_t* __restrict__ tile_ptr = t.tile_ptr;
  const uint32_t* __restrict__ tile_cols = reinterpret_cast<const uint32_t*>(t.tile_cols);
  const uint16_t* __restrict__ lid = t.lid;
  float dsum = 0.f, dsum2 = 0.f;
  const int64_t t0 = (int64_t)lb * t.tiles_per_block;
  const int64_t t1 = t0 + t.tiles_per_block < t.ntiles ? t0 + t.tiles_per_block : t.ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    skipl = *skip_ptr;
    tickl = *tick_ptr;
    const int64_t r0 = tile * TR;
    const int64_t r1 = r0 + TR < p.n ? r0 + TR : p.n;
    const int e0 = rowptr[r0], e1 = rowptr[r1];
    const int dp = tile_ptr[tile];
    const int D = tile_ptr[tile + 1] - dp;
    const int qb = e0 >> 2, Q = (e1 - e0) >> 2;
    // phase 0 (loads are retired in issue order, so the order below is the order they are needed in):
    // dictionary ids first, then the matrix stream, then the row-phase operands -- all unconditional on
    // clamped addresses, all in flight before the first wait
    unsigned c[NQ];   // unsigned: a signed id is sign-extended right behind its load, i.e. waited for at once
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int j = tid + k * BS;
      c[k] = (unsigned)tile_cols[j < D ? dp + j : 0];   // empty tile: dp may be one past the end
      // (round 4, measured and removed: a fixed-stride copy of the dictionaries, whose ids can be requested in the same round
      // trip as the tile's offsets instead of behind it -- 5.46 against 5.43 us back to back, 56.0 against 54.0 us per solve)
    }
    __builtin_amdgcn_sched_barrier(0);
    mgp_v4f v[NQ];
    mgp_v4h l[NQ];
#pragma unroll
    for (int k = 0; k < NQ / 2; ++k) {
      const int q = tid + k * BS;
      const int qi = q < Q ? qb + q : 0;
      v[k] = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)qi);
      l[k] = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)qi);
    }
    __builtin_amdgcn_sched_barrier(0);
    // phase 1: dictionary -> LDS (first NQ * BS columns from the ids loaded above).  The gathers sit
    // in the MIDDLE of the stream: loads retire in order, so the dictionary (and the first half of
    // the stream) is complete while the second half is still arriving and the LDS work overlaps it.
    float g[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      g[k] = x[c[k]];
      if (PRE) g[k] *= prev[c[k]];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = NQ / 2; k < NQ; ++k) {
      const int q = tid + k * BS;
      const int qi = q < Q ? qb + q : 0;
      v[k] = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)qi);
      l[k] = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)qi);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int64_t row = r0 + (tid >> 2);
    const bool valid = row < r1;
    const int64_t pr = valid ? row : r0;                                 // position in the tile order
    const int64_t rr = t.rowid ? (int64_t)t.rowid[pr] : pr;              // row of the CSR / of the vectors
    const int64_t grr = rr + p.goff;
    const int rs = rowptr[pr], re = rowptr[pr + 1];
    const float raw_x = x[grr];
    float e_x = raw_x;
    if (PRE) e_x *= prev[grr];
    const float e_diag = p.diag[rr];
    // nullable operands: unconditional load from a valid stand-in + select (no branch around a load)
    const float l_post = (p.post ? p.post : x)[grr], l_base = (p.base ? p.base : x)[grr], l_dotw = (p.dotw ? p.dotw : x)[grr];
    const float e_post = p.post ? l_post : 1.f;
    const float e_base = p.base ? l_base : 0.f;
    const float e_dotw = p.dotw ? l_dotw : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    if (p.skip && skipl) return;          // CG converged: every load above went to a valid address, nothing is written
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int j = tid + k * BS;
      if (j < D) xl[j] = g[k];
    }
    for (int j = tid + NQ * BS; j < D; j += BS) {      // dictionaries longer than NQ * BS (rare)
      const int cc = tile_cols[dp + j];
      float gg = x[cc];
      if (PRE) gg *= prev[cc];
      xl[j] = gg;
    }
    __syncthreads();
    // phase 2: quad products
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int q = tid + k * BS;
      if (q < Q) {
        float s = v[k].x * xl[l[k].x];
        s = fmaf(v[k].y, xl[l[k].y], s);
        s = fmaf(v[k].z, xl[l[k].z], s);
        s = fmaf(v[k].w, xl[l[k].w], s);
        part[q] = s;
      }
    }
    for (int q = tid + NQ * BS; q < Q; q += BS) {      // tiles with more than 4 * NQ * BS entries (rare)
      const mgp_v4f vv = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)(qb + q));
      const mgp_v4h ll = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)(qb + q));
      float s = vv.x * xl[ll.x];
      s = fmaf(vv.y, xl[ll.y], s);
      s = fmaf(vv.z, xl[ll.z], s);
      s = fmaf(vv.w, xl[ll.w], s);
      part[q] = s;
    }
    __syncthreads();
    // phase 3: rows
    float acc = 0.f;
    {
      int i = (rs >> 2) - qb + sub;
      const int e = (re >> 2) - qb;
      for (; i + 12 < e; i += 16) {        // four independent ds_reads in flight per pass
        const float a0 = part[i], a1 = part[i + 4], a2 = part[i + 8], a3 = part[i + 12];
        acc += a0; acc += a1; acc += a2; acc += a3;
      }
      for (; i < e; i += 4) acc += part[i];
    }
    acc = mgp_quad_sum(acc);
    if (valid && sub == 0) {
      const float lx = e_diag * e_x - acc;
      const float tt = (p.a * e_x + p.b * lx) * e_post;
      const float y = p.co * tt + p.cb * e_base;
      p.Y[grr] = y;
      dsum = fmaf(e_dotw, y, dsum);
      dsum2 = fmaf(e_dotw, e_dotw, dsum2);
      if (p.copy_x) p.copy_x[grr] = raw_x;
    }
    if (tile + 1 < t1) __syncthreads();                // the next tile overwrites xl / part
  }
  if (p.tick && blockIdx.x == 0 && tid == 0 && !(p.skip && skipl)) {
    if (p.tick_reset) { p.tick[0] = 1; p.tick[1] = 0; p.tick[2] = 0; }
    else *p.tick = tickl + 1;
  }
  if (p.dot_partials) {
    __shared__ float red[2][BS / MGP_WAVE];
    dsum = mgp_wave_sum(dsum);
    if (p.dot2_partials) dsum2 = mgp_wave_sum(dsum2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = dsum; red[1][tid >> 6] = dsum2; }
    __syncthreads();
    if (tid == 0) {
      float s = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < BS / MGP_WAVE; ++w) { s += red[0][w]; s2 += red[1][w]; }
      p.dot_partials[lb] = s;
      if (p.dot2_partials) p.dot2_partials[lb] = s2;
    }
  }
#ifdef MGP_STAMP
  if (st_base && blockIdx.x == 0 && tid == 0) {
    const int si = atomicAdd(st_base + 8, 1);
    reinterpret_cast<unsigned long long*>(st_base + 16)[si & 255] = wall_clock64() * 8 + 1;
  }
  if (st_base && tid == 0 && blockIdx.x < 2048) reinterpret_cast<unsigned long long*>(st_base + 16)[256 + 2 * blockIdx.x + 1] = wall_clock64();
#endif
}

// ---------------------------------------------------------------- C > 1
// G lanes per row laid over columns; NACC column accumulators per lane (columns lane + a*G).
template <int G, int NACC, bool PRE>
__global__ __launch_bounds__(kBlock) void spmm_kernel(SpmmArgs p) {
  if (p.skip && *p.skip) return;
  if (p.tick && blockIdx.x == 0 && threadIdx.x == 0) *p.tick += 1;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  constexpr int kGroups = kBlock / G;
  const int C = p.C;
  const int64_t r0 = (int64_t)lb * p.rows_per_block;
  int64_t r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  float dsum[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a) dsum[a] = 0.f;

  for (int64_t r = r0 + grp; r < r1; r += kGroups) {
    const int s = p.rowptr[r], e = p.rowptr[r + 1];
    float acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = 0.f;
    for (int i0 = s; i0 < e; i0 += G) {
      const int i = i0 + lane;
      int cj = (int)(r + p.goff);
      float vj = 0.f;
      if (i < e) { cj = p.col[i]; vj = p.vals[i]; }
      if (PRE) vj *= p.pre[cj];
      const int cnt = (e - i0) < G ? (e - i0) : G;
      // U entries per pass, all their X loads issued before the first FMA: with one entry at a time a
      // wave has a single L2 round trip in flight and a 60-entry row costs 60 of them back to back
      // (measured 370 us per 125-column SpMM at N = 60k = pure latency).  Lanes past the row's end hold
      // (own row, 0.0), so a pass may run over `cnt`; the FMA order is unchanged.
      constexpr int U = G >= 8 ? 8 : G;
      for (int t0 = 0; t0 < cnt; t0 += U) {
        float vts[U];
        float xv[U][NACC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = t0 + u;
          int ct;
          if (G == 64) {
            ct = __builtin_amdgcn_readlane(cj, t);
            vts[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vj), t));
          } else {
            ct = __shfl(cj, t, G);
            vts[u] = __shfl(vj, t, G);
          }
          const float* xr = p.X + (int64_t)ct * C;
#pragma unroll
          for (int a = 0; a < NACC; ++a) {
            const int c = lane + a * G;
            xv[u][a] = xr[c < C ? c : 0];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int a = 0; a < NACC; ++a) acc[a] = fmaf(vts[u], xv[u][a], acc[a]);
        }
      }
    }
    const int64_t gr = r + p.goff;
    const float prer = PRE ? p.pre[gr] : 1.f;
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      const int c = lane + a * G;
      if (c < C) {
        const float xs = p.X[gr * C + c] * prer;
        const float y = epilogue(p, r, c, xs, acc[a]);
        p.Y[gr * C + c] = y;
        if (p.dotw) dsum[a] = fmaf(p.dotw[gr * C + c], y, dsum[a]);
      }
    }
  }
  if (p.dot_partials) {
    // cross-group reduction through LDS: red[group][G*NACC]
    __shared__ float red[kBlock * NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a) red[(grp * NACC + a) * G + lane] = dsum[a];
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * NACC; idx += kBlock) {
      const int a = idx / G, l = idx % G;
      const int c = l + a * G;
      if (c < C) {
        float t = 0.f;
        for (int g = 0; g < kGroups; ++g) t += red[(g * NACC + a) * G + l];
        p.dot_partials[(int64_t)lb * C + c] = t;
      }
    }
  }
}

// value (.) float4 products of one quad of entries: sq = v.x a0 + v.y a1 + v.z a2 + v.w a3 (mul, then three fmas per
// component, in this order).  Written with v_pk_*_f32 and the op_sel broadcast of ONE half of the (x, y) / (z, w) value
// pair: hipcc selects the packed instructions by itself but materialises every multiplier as an (x, x) register pair
// (4 copies per value seen in the .s: 54 extra VGPRs in spmm_tile_q_kernel, which then ran at two workgroups per CU).
typedef float mgp_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ mgp_v2f pk_mul_lo(mgp_v2f s, mgp_v2f a) {
  mgp_v2f d;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(s), "v"(a));
  return d;
}
__device__ __forceinline__ mgp_v2f pk_fma_lo(mgp_v2f s, mgp_v2f a, mgp_v2f c) {
  mgp_v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s), "v"(a), "v"(c));
  return d;
}
__device__ __forceinline__ mgp_v2f pk_fma_hi(mgp_v2f s, mgp_v2f a, mgp_v2f c) {
  mgp_v2f d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(s), "v"(a), "v"(c));
  return d;
}
__device__ __forceinline__ mgp_v4f quad_products(mgp_v4f vv, mgp_v4f a0, mgp_v4f a1, mgp_v4f a2, mgp_v4f a3) {
  const mgp_v2f vlo = mgp_v2f{vv.x, vv.y}, vhi = mgp_v2f{vv.z, vv.w};
  mgp_v2f lo = pk_mul_lo(vlo, mgp_v2f{a0.x, a0.y}), hi = pk_mul_lo(vlo, mgp_v2f{a0.z, a0.w});
  lo = pk_fma_hi(vlo, mgp_v2f{a1.x, a1.y}, lo); hi = pk_fma_hi(vlo, mgp_v2f{a1.z, a1.w}, hi);
  lo = pk_fma_lo(vhi, mgp_v2f{a2.x, a2.y}, lo); hi = pk_fma_lo(vhi, mgp_v2f{a2.z, a2.w}, hi);
  lo = pk_fma_hi(vhi, mgp_v2f{a3.x, a3.y}, lo); hi = pk_fma_hi(vhi, mgp_v2f{a3.z, a3.w}, hi);
  return mgp_v4f{lo.x, lo.y, hi.x, hi.y};
}

// ---------------------------------------------------------------- 16 < C <= 256, C % 4 == 0: float4 lanes
// spmm_kernel above spends ~12 instructions per entry and 64 columns (two readlanes, 64-bit scalar address arithmetic,
// one 4-byte load and one fma per lane) against 4 cycles of texture-path time: it runs at half the rate the 64 B/clk
// L1 path allows (91 us at C = 128 on the 60k graph; 48 us of L1 time).  Here a lane owns one float4 of the row
// (LPR = pow2 >= C / 4 lanes per row, 64 / LPR rows per wave side by side); every lane of a row
// reads the row's (column, value) quads itself -- the same address across the row's lanes: one request -- and then the
// 16 bytes of each of the four X rows that are its own; two quads = eight X pieces per lane are in flight while the
// next quads' ids are fetched.  Per entry and 4 columns: a quarter of a 16-byte id load, one 64-bit address, one
// 16-byte load, two packed fmas.
template <int LPR, bool PRE>
__global__ __launch_bounds__(kBlock) void spmm_v4_kernel(SpmmArgs p) {
  if (p.skip && *p.skip) return;
  if (p.tick && blockIdx.x == 0 && threadIdx.x == 0) *p.tick += 1;
  constexpr int kGroups = kBlock / LPR;                 // rows a workgroup walks side by side
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int l = threadIdx.x & (LPR - 1), grp = threadIdx.x / LPR;
  const int C = p.C, C4 = C >> 2;
  const bool lon = l < C4;
  const int lc = lon ? l : 0;
  const int64_t r0 = (int64_t)lb * p.rows_per_block;
  int64_t r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  const mgp_v4f* __restrict__ X4 = reinterpret_cast<const mgp_v4f*>(p.X);
  const mgp_v4i* __restrict__ col4 = reinterpret_cast<const mgp_v4i*>(p.col);
  const mgp_v4f* __restrict__ val4 = reinterpret_cast<const mgp_v4f*>(p.vals);
  mgp_v4f dsum = mgp_v4f{0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0 + grp; r < r1; r += kGroups) {
    const int qs = p.rowptr[r] >> 2, qe = p.rowptr[r + 1] >> 2;      // rows are padded to quads of entries
    const int64_t gr = r + p.goff;
    // epilogue operands: in flight with the row walk
    const mgp_v4f e_x = X4[gr * C4 + lc];
    const float e_pre = PRE ? p.pre[gr] : 1.f;
    const float e_diag = p.diag[r];
    const float l_post = p.post ? p.post[gr] : 1.f;
    const mgp_v4f l_base = *reinterpret_cast<const mgp_v4f*>((p.base ? p.base : p.X) + gr * C + 4 * lc);
    const mgp_v4f l_dotw = *reinterpret_cast<const mgp_v4f*>((p.dotw ? p.dotw : p.X) + gr * C + 4 * lc);
    mgp_v4f acc = mgp_v4f{0.f, 0.f, 0.f, 0.f};
    if (qs < qe) {
      mgp_v4i c0 = col4[qs], c1 = col4[qs + 1 < qe ? qs + 1 : qs];
      mgp_v4f v0 = val4[qs], v1 = val4[qs + 1 < qe ? qs + 1 : qs];
      for (int q = qs; q < qe; q += 2) {
        const bool two = q + 1 < qe;
        mgp_v4f x[8];
        x[0] = X4[(int64_t)c0.x * C4 + lc]; x[1] = X4[(int64_t)c0.y * C4 + lc];
        x[2] = X4[(int64_t)c0.z * C4 + lc]; x[3] = X4[(int64_t)c0.w * C4 + lc];
        x[4] = X4[(int64_t)c1.x * C4 + lc]; x[5] = X4[(int64_t)c1.y * C4 + lc];
        x[6] = X4[(int64_t)c1.z * C4 + lc]; x[7] = X4[(int64_t)c1.w * C4 + lc];
        mgp_v4f w0 = v0, w1 = v1;
        if (PRE) {
          w0.x *= p.pre[c0.x]; w0.y *= p.pre[c0.y]; w0.z *= p.pre[c0.z]; w0.w *= p.pre[c0.w];
          w1.x *= p.pre[c1.x]; w1.y *= p.pre[c1.y]; w1.z *= p.pre[c1.z]; w1.w *= p.pre[c1.w];
        }
        if (!two) w1 = mgp_v4f{0.f, 0.f, 0.f, 0.f};
        // the next two quads' ids and values: requested before this pair's products wait for its X pieces
        const int qa = q + 2 < qe ? q + 2 : q, qb = q + 3 < qe ? q + 3 : qa;
        c0 = col4[qa]; c1 = col4[qb];
        v0 = val4[qa]; v1 = val4[qb];
        const mgp_v4f s0 = quad_products(w0, x[0], x[1], x[2], x[3]);
        const mgp_v4f s1 = quad_products(w1, x[4], x[5], x[6], x[7]);
        acc.x += s0.x; acc.y += s0.y; acc.z += s0.z; acc.w += s0.w;
        acc.x += s1.x; acc.y += s1.y; acc.z += s1.z; acc.w += s1.w;
      }
    }
    if (lon) {
      const float xs0 = e_x.x * e_pre, xs1 = e_x.y * e_pre, xs2 = e_x.z * e_pre, xs3 = e_x.w * e_pre;
      mgp_v4f y;
      y.x = p.co * ((p.a * xs0 + p.b * (e_diag * xs0 - acc.x)) * l_post) + (p.base ? p.cb * l_base.x : 0.f);
      y.y = p.co * ((p.a * xs1 + p.b * (e_diag * xs1 - acc.y)) * l_post) + (p.base ? p.cb * l_base.y : 0.f);
      y.z = p.co * ((p.a * xs2 + p.b * (e_diag * xs2 - acc.z)) * l_post) + (p.base ? p.cb * l_base.z : 0.f);
      y.w = p.co * ((p.a * xs3 + p.b * (e_diag * xs3 - acc.w)) * l_post) + (p.base ? p.cb * l_base.w : 0.f);
      *reinterpret_cast<mgp_v4f*>(p.Y + gr * C + 4 * l) = y;
      if (p.dotw) {
        dsum.x = fmaf(l_dotw.x, y.x, dsum.x); dsum.y = fmaf(l_dotw.y, y.y, dsum.y);
        dsum.z = fmaf(l_dotw.z, y.z, dsum.z); dsum.w = fmaf(l_dotw.w, y.w, dsum.w);
      }
    }
  }
  if (p.dot_partials) {
    // red[group][4 LPR]: the row groups' sums per column, added in group order by the first 4 LPR lanes
    __shared__ __attribute__((aligned(16))) float red[kBlock * 4];
    *reinterpret_cast<mgp_v4f*>(red + (grp * LPR + l) * 4) = dsum;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += kBlock) {
      float tsum = 0.f;
      for (int g = 0; g < kGroups; ++g) tsum += red[g * LPR * 4 + c];
      p.dot_partials[(int64_t)lb * C + c] = tsum;
    }
  }
}

// ---------------------------------------------------------------- C in {4, 8, 12, 16}
// Lanes are laid over ENTRIES, not over columns.  spmm_kernel above broadcasts every (col, val) pair to
// the lanes of a row group with __shfl, which hipcc lowers to ds_bpermute_b32: two LDS-pipe round trips
// per entry, ~45 us per launch at N = 60k for any C <= 16.  Here a row is owned by one 16-lane DPP row;
// LPE = 1 / 2 / 4 adjacent lanes share one entry and each loads ONE 16-byte quarter of the X row of its
// column (adjacent lanes -> one 64-byte texture access per entry), multiplies and accumulates its own
// partial sums over the passes of the row; at the end of the row the partials of the lanes holding the
// same quarter are summed with DPP row rotations (8, 4, 2, 1 down to LPE): no LDS, no broadcast.
// Lanes 0 .. C/4-1 then hold the row's C results, run the epilogue and store 16 bytes each.
template <int ROT>
__device__ __forceinline__ float dpp_ror_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + ROT, 0xf, 0xf, true));
}

template <int LPE>
__device__ __forceinline__ float row16_class_sum(float v) {   // sum over lanes l, l + LPE, l + 2 LPE, ... of a DPP row
  v = dpp_ror_add<8>(v);
  if (LPE <= 4) v = dpp_ror_add<4>(v);
  if (LPE <= 2) v = dpp_ror_add<2>(v);
  if (LPE <= 1) v = dpp_ror_add<1>(v);
  return v;
}

template <int C4, bool PRE>
__global__ __launch_bounds__(kBlock) void spmm_row16_kernel(SpmmArgs p) {
  if (p.skip && *p.skip) return;
  if (p.tick && blockIdx.x == 0 && threadIdx.x == 0) *p.tick += 1;
  constexpr int C = 4 * C4, G = 16, kGroups = kBlock / G;
  constexpr int LPE = C4 == 1 ? 1 : (C4 == 2 ? 2 : 4);   // lanes per entry
  constexpr int EP = G / LPE;                            // entries per pass of a row group
  constexpr int U = 4;                                   // passes in flight
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int lane = threadIdx.x & (G - 1);
  const int grp = threadIdx.x / G;
  const int q = lane % LPE, el = lane / LPE;
  const bool qon = q < C4;                               // C = 12: the fourth lane of an entry idles
  const int64_t r0 = (int64_t)lb * p.rows_per_block;
  int64_t r1 = r0 + p.rows_per_block;
  if (r1 > p.n) r1 = p.n;
  const mgp_v4f* __restrict__ X4 = reinterpret_cast<const mgp_v4f*>(p.X);
  mgp_v4f dsum = mgp_v4f{0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0 + grp; r < r1; r += kGroups) {
    const int s = p.rowptr[r], e = p.rowptr[r + 1];
    mgp_v4f acc = mgp_v4f{0.f, 0.f, 0.f, 0.f};
    for (int i0 = s; i0 < e; i0 += U * EP) {
      int cj[U];
      float vj[U];
      mgp_v4f xr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * EP + el;
        const bool on = i < e;
        const int ii = on ? i : s;               // clamped: unconditional loads (s < e here)
        const int cc = p.col[ii];
        const float vv = p.vals[ii];
        cj[u] = cc;
        vj[u] = on ? vv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (PRE) vj[u] *= p.pre[cj[u]];
        xr[u] = X4[(int64_t)cj[u] * C4 + (qon ? q : 0)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc.x = fmaf(vj[u], xr[u].x, acc.x); acc.y = fmaf(vj[u], xr[u].y, acc.y);
        acc.z = fmaf(vj[u], xr[u].z, acc.z); acc.w = fmaf(vj[u], xr[u].w, acc.w);
      }
    }
    acc.x = row16_class_sum<LPE>(acc.x); acc.y = row16_class_sum<LPE>(acc.y);
    acc.z = row16_class_sum<LPE>(acc.z); acc.w = row16_class_sum<LPE>(acc.w);
    if (el == 0 && qon) {                        // lanes 0 .. C4-1: columns 4q .. 4q+3 of this row
      const int64_t gr = r + p.goff;
      const float prer = PRE ? p.pre[gr] : 1.f;
      const mgp_v4f xin = X4[gr * C4 + q];
      mgp_v4f y;
      y.x = epilogue(p, r, 4 * q + 0, xin.x * prer, acc.x); y.y = epilogue(p, r, 4 * q + 1, xin.y * prer, acc.y);
      y.z = epilogue(p, r, 4 * q + 2, xin.z * prer, acc.z); y.w = epilogue(p, r, 4 * q + 3, xin.w * prer, acc.w);
      *reinterpret_cast<mgp_v4f*>(p.Y + gr * C + 4 * q) = y;
      if (p.dotw) {
        const float* dw = p.dotw + gr * C + 4 * q;
        dsum.x = fmaf(dw[0], y.x, dsum.x); dsum.y = fmaf(dw[1], y.y, dsum.y);
        dsum.z = fmaf(dw[2], y.z, dsum.z); dsum.w = fmaf(dw[3], y.w, dsum.w);
      }
    }
  }
  if (p.dot_partials) {
    __shared__ float red[kGroups][16];
    if (el == 0 && qon) { red[grp][4 * q] = dsum.x; red[grp][4 * q + 1] = dsum.y; red[grp][4 * q + 2] = dsum.z; red[grp][4 * q + 3] = dsum.w; }
    __syncthreads();
    if (threadIdx.x < C) {
      float t = 0.f;
      for (int g = 0; g < kGroups; ++g) t += red[g][threadIdx.x];
      p.dot_partials[(int64_t)lb * C + threadIdx.x] = t;
    }
  }
}

// ---------------------------------------------------------------- C in {4, 8, 12, 16}, row tiles + LDS dictionary
// The multi-column workloads of training run at 12 columns (the probes of the stochastic log-determinant, the
// inner solves of the Schur complement): spmm_row16_kernel above fetches a 64-byte X row PER ENTRY (3.7 M texture
// accesses, 235 MB through L1 / L2 per launch at N = 60k: 27 us).  With the tile dictionaries of the C = 1 kernel an X
// row is fetched once per TILE.  The kernel is spmv_tile_kernel with float4 elements, run as C / 4 PASSES over the
// tile ("quarters" of 4 columns): the matrix stream (values + 16-bit local ids, one quad of entries per lane and
// slot, fully coalesced) is loaded once into registers; per pass the quarter's dictionary goes to LDS (16 bytes per
// column; the next quarter's X rows are requested right behind it), every lane forms the float4 partial sums of its
// quads (ds_read_b128 gathers through the local ids) into LDS, four lanes per row add the row's quads, and lane
// `pass` of the row's quad runs the epilogue for its 4 columns and stores 16 bytes.  LDS: (max_cols + window) x 16
// bytes whatever C is, the partial sums staged `window` quads at a time (tile_small_window: 39.8 KB on the C3 graph
// whose largest tile has 1 209 columns and 7 468 entries, four workgroups per CU, <= 128 VGPRs).
// Measured on the C3 graph (tools/lab/ab_variants.py, round 2): 7.1 / 11.4 / 16.4 / 22.0 us at C = 4 / 8 / 12 / 16, and
// the same within 0.5 us at 2, 3 or 4 workgroups per CU, with all quarters of the dictionary prefetched into registers
// or one at a time, with the epilogue operands loaded per tile or per pass: a pass costs ~4.7 us whatever is
// overlapped with it.  Counters (tools/pmc_kernel.sh, C = 12): LDS array busy 12 k cycles per CU (51 % of them bank
// conflicts of the random gathers), 3.2 M L1 accesses, 2.1 M VALU wave-instructions, waves 42 % waiting on a
// counter and 20 % issuing: no unit is saturated, the phases between the three barriers of a pass do not overlap.
template <int C4, bool PRE>
__global__ __launch_bounds__(256, 4) void spmm_tile_q_kernel(SpmmArgs p, TileArgs t) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  constexpr int BS = 256, TR = 64, NQ = 4, C = 4 * C4;
  const int skipv = p.skip ? *p.skip : 0;
  const int tickv = (p.tick && blockIdx.x == 0) ? *p.tick : 0;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, sub = tid & 3;
  mgp_v4f* __restrict__ xl = reinterpret_cast<mgp_v4f*>(tile_lds);
  mgp_v4f* __restrict__ part = xl + t.max_cols;
  const mgp_v4f* __restrict__ X4 = reinterpret_cast<const mgp_v4f*>(p.X);
  const float* __restrict__ prev = p.pre;
  const int32_t* __restrict__ rowptr = t.rowptr_t ? t.rowptr_t : p.rowptr;
  const float* __restrict__ vals = t.vals_t ? t.vals_t : p.vals;
  const int32_t* __restrict__ tile_ptr = t.tile_ptr;
  const uint32_t* __restrict__ tile_cols = reinterpret_cast<const uint32_t*>(t.tile_cols);
  const uint16_t* __restrict__ lid = t.lid;
  mgp_v4f dsum = mgp_v4f{0.f, 0.f, 0.f, 0.f};
  const int64_t t0 = (int64_t)lb * t.tiles_per_block;
  const int64_t t1 = t0 + t.tiles_per_block < t.ntiles ? t0 + t.tiles_per_block : t.ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t r0 = tile * TR;
    const int64_t r1 = r0 + TR < p.n ? r0 + TR : p.n;
    const int e0 = rowptr[r0], e1 = rowptr[r1];
    const int dp = tile_ptr[tile];
    const int D = tile_ptr[tile + 1] - dp;
    const int qb = e0 >> 2, Q = (e1 - e0) >> 2;
    if (skipv) return;
    unsigned c[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int j = tid + k * BS;
      c[k] = (unsigned)tile_cols[j < D ? dp + j : 0];
    }
    __builtin_amdgcn_sched_barrier(0);
    mgp_v4f v[NQ];
    mgp_v4h l[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      const int q = tid + k * BS;
      const int qi = q < Q ? qb + q : 0;
      v[k] = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)qi);
      l[k] = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)qi);
    }
    __builtin_amdgcn_sched_barrier(0);
    // quarter 0 of the dictionary rows this lane stages (the ids are back by now: loads retire in order); quarter
    // u + 1 is requested while pass u runs.  (All C / 4 quarters up front cost 16 C / 4 registers: 192 VGPRs at
    // C = 12, two workgroups per CU and the 938 workgroups of the 60k graph in two rounds.)
    mgp_v4f g[NQ];
    float gp[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
      g[k] = X4[(int64_t)c[k] * C4];
      gp[k] = PRE ? prev[c[k]] : 1.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    // row operands: lane `sub` of a row's quad owns quarter `sub` (columns 4 sub .. 4 sub + 3)
    const int64_t row = r0 + (tid >> 2);
    const bool valid = row < r1;
    const int64_t pr = valid ? row : r0;
    const int64_t rr = t.rowid ? (int64_t)t.rowid[pr] : pr;
    const int64_t grr = rr + p.goff;
    const int rs = rowptr[pr], re = rowptr[pr + 1];
    const float e_pre = PRE ? prev[grr] : 1.f;
    const float e_diag = p.diag[rr];
    const float l_post = p.post ? p.post[grr] : 1.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < C4; ++u) {
      // epilogue operands of quarter u (used by lane u of the row's quad at the end of the pass; the four lanes read
      // the same 16 bytes): requested here, they are live for one pass instead of the whole tile
      const mgp_v4f e_x = X4[grr * C4 + u];
      const mgp_v4f l_base = *reinterpret_cast<const mgp_v4f*>((p.base ? p.base : p.X) + grr * C + 4 * u);
      const mgp_v4f l_dotw = *reinterpret_cast<const mgp_v4f*>((p.dotw ? p.dotw : p.X) + grr * C + 4 * u);
      // ---- quarter u of the dictionary -> LDS
#pragma unroll
      for (int k = 0; k < NQ; ++k) {
        const int j = tid + k * BS;
        if (j < D) {
          mgp_v4f w = g[k];
          if (PRE) { w.x *= gp[k]; w.y *= gp[k]; w.z *= gp[k]; w.w *= gp[k]; }
          xl[j] = w;
        }
      }
      if (u + 1 < C4) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) g[k] = X4[(int64_t)c[k] * C4 + (u + 1)];
      }
      for (int j = tid + NQ * BS; j < D; j += BS) {        // dictionaries longer than NQ * BS (rare)
        const unsigned cc = tile_cols[dp + j];
        mgp_v4f w = X4[(int64_t)cc * C4 + u];
        if (PRE) { const float sc = prev[cc]; w.x *= sc; w.y *= sc; w.z *= sc; w.w *= sc; }
        xl[j] = w;
      }
      // ---- quad partial sums of this quarter, staged `W` quads at a time (one window for most tiles: the staging
      // area is sized so that four workgroups share a CU's LDS -- with the largest tile's 1 867 quads staged at once it
      // was three, and the 938 workgroups of the 60k graph ran as a full round plus a 170-workgroup tail)
      mgp_v4f acc = mgp_v4f{0.f, 0.f, 0.f, 0.f};
      const int W = t.part_window;
      for (int w0 = 0; w0 == 0 || w0 < Q; w0 += W) {
        const int w1 = Q < w0 + W ? Q : w0 + W;
        __syncthreads();     // first window: the dictionary is staged; later ones: the rows have read the previous window
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
          const int q = tid + k * BS;
          if (q >= w0 && q < w1) {
            const mgp_v4f a0 = xl[l[k].x], a1 = xl[l[k].y], a2 = xl[l[k].z], a3 = xl[l[k].w];
            part[q - w0] = quad_products(v[k], a0, a1, a2, a3);
          }
          // one slot's four ds_read_b128 in flight at a time: with all 16 hoisted (64 registers) the kernel needs
          // ~190 VGPRs; the other waves of the CU cover the LDS latency instead
          if (k & 1) __builtin_amdgcn_sched_barrier(0);
        }
        {                                                    // quads past the NQ * BS held in registers
          int q = tid + NQ * BS;
          if (q < w0) q += (w0 - q + BS - 1) / BS * BS;
          for (; q < w1; q += BS) {
            const mgp_v4f vv = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)(qb + q));
            const mgp_v4h ll = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)(qb + q));
            const mgp_v4f a0 = xl[ll.x], a1 = xl[ll.y], a2 = xl[ll.z], a3 = xl[ll.w];
            part[q - w0] = quad_products(vv, a0, a1, a2, a3);
          }
        }
        __syncthreads();
        // ---- rows: four lanes add the row's quads (stride 4) that lie in this window, in increasing order
        {
          int i = (rs >> 2) - qb + sub;
          if (i < w0) i += (w0 - i + 3) / 4 * 4;
          int e = (re >> 2) - qb;
          if (e > w1) e = w1;
          for (; i + 4 < e; i += 8) {
            const mgp_v4f a0 = part[i - w0], a1 = part[i + 4 - w0];
            acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
            acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
          }
          for (; i < e; i += 4) {
            const mgp_v4f a0 = part[i - w0];
            acc.x += a0.x; acc.y += a0.y; acc.z += a0.z; acc.w += a0.w;
          }
        }
      }
      acc.x = mgp_quad_sum(acc.x); acc.y = mgp_quad_sum(acc.y); acc.z = mgp_quad_sum(acc.z); acc.w = mgp_quad_sum(acc.w);
      if (valid && sub == u) {
        const float xs0 = e_x.x * e_pre, xs1 = e_x.y * e_pre, xs2 = e_x.z * e_pre, xs3 = e_x.w * e_pre;
        mgp_v4f y;
        y.x = p.co * ((p.a * xs0 + p.b * (e_diag * xs0 - acc.x)) * l_post) + (p.base ? p.cb * l_base.x : 0.f);
        y.y = p.co * ((p.a * xs1 + p.b * (e_diag * xs1 - acc.y)) * l_post) + (p.base ? p.cb * l_base.y : 0.f);
        y.z = p.co * ((p.a * xs2 + p.b * (e_diag * xs2 - acc.z)) * l_post) + (p.base ? p.cb * l_base.z : 0.f);
        y.w = p.co * ((p.a * xs3 + p.b * (e_diag * xs3 - acc.w)) * l_post) + (p.base ? p.cb * l_base.w : 0.f);
        *reinterpret_cast<mgp_v4f*>(p.Y + grr * C + 4 * u) = y;
        if (p.dotw) {
          dsum.x = fmaf(l_dotw.x, y.x, dsum.x); dsum.y = fmaf(l_dotw.y, y.y, dsum.y);
          dsum.z = fmaf(l_dotw.z, y.z, dsum.z); dsum.w = fmaf(l_dotw.w, y.w, dsum.w);
        }
      }
      if (u + 1 < C4 || tile + 1 < t1) __syncthreads();    // the next pass / tile overwrites xl and part
    }
  }
  if (p.tick && blockIdx.x == 0 && tid == 0 && !skipv) *p.tick = tickv + 1;
  if (p.dot_partials) {
    // red[row][column]: lane `sub` of every row leaves its quarter, then C threads add the 64 rows in a fixed order
    __syncthreads();
    float* red = tile_lds;
    if (sub < C4) *reinterpret_cast<mgp_v4f*>(red + (size_t)(tid >> 2) * C + 4 * sub) = dsum;
    __syncthreads();
    if (tid < C) {
      float sacc = 0.f;
      for (int r = 0; r < TR; ++r) sacc += red[r * C + tid];
      p.dot_partials[(int64_t)lb * C + tid] = sacc;
    }
  }
}

// ---------------------------------------------------------------- 16 < C <= 256, C % 4 == 0: row tiles + LDS dictionary
// The eigensolver's 128-column blocks and the 100 right-hand sides of `_average_variance` ran on spmm_kernel above: one
// 256-byte piece of an X row per entry and 64 columns through L1 -- 1.9 GB per launch at C = 128 on the 60k graph, 99 us,
// 343 launches = two thirds of an eigensolve.  A tile's rows share their neighbours (486 distinct columns for 3 910
// entries), so here an X row goes to LDS once per tile and 16-column CHUNK, and the per-entry gather is a ds_read_b128:
//   * a workgroup owns a 64-row tile, wave w its rows 16 w .. 16 w + 15, four lanes per row, lane `sub` the float4
//     4 sub .. 4 sub + 3 of the chunk: the 16 rows of a wave advance together, one entry per step;
//   * per chunk: the dictionary's X pieces (64 bytes per column, x pre) are staged by all 256 lanes (four lanes per
//     column: 64-byte segments), barrier, then every lane walks its row's quads -- 16 bytes of values + 8 bytes of local
//     ids per 4 entries straight from global memory (the four lanes of a row read the same address; the next quad is
//     requested before the current one is used), four ds_read_b128 + packed fmas per quad -- and finishes with the
//     epilogue of its 4 columns (16-byte store: the four lanes of a row write 64 contiguous bytes);
//   * no partial sums through LDS and no barrier inside a chunk's row walk (the 12-column kernel's three phases per
//     pass did not overlap); dot partials: the chunk's 64 x 16 products cross LDS once per chunk;
//   * LDS: 64 bytes x min(dictionary, 1024 columns): 64 KB on the C3 graph (two workgroups per CU), 40 KB on the 1M swiss
//     roll.  The 0.7 % of tiles with more than 1024 columns take a second slice pass per chunk (entries whose id lies
//     outside the staged slice contribute value 0 through a clamped id).
constexpr int kWideCap = 1024;     // dictionary columns staged per slice (64 bytes each)
constexpr int kWideRQ = 16;        // quads of a row held in registers for all chunks of a tile

template <bool PRE>
__global__ __launch_bounds__(256, 2) void spmm_tile_wide_kernel(SpmmArgs p, TileArgs t, int dict_cap) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  constexpr int BS = 256, TR = 64;
  const int skipv = p.skip ? *p.skip : 0;
  const int tickv = (p.tick && blockIdx.x == 0) ? *p.tick : 0;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, sub = tid & 3;
  const int C = p.C, C4 = C >> 2;
  mgp_v4f* __restrict__ xl = reinterpret_cast<mgp_v4f*>(tile_lds);
  const mgp_v4f* __restrict__ X4 = reinterpret_cast<const mgp_v4f*>(p.X);
  const float* __restrict__ prev = p.pre;
  const int32_t* __restrict__ rowptr = t.rowptr_t ? t.rowptr_t : p.rowptr;
  const float* __restrict__ vals = t.vals_t ? t.vals_t : p.vals;
  const int32_t* __restrict__ tile_ptr = t.tile_ptr;
  const uint32_t* __restrict__ tile_cols = reinterpret_cast<const uint32_t*>(t.tile_cols);
  const uint16_t* __restrict__ lid = t.lid;
  if (skipv) return;
  const int64_t t0 = (int64_t)lb * t.tiles_per_block;
  const int64_t t1 = t0 + t.tiles_per_block < t.ntiles ? t0 + t.tiles_per_block : t.ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t r0 = tile * TR;
    const int64_t r1 = r0 + TR < p.n ? r0 + TR : p.n;
    const int dp = tile_ptr[tile];
    const int D = tile_ptr[tile + 1] - dp;
    const int64_t row = r0 + (tid >> 2);
    const bool valid = row < r1;
    const int64_t pr = valid ? row : r0;
    const int64_t rr = t.rowid ? (int64_t)t.rowid[pr] : pr;
    const int64_t grr = rr + p.goff;
    const int qs = rowptr[pr] >> 2, qe = valid ? rowptr[pr + 1] >> 2 : qs;      // the row's quads (rows are padded to quads)
    const float e_pre = PRE ? prev[grr] : 1.f;
    const float e_diag = p.diag[rr];
    const float l_post = p.post ? p.post[grr] : 1.f;
    // the first RQ quads of the row (64 entries: most rows whole) stay in registers for all chunks of the tile: loaded
    // once, all in flight together.  (Loaded per chunk, one quad ahead of its use, every step of the row walk waited
    // out a memory round trip: 29 us per chunk.)
    mgp_v4f rv[kWideRQ];
    mgp_v4h rl[kWideRQ];
#pragma unroll
    for (int k = 0; k < kWideRQ; ++k) {
      const int qi = qs + k < qe ? qs + k : qs;
      rv[k] = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)qi);
      rl[k] = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)qi);
    }
    for (int c4 = 0; c4 < C4; c4 += 4) {
      const int f = c4 + sub;                       // this lane's float4 of the X / Y rows
      const bool fon = f < C4;
      const int fc = fon ? f : c4;
      // epilogue operands of this chunk: in flight with the staging below
      const mgp_v4f e_x = X4[grr * C4 + fc];
      const mgp_v4f l_base = *reinterpret_cast<const mgp_v4f*>((p.base ? p.base : p.X) + grr * C + 4 * fc);
      const mgp_v4f l_dotw = *reinterpret_cast<const mgp_v4f*>((p.dotw ? p.dotw : p.X) + grr * C + 4 * fc);
      mgp_v4f acc = mgp_v4f{0.f, 0.f, 0.f, 0.f};
      for (int s0 = 0; s0 < D || s0 == 0; s0 += dict_cap) {
        const int Ds = D - s0 < dict_cap ? D - s0 : dict_cap;
        // ---- stage the slice's X pieces: lane (j, sub) -> 16 bytes, 8 pieces per lane in flight
        for (int i0 = tid; i0 < 4 * Ds; i0 += 8 * BS) {
          unsigned cc[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * BS;
            cc[k] = tile_cols[dp + s0 + (i < 4 * Ds ? i >> 2 : 0)];
          }
          mgp_v4f w[8];
          float sc[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            w[k] = X4[(int64_t)cc[k] * C4 + fc];
            sc[k] = PRE ? prev[cc[k]] : 1.f;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * BS;
            if (i < 4 * Ds) {
              mgp_v4f v = w[k];
              if (PRE) { v.x *= sc[k]; v.y *= sc[k]; v.z *= sc[k]; v.w *= sc[k]; }
              xl[i] = fon ? v : mgp_v4f{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
        __syncthreads();
        // ---- the row's quads: ids outside the staged slice -> clamped id, value 0 (only tiles with more than
        // dict_cap columns take more than one slice)
        auto quad = [&](const mgp_v4f vq, const mgp_v4h lq) __attribute__((always_inline)) {
          const unsigned i0 = (unsigned)lq.x - (unsigned)s0, i1 = (unsigned)lq.y - (unsigned)s0,
                         i2 = (unsigned)lq.z - (unsigned)s0, i3 = (unsigned)lq.w - (unsigned)s0;
          const bool o0 = i0 < (unsigned)Ds, o1 = i1 < (unsigned)Ds, o2 = i2 < (unsigned)Ds, o3 = i3 < (unsigned)Ds;
          const mgp_v4f a0 = xl[(o0 ? i0 : 0u) * 4 + sub], a1 = xl[(o1 ? i1 : 0u) * 4 + sub],
                        a2 = xl[(o2 ? i2 : 0u) * 4 + sub], a3 = xl[(o3 ? i3 : 0u) * 4 + sub];
          const mgp_v4f vv = mgp_v4f{o0 ? vq.x : 0.f, o1 ? vq.y : 0.f, o2 ? vq.z : 0.f, o3 ? vq.w : 0.f};
          const mgp_v4f sq = quad_products(vv, a0, a1, a2, a3);
          acc.x += sq.x; acc.y += sq.y; acc.z += sq.z; acc.w += sq.w;
        };
        // (no branch per quad: a quad past the row's end holds a copy of the row's first quad and counts with value 0,
        // so that the LDS reads of several quads are in flight together)
#pragma unroll
        for (int k = 0; k < kWideRQ; ++k) {
          const bool on = qs + k < qe;
          quad(mgp_v4f{on ? rv[k].x : 0.f, on ? rv[k].y : 0.f, on ? rv[k].z : 0.f, on ? rv[k].w : 0.f}, rl[k]);
        }
        // rows longer than 4 RQ entries: the rest in batches of four quads from global memory, their loads in flight
        // together (a memory round trip per batch and chunk: 60 of 158 us at C = 128 on the 60k graph, whose tiles
        // nearly all hold such a row -- keeping these tails in LDS, walked quad by quad, was no faster)
        for (int q = qs + kWideRQ; q < qe; q += 4) {
          mgp_v4f bv[4];
          mgp_v4h bl[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int qi = q + k < qe ? q + k : q;
            bv[k] = *reinterpret_cast<const mgp_v4f*>(vals + 4 * (int64_t)qi);
            bl[k] = *reinterpret_cast<const mgp_v4h*>(lid + 4 * (int64_t)qi);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (q + k < qe) quad(bv[k], bl[k]);
        }
        __syncthreads();                               // the next slice / chunk / tile overwrites xl
      }
      // ---- epilogue of this lane's 4 columns
      mgp_v4f y = mgp_v4f{0.f, 0.f, 0.f, 0.f};
      if (valid && fon) {
        const float xs0 = e_x.x * e_pre, xs1 = e_x.y * e_pre, xs2 = e_x.z * e_pre, xs3 = e_x.w * e_pre;
        y.x = p.co * ((p.a * xs0 + p.b * (e_diag * xs0 - acc.x)) * l_post) + (p.base ? p.cb * l_base.x : 0.f);
        y.y = p.co * ((p.a * xs1 + p.b * (e_diag * xs1 - acc.y)) * l_post) + (p.base ? p.cb * l_base.y : 0.f);
        y.z = p.co * ((p.a * xs2 + p.b * (e_diag * xs2 - acc.z)) * l_post) + (p.base ? p.cb * l_base.z : 0.f);
        y.w = p.co * ((p.a * xs3 + p.b * (e_diag * xs3 - acc.w)) * l_post) + (p.base ? p.cb * l_base.w : 0.f);
        *reinterpret_cast<mgp_v4f*>(p.Y + grr * C + 4 * f) = y;
      }
      if (p.dot_partials) {
        // red[row][16]: the chunk's products of the tile's 64 rows, then 16 lanes add the rows in a fixed order
        // (tiles_per_block > 1: accumulated over the workgroup's tiles in tile order)
        mgp_v4f d = mgp_v4f{0.f, 0.f, 0.f, 0.f};
        if (valid && fon && p.dotw) { d.x = l_dotw.x * y.x; d.y = l_dotw.y * y.y; d.z = l_dotw.z * y.z; d.w = l_dotw.w * y.w; }
        xl[tid] = d;                                   // tid = row-in-tile * 4 + sub
        __syncthreads();
        if (tid < 16 && c4 * 4 + tid < C) {
          const float* red = tile_lds;
          float sacc = 0.f;
          for (int r = 0; r < TR; ++r) sacc += red[r * 16 + tid];
          float* dst = p.dot_partials + (int64_t)lb * C + c4 * 4 + tid;
          *dst = (tile == t0 ? 0.f : *dst) + sacc;
        }
        __syncthreads();
      }
    }
  }
  if (p.tick && blockIdx.x == 0 && tid == 0) *p.tick = tickv + 1;
}

// ---------------------------------------------------------------- 16 < C <= 256 on tile dictionaries, LANES OVER COLUMNS
// (round 3).  What the counters said about the per-entry gather kernels at C = 128 on the 60k graph
// (profiles/r02_pmc_spmm_wide_kernel_block.txt): 33.8 M 64-byte L1 accesses = 2.16 GB per launch, 24x the algorithmic
// bytes -- every ENTRY fetches its 512-byte X row through the vector memory path (64 B/clk/CU: 48 us of a 91 us launch).
// A 64-row tile references only ~486 distinct X rows for its ~3 900 entries, so here an X row crosses the vector memory
// path once per TILE (233 MB per launch instead of 1.9 GB) and every entry reads it from LDS.  The two earlier
// dictionary kernels did that too and lost to LDS bank conflicts: they laid lanes over ENTRIES (each lane a 16-byte piece
// of a different dictionary slot: random 16- and 32-byte reads collide about three deep).  Here the 16 lanes of a group
// lie over the COLUMNS of one dictionary slot: a slot is NV x 256 bytes, 256-byte aligned, lane l reads bytes
// [16 l, 16 l + 16) of each 256-byte piece.  Whatever slots the four groups of a wave address, every lane group of a
// ds_read_b128 (MI355X_MICROARCH.md, LDS table) then covers all 64 banks exactly once: conflict-free at 256 B/clk/CU.
//   * one 1024-thread workgroup per tile: group g (16 lanes) owns row g of the tile, its accumulators (NV float4 per
//     lane) stay in registers over the whole tile;
//   * the tile's matrix stream (values + 16-bit dictionary ids, 6 B per entry) is copied to LDS once, coalesced; a group
//     reads its row's quads from there (same address in all 16 lanes: a broadcast);
//   * the dictionary does not fit LDS at these widths (486 x 512 B), so it is staged in SLICES of S slots, double
//     buffered: the X rows of slice k + 1 are in flight (registers) while slice k is walked, their column ids one slice
//     further ahead; one barrier per slice.  A row's entries are sorted by column, hence by dictionary id: per slice a
//     row contributes one contiguous run of entries, found by a per-row cursor.  Quads that straddle a slice boundary are
//     visited in both slices with the out-of-slice entries masked (value 0, clamped slot); the padding entries at a
//     row's end (value 0, id = the row's own slot) are masked the same way and never stall anything but their own row.
// Summation order per row: entries in storage order within a slice, slices ascending -- fixed, independent of the grid.
constexpr int kDictThreads = 1024;
constexpr int kDictLdsBudget = 160 * 1024 - 2048;   // bytes of LDS a workgroup may carve (160 KiB per CU)
// staged float4 per thread and slice (their registers are live across the walk: what the 128-VGPR budget of a
// 16-wave workgroup leaves)
constexpr int dict_max_u(int nv) { return nv == 1 ? 7 : (nv == 2 ? 5 : 4); }

// acc += v_e * a_e for the four entries of a quad, two floats at a time (v_pk_fma_f32 with the op_sel broadcast of one
// half of the (x, y) / (z, w) value pair; entry order 0, 1, 2, 3)
__device__ __forceinline__ void quad_fma(mgp_v4f& acc, mgp_v4f vv, mgp_v4f a0, mgp_v4f a1, mgp_v4f a2, mgp_v4f a3) {
  const mgp_v2f vlo = mgp_v2f{vv.x, vv.y}, vhi = mgp_v2f{vv.z, vv.w};
  mgp_v2f lo = mgp_v2f{acc.x, acc.y}, hi = mgp_v2f{acc.z, acc.w};
  lo = pk_fma_lo(vlo, mgp_v2f{a0.x, a0.y}, lo); hi = pk_fma_lo(vlo, mgp_v2f{a0.z, a0.w}, hi);
  lo = pk_fma_hi(vlo, mgp_v2f{a1.x, a1.y}, lo); hi = pk_fma_hi(vlo, mgp_v2f{a1.z, a1.w}, hi);
  lo = pk_fma_lo(vhi, mgp_v2f{a2.x, a2.y}, lo); hi = pk_fma_lo(vhi, mgp_v2f{a2.z, a2.w}, hi);
  lo = pk_fma_hi(vhi, mgp_v2f{a3.x, a3.y}, lo); hi = pk_fma_hi(vhi, mgp_v2f{a3.z, a3.w}, hi);
  acc = mgp_v4f{lo.x, lo.y, hi.x, hi.y};
}

template <int NV, bool PRE>
__global__ __launch_bounds__(kDictThreads) void spmm_dict_kernel(SpmmArgs p, TileArgs t, int S, int stream_cap) {
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  constexpr int TR = 64, SLOT4 = 16 * NV;            // float4 per dictionary slot
  constexpr int kDictMaxU = dict_max_u(NV);
  mgp_v4f* __restrict__ dict = reinterpret_cast<mgp_v4f*>(tile_lds);                       // [S + 1][SLOT4]; slot S = zeros
  mgp_v4f* __restrict__ svals = dict + (size_t)(S + 1) * SLOT4;                            // [stream_cap / 4]
  mgp_v4h* __restrict__ slid = reinterpret_cast<mgp_v4h*>(svals + (stream_cap >> 2));      // [stream_cap / 4]
  const int skipv = p.skip ? *p.skip : 0;
  const int tickv = (p.tick && blockIdx.x == 0) ? *p.tick : 0;
  if (skipv) return;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int tid = threadIdx.x, grp = tid >> 4, gl = tid & 15;
  const int C = p.C, C4 = C >> 2;
  const mgp_v4f* __restrict__ X4 = reinterpret_cast<const mgp_v4f*>(p.X);
  const int32_t* __restrict__ rowptr = t.rowptr_t ? t.rowptr_t : p.rowptr;
  const float* __restrict__ vals = t.vals_t ? t.vals_t : p.vals;
  const uint32_t* __restrict__ tile_cols = reinterpret_cast<const uint32_t*>(t.tile_cols);
  const int total4 = S * SLOT4;                       // float4 per staged slice
  // the zero slot: entries outside the staged slice are redirected here (value x 0) instead of being masked one by one
  if (tid < SLOT4) dict[total4 + tid] = mgp_v4f{0.f, 0.f, 0.f, 0.f};
  mgp_v4f dsum[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) dsum[v] = mgp_v4f{0.f, 0.f, 0.f, 0.f};

  // workgroup lb walks the contiguous tile range [lb T / G, (lb + 1) T / G): any grid; one tile per workgroup up to
  // kMaxGrid tiles.  (A persistent grid of one workgroup per CU was measured too: the launch of a 1024-thread workgroup
  // with ~150 KB of LDS costs ~5 us per round of the grid, but a persistent workgroup serialises its tiles' prologues
  // completely: 115 against 90 us at C = 128 on the 60k graph.)
  const int64_t t0 = (int64_t)lb * t.ntiles / gridDim.x;
  const int64_t t1 = ((int64_t)lb + 1) * t.ntiles / gridDim.x;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t r0 = tile * TR;
    const int64_t r1 = r0 + TR < p.n ? r0 + TR : p.n;
    const int dp = t.tile_ptr[tile];
    const int D = t.tile_ptr[tile + 1] - dp;
    const int e0 = rowptr[r0], e1 = rowptr[r1];
    const int nq = (e1 - e0) >> 2;                     // quads of the tile (rows are padded to quads)
    const int nsl = (D + S - 1) / S;
    // ---- this group's row
    const int64_t prow = r0 + grp;
    const bool valid = prow < r1;
    const int64_t pr = valid ? prow : r0;
    const int64_t rr = t.rowid ? (int64_t)t.rowid[pr] : pr;
    const int64_t grr = rr + p.goff;
    int cur = (rowptr[pr] - e0) >> 2;                  // in quads, relative to the tile's stream
    const int end = valid ? (rowptr[pr + 1] - e0) >> 2 : cur;
    // ---- staging helpers: thread -> float4 j = tid + 1024 u of the slice image, slot j / SLOT4, piece j % SLOT4
    unsigned ids[kDictMaxU];
    mgp_v4f xr[kDictMaxU];
    float sc[PRE ? kDictMaxU : 1];
    auto load_ids = [&](int k) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kDictMaxU; ++u) {
        const int j = tid + kDictThreads * u;
        const int slot = k * S + j / SLOT4;
        ids[u] = tile_cols[dp + ((j < total4 && slot < D) ? slot : 0)];
      }
    };
    auto load_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kDictMaxU; ++u) {
        const int j = tid + kDictThreads * u;
        const int f = j % SLOT4;
        xr[u] = X4[(int64_t)ids[u] * C4 + (f < C4 ? f : 0)];
        if (PRE) sc[u] = p.pre[ids[u]];
      }
    };
    // slots past the dictionary's end (last slice) and pieces past the row's end hold zeros
    auto write_rows = [&](int k) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kDictMaxU; ++u) {
        const int j = tid + kDictThreads * u;
        if (j < total4) {
          const bool on = k * S + j / SLOT4 < D && j % SLOT4 < C4;
          mgp_v4f v = xr[u];
          if (PRE) { v.x *= sc[u]; v.y *= sc[u]; v.z *= sc[u]; v.w *= sc[u]; }
          dict[j] = on ? v : mgp_v4f{0.f, 0.f, 0.f, 0.f};
        }
      }
    };
    // ---- prologue: ids of slice 0 -> rows of slice 0 + ids of slice 1 + the matrix stream, all in flight
    if (nsl > 0) load_ids(0);
    const float e_pre = PRE ? p.pre[grr] : 1.f;
    const float e_diag = p.diag[rr];
    const float l_post = p.post ? p.post[grr] : 1.f;
    for (int i = tid; i < nq; i += kDictThreads) {
      svals[i] = *reinterpret_cast<const mgp_v4f*>(vals + (int64_t)e0 + 4 * (int64_t)i);
      slid[i] = *reinterpret_cast<const mgp_v4h*>(t.lid + (int64_t)e0 + 4 * (int64_t)i);
    }
    if (nsl > 0) {
      load_rows();
      write_rows(0);
      if (nsl > 1) load_ids(1);
    }
    __syncthreads();
    mgp_v4f acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = mgp_v4f{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nsl; ++k) {
      if (k + 1 < nsl) {
        load_rows();                                   // X rows of slice k + 1: in flight (registers) during the walk below
        if (k + 2 < nsl) load_ids(k + 2);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- walk this row's run of slice k
      const unsigned lo = (unsigned)(k * S);
      const int Sk = D - k * S < S ? D - k * S : S;
      const mgp_v4f* __restrict__ db = dict + gl;
      while (cur < end) {
        const mgp_v4f vq = svals[cur];
        const mgp_v4h lq = slid[cur];
        const unsigned i0 = (unsigned)lq.x - lo, i1 = (unsigned)lq.y - lo, i2 = (unsigned)lq.z - lo, i3 = (unsigned)lq.w - lo;
        // the first entry of a quad is never padding: a quad that starts behind this slice ends the run
        if ((int)i0 >= Sk) break;
        // outside the slice (already done, still to come, or padding): the zero slot
        const unsigned s0 = min(i0, (unsigned)S), s1 = min(i1, (unsigned)S), s2 = min(i2, (unsigned)S), s3 = min(i3, (unsigned)S);
        const mgp_v4f* __restrict__ b0 = db + s0 * SLOT4;
        const mgp_v4f* __restrict__ b1 = db + s1 * SLOT4;
        const mgp_v4f* __restrict__ b2 = db + s2 * SLOT4;
        const mgp_v4f* __restrict__ b3 = db + s3 * SLOT4;
        // two 256-byte pieces of the four slots at a time (32 VGPRs of operands whatever NV is)
#pragma unroll
        for (int v0 = 0; v0 < NV; v0 += 2) {
          constexpr int kTwo = 2;
          mgp_v4f a0[kTwo], a1[kTwo], a2[kTwo], a3[kTwo];
#pragma unroll
          for (int w = 0; w < kTwo; ++w)
            if (v0 + w < NV) {
              a0[w] = b0[16 * (v0 + w)];
              a1[w] = b1[16 * (v0 + w)];
              a2[w] = b2[16 * (v0 + w)];
              a3[w] = b3[16 * (v0 + w)];
            }
#pragma unroll
          for (int w = 0; w < kTwo; ++w)
            if (v0 + w < NV) quad_fma(acc[v0 + w], vq, a0[w], a1[w], a2[w], a3[w]);
        }
        // entries behind this slice (or padding whose own slot lies there) keep the cursor on this quad
        const int mx = max(max((int)i1, (int)i2), (int)i3);
        if (mx >= Sk) break;
        ++cur;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k + 1 < nsl) {
        __syncthreads();                               // every group is done with slice k: the buffer may be overwritten
        write_rows(k + 1);
      }
      __syncthreads();
    }
    // ---- epilogue of this lane's 4 NV columns
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int f = gl + 16 * v;
      if (valid && f < C4) {
        const mgp_v4f ex = X4[grr * C4 + f];
        const mgp_v4f lb4 = *reinterpret_cast<const mgp_v4f*>((p.base ? p.base : p.X) + grr * C + 4 * f);
        const mgp_v4f ld4 = *reinterpret_cast<const mgp_v4f*>((p.dotw ? p.dotw : p.X) + grr * C + 4 * f);
        const float xs0 = ex.x * e_pre, xs1 = ex.y * e_pre, xs2 = ex.z * e_pre, xs3 = ex.w * e_pre;
        mgp_v4f y;
        y.x = p.co * ((p.a * xs0 + p.b * (e_diag * xs0 - acc[v].x)) * l_post) + (p.base ? p.cb * lb4.x : 0.f);
        y.y = p.co * ((p.a * xs1 + p.b * (e_diag * xs1 - acc[v].y)) * l_post) + (p.base ? p.cb * lb4.y : 0.f);
        y.z = p.co * ((p.a * xs2 + p.b * (e_diag * xs2 - acc[v].z)) * l_post) + (p.base ? p.cb * lb4.z : 0.f);
        y.w = p.co * ((p.a * xs3 + p.b * (e_diag * xs3 - acc[v].w)) * l_post) + (p.base ? p.cb * lb4.w : 0.f);
        *reinterpret_cast<mgp_v4f*>(p.Y + grr * C + 4 * f) = y;
        if (p.dotw) {
          dsum[v].x = fmaf(ld4.x, y.x, dsum[v].x); dsum[v].y = fmaf(ld4.y, y.y, dsum[v].y);
          dsum[v].z = fmaf(ld4.z, y.z, dsum[v].z); dsum[v].w = fmaf(ld4.w, y.w, dsum[v].w);
        }
      }
    }
    // (the next tile's prologue overwrites the stream and the dictionary: every group is past its last read of both at
    // the slice loop's final barrier; a tile without slices has read neither)
  }
  if (p.dot_partials) {
    // red[group][SLOT4] float4: the workgroup's 64 rows (x its tiles) per column, added in group order
    mgp_v4f* red = dict;
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) red[grp * SLOT4 + gl + 16 * v] = dsum[v];
    __syncthreads();
    if (tid < C) {
      const float* rf = reinterpret_cast<const float*>(red);
      float sacc = 0.f;
      for (int g = 0; g < TR; ++g) sacc += rf[g * (SLOT4 * 4) + tid];
      p.dot_partials[(int64_t)lb * C + tid] = sacc;
    }
  }
  if (p.tick && blockIdx.x == 0 && tid == 0) *p.tick = tickv + 1;
}

// ---------------------------------------------------------------- 16 < C <= 256 on the MATRIX CORES: dense 16-row tiles
// The gather kernels above move one X row through the vector memory path per ENTRY (3.6 M x 512 bytes at N = 60k, C = 128:
// 91 us, bound by the L1 miss path); the LDS dictionary kernels move it once per 64-row tile but pay for it in barriers and
// LDS reads per entry (151 us).  Here a tile is 16 rows and is stored DENSE in its own D distinct columns, in the operand layout
// of v_mfma_f32_16x16x4_f32 (mgp_spmm_mt_fill: step s of a tile is 64 floats, lane (i, kq) = A[row i][distinct column
// 4 s + kq]; 21 % of the cells are non-zero on the C3 graph).  A wave owns (tile, 64-column block): per step it loads its
// image dword and, for the step's four distinct columns, 16 bytes of X per lane -- lane (j, kq) takes X[d(4 s + kq)][64 cb +
// 4 j .. + 3], whose four components are the B operands of four MFMAs that produce output columns 64 cb + 4 j + e.  Every
// distinct X row crosses the memory path once per tile (5.3 entries share it), nothing goes through LDS, no barrier, no
// atomics, and the sum of a row is taken in ascending column order whatever the entry order of the CSR.
//  * steps go in BLOCKS of four; a wave keeps three blocks of operands in flight ahead of the one it multiplies (ring of four
//    register sets, loop unrolled by four blocks = one 64-entry batch of the tile's column list): with one block in flight
//    the launch was bound by its longest tile x the memory latency.  All loads are inline asm into fixed registers and the
//    waits are `s_waitcnt vmcnt(N)`, N = the loads younger than the block waited for (a wave's vector memory operations
//    complete in order).  Order of a body's memory operations (body = blocks k .. k + 3, R(i) = the LR loads of block i,
//    D(b) = the column-list batch of blocks 4 b .. 4 b + 3):
//        ... R(k) R(k+1) R(k+2) | D(k/4 + 2) . wait(k) R(k+3) . wait(k+1) R(k+4) . wait(k+2) R(k+5) . wait(k+3) R(k+6) | ...
//    so wait(k), wait(k+1), wait(k+2) leave 2 LR + 1 operations in flight and wait(k+3) 2 LR.  tools/check_kblock_isa.py
//    replays the loop against an in-order queue (CPU test).
//  * the requests run up to 6 blocks and 2 batches past a tile's end into the next tile's (valid) data or the padding behind
//    the last tile; steps past the end are multiplied by zero.  Before the epilogue EVERYTHING in flight is waited for: the
//    compiler believes an asm load's destination is written at the asm statement and reuses the register afterwards -- the
//    first pipelined lab version computed its store addresses in registers a late load then overwrote (a memory fault).
//  * consecutive tiles (locality order: they share X rows) run on ONE XCD, so that its L2 holds its slice of X (76.7 -> 67.4 us).
//  * measured at N = 60k, C = 128 (tools/lab/time_mt.py, Y = L X with the epilogue): 61.7 us against 91.1 for the gather kernel;
//    tried on it and dropped (docs/scope.md, open levers): one-wave workgroups, the long tiles first / at raised priority,
//    non-temporal image loads, both 64-column blocks of a tile in one wave.
struct MtArgs {
  const int32_t* sptr;   // [T + 1] steps before tile t; every tile has a multiple of 4
  const int32_t* dcol;   // [4 * steps + 192] distinct columns per step, padded with a valid column whose image cells are 0
  const float* img;      // [64 * (steps + 32)]
  int T, NCB;
  int img_bytes, dic_bytes;
#ifdef MGP_MT_STAMP      // lab build (tools/lab/stamp_mt.py): per wave {start, loop entry, loop exit, end} wall clocks, blocks, HW_ID, XCC_ID
  unsigned long long* stamps;
#endif
};
#ifdef MGP_MT_STAMP
unsigned long long* g_mt_stamps = nullptr;
#define MT_STAMP(k) do { if (m.stamps && lane == 0) m.stamps[(size_t)w * 8 + (k)] = wall_clock64(); } while (0)
#else
#define MT_STAMP(k) do { } while (0)
#endif

template <bool PRE>
struct MtBuf {
  float a[4];
  mgp_v4f b[4];
  float pr[PRE ? 4 : 1];
};

template <bool PRE>
__device__ __forceinline__ void mt_request(MtBuf<PRE>& nb, int dq, int p0, int kq, int joff, int rowbytes, int so,
                                           __amdgpu_buffer_rsrc_t rimg, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rpre,
                                           int lane4) {
  int off[4], offp[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int d = __builtin_amdgcn_ds_bpermute((4 * (p0 + p) + kq) * 4, dq);
    // (a 24-bit multiply: column ids and row bytes are below 2^24, the product below 2^31.  The plain `d * rowbytes + joff` became
    // v_mad_u64_u32 with a 64-bit addend whose unused high half the compiler took from any register at hand -- in round 5 the
    // destination of an operand load still in flight: harmless, but the stream check rightly flags every touch of such a register)
    off[p] = (int)__umul24((unsigned)d, (unsigned)rowbytes) + joff;
    offp[p] = d * 4;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(nb.a[p]) : "v"(lane4), "s"(rimg), "s"(so), "n"(256 * p));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(nb.b[p]) : "v"(off[p]), "s"(rx));
    if constexpr (PRE) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(nb.pr[p]) : "v"(offp[p]), "s"(rpre));
  }
  // the addresses stay live (= in registers of their own) until the last load of the group has been issued: the compiler
  // believes a load's destination is written AT the asm statement and is free to compute a later address in an earlier
  // load's destination, which the hardware may overwrite first when the issue of the later load stalls
  asm volatile("" :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]));
  if constexpr (PRE) asm volatile("" :: "v"(offp[0]), "v"(offp[1]), "v"(offp[2]), "v"(offp[3]));
}

template <int N, bool PRE>
__device__ __forceinline__ void mt_wait(MtBuf<PRE>& cb) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cb.a[0]) : "n"(N));
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    asm volatile("" : "+v"(cb.a[p]), "+v"(cb.b[p]));
    if constexpr (PRE) asm volatile("" : "+v"(cb.pr[p]));
  }
}

template <bool PRE>
__device__ __forceinline__ void mt_mfma(const MtBuf<PRE>& cb, mgp_v4f (&acc)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float av = cb.a[p];
    if constexpr (PRE) av *= cb.pr[p];           // x[col] * pre[col]: the scale rides on the matrix value
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cb.b[p][e], acc[e], 0, 0, 0);
  }
}

// Round 5, after per-wave stamps of the round-4 kernel (tools/lab/stamp_mt.py, C = 128 on the 60k graph, 66 us): a wave lived
// 9-11 us + 0.5 us per block of four steps -- a 2.2 us prologue (tile offsets, column batches, first requests), the first
// operands a memory latency later, a final drain that also waited for six blocks requested PAST the tile's end, then the
// epilogue's own load round trip -- half the life of a mean tile of 18.6 blocks, 2.4 times per wave slot; while its waves were
// in their loops a SIMD's matrix pipe was ~90 % busy (the launch is 29 us of pipe work).  The same launch with every image / X
// request answered without memory traffic took 63 us (tools/lab/mt_bounds.sh): it was never bound by cache misses.
// What changed:
//   * every tile is a whole number of BODIES (four blocks = 64 distinct columns = one column batch; +7 % steps on the C3 graph),
//     so the unrolled loop needs no masked steps and its LAST body is peeled: nothing is requested past the tile's end (30 % of
//     the round-4 kernel's memory traffic) and the waits of that body count down to zero;
//   * the epilogue's operands (the X block of the tile's own rows, the diagonal) are REQUESTED WITH THE FIRST BLOCKS -- inline-asm
//     loads like the stream's; they depend on nothing the loop computes -- and have landed long before the loop ends.  Extra
//     operations in the queue only make the counted waits conservative (they wait for all but the N youngest).  Operands the
//     common callers do not pass at these widths (pre / post scalings, base, a dot weight other than X) are fetched with
//     ordinary loads in the epilogue: correct, a round trip slower, rare.
// Tried in between and dropped, both measured (docs/kernels/spmm.md, round 5): PERSISTENT waves over contiguous tile ranges of
// equal block count with the operand pipeline carried across tile boundaries (pending accumulators, the epilogue a body later):
// 38 us at C = 64 (one wave per tile: 41), but 77 us at C = 128 and 177 at C = 256 -- ranges of two or three whole tiles differ
// by a tile (max 88 blocks against a mean of 49) -- and 493 us against 375 at 1M nodes / C = 64, where waves far apart in the
// tile order no longer share X rows in L2.  The hardware's dispatch of one wave per tile balances and co-locates better.
template <bool PRE>
__global__ __launch_bounds__(kBlock) void spmm_mt_kernel(SpmmArgs p, MtArgs m) {
  constexpr int LR = PRE ? 12 : 8;      // loads per block request
  if (p.skip && *p.skip) return;
  if (p.tick && blockIdx.x == 0 && threadIdx.x == 0) *p.tick += 1;
  const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
  const int lb = mgp_xcd_block(blockIdx.x, gridDim.x);
  const int w = __builtin_amdgcn_readfirstlane(lb * (kBlock / 64) + (int)(threadIdx.x >> 6));
  const int t = w / m.NCB, cb = w % m.NCB;           // the column blocks of a tile side by side: they share its image
  const int C = p.C;
  if (t >= m.T && !p.dot_partials) return;           // (with dot partials every wave of the workgroup meets at the barrier below)
  mgp_v4f ds = {0.f, 0.f, 0.f, 0.f};                 // this lane's share of sum_rows dotw * y for columns c0 .. c0 + 3
  MT_STAMP(0);
  if (t < m.T) {
  const int blk0 = __builtin_amdgcn_readfirstlane(m.sptr[t]) >> 2, blk1 = __builtin_amdgcn_readfirstlane(m.sptr[t + 1]) >> 2;
  const int64_t nx = p.n + p.goff;                   // rows of X the columns can name (host side: goff == 0)
#if defined(MGP_MT_LAB) && (MGP_MT_LAB & 2)     // lab (tools/lab/mt_bounds.sh): every X request out of range -> zeros, no memory traffic
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), (short)0, 0, 0x00020000);
#else
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X), (short)0, (int)(nx * C * 4), 0x00020000);
#endif
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.img), (short)0, m.img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdic = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(m.dcol), (short)0, m.dic_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rpre = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(PRE ? p.pre : p.X), (short)0, (int)(nx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rdiag = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.diag), (short)0, (int)(p.n * 4), 0x00020000);
#if defined(MGP_MT_LAB) && (MGP_MT_LAB & 4)     // lab: every X request goes to row 0 (always cached)
  const int lane4 = lane * 4, rowbytes = 0, joff = cb * 256 + j * 16;
#else
  const int lane4 = lane * 4, rowbytes = C * 4, joff = cb * 256 + j * 16;
#endif
  const int c0 = 64 * cb + 4 * j;                    // acc[e][r]: row 16 t + 4 kq + r, column c0 + e
  mgp_v4f acc[4], ex[4];
  float ed[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { acc[e] = mgp_v4f{0.f, 0.f, 0.f, 0.f}; ex[e] = mgp_v4f{0.f, 0.f, 0.f, 0.f}; ed[e] = 0.f; }
  MtBuf<PRE> buf0, buf1, buf2, buf3;
  int dq0, dq1, dq2;          // column-list batches of the body in hand, of the next one, and the one in flight
  const int dic0 = blk0 * 64, img0 = blk0 * 1024;      // byte offsets of the tile's first block
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dq0) : "v"(lane4), "s"(rdic), "s"(dic0));
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:256" : "=v"(dq1) : "v"(lane4), "s"(rdic), "s"(dic0));
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq0), "+v"(dq1));
  mt_request<PRE>(buf0, dq0, 0, kq, joff, rowbytes, img0, rimg, rx, rpre, lane4);
  mt_request<PRE>(buf1, dq0, 4, kq, joff, rowbytes, img0 + 1024, rimg, rx, rpre, lane4);
  mt_request<PRE>(buf2, dq0, 8, kq, joff, rowbytes, img0 + 2048, rimg, rx, rpre, lane4);
  {
    // the epilogue's operands: the X block of the tile's own rows (16 bytes per lane and row) and the diagonal
    int offx[4], offd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + 4 * kq + r;
      // (rows past n / columns past C: in range of the descriptor or answered with 0; 24-bit multiply as in mt_request)
      offx[r] = (int)__umul24((unsigned)row, (unsigned)(C * 4)) + c0 * 4;
      offd[r] = row * 4;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(ex[r]) : "v"(offx[r]), "s"(rx));
      asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(ed[r]) : "v"(offd[r]), "s"(rdiag));
    }
    asm volatile("" :: "v"(offx[0]), "v"(offx[1]), "v"(offx[2]), "v"(offx[3]), "v"(offd[0]), "v"(offd[1]), "v"(offd[2]), "v"(offd[3]));
  }
  MT_STAMP(1);
  for (int k = blk0; k < blk1 - 4; k += 4) {
    const int so = k * 1024, sd = k * 64;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:512" : "=v"(dq2) : "v"(lane4), "s"(rdic), "s"(sd));
    mt_wait<2 * LR + 1, PRE>(buf0);
    mt_request<PRE>(buf3, dq0, 12, kq, joff, rowbytes, so + 3 * 1024, rimg, rx, rpre, lane4);
    __builtin_amdgcn_sched_barrier(0);      // requests stay in front of the block's MFMAs (left alone, the scheduler sinks them)
    mt_mfma<PRE>(buf0, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<2 * LR + 1, PRE>(buf1);
    mt_request<PRE>(buf0, dq1, 0, kq, joff, rowbytes, so + 4 * 1024, rimg, rx, rpre, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma<PRE>(buf1, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<2 * LR + 1, PRE>(buf2);
    mt_request<PRE>(buf1, dq1, 4, kq, joff, rowbytes, so + 5 * 1024, rimg, rx, rpre, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma<PRE>(buf2, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<2 * LR, PRE>(buf3);
    mt_request<PRE>(buf2, dq1, 8, kq, joff, rowbytes, so + 6 * 1024, rimg, rx, rpre, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma<PRE>(buf3, acc);
    __builtin_amdgcn_sched_barrier(0);
    dq0 = dq1;
    // the batch requested at the top of this body is older than R(k+3), which wait(k+3) has seen land: 3 LR = R(k+4..k+6)
    // waits for nothing new, it only tells the compiler where dq2 becomes readable
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dq2) : "n"(3 * LR));
    dq1 = dq2;
  }
  // ---- the tile's LAST body, peeled: in flight at the top R(k) R(k+1) R(k+2); behind the one request left, R(k+1) R(k+2) R(k+3)
  {
    const int so = (blk1 - 4) * 1024;
    mt_wait<2 * LR, PRE>(buf0);
    mt_request<PRE>(buf3, dq0, 12, kq, joff, rowbytes, so + 3 * 1024, rimg, rx, rpre, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma<PRE>(buf0, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<2 * LR, PRE>(buf1);
    mt_mfma<PRE>(buf1, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<LR, PRE>(buf2);
    mt_mfma<PRE>(buf2, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<0, PRE>(buf3);
    mt_mfma<PRE>(buf3, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq0), "+v"(dq1));
  mt_wait<0, PRE>(buf0); mt_wait<0, PRE>(buf1); mt_wait<0, PRE>(buf2);
#pragma unroll
  for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(ex[r]), "+v"(ed[r]));
  MT_STAMP(2);
#ifdef MGP_MT_STAMP
  if (m.stamps && lane == 0) {
    m.stamps[(size_t)w * 8 + 4] = (unsigned long long)(blk1 - blk0);
    m.stamps[(size_t)w * 8 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);     // HW_REG_HW_ID
    m.stamps[(size_t)w * 8 + 6] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);     // HW_REG_XCC_ID
  }
#endif
  if (c0 < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = (int64_t)16 * t + 4 * kq + r;
      if (row < p.n) {
        const int64_t gr = row + p.goff;
        mgp_v4f xs = ex[r];
        if (PRE) xs *= p.pre[gr];
        const mgp_v4f av = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        const mgp_v4f lx = ed[r] * xs - av;
        mgp_v4f tt = p.a * xs + p.b * lx;
        if (p.post) tt *= p.post[gr];
        mgp_v4f y = p.co * tt;
        if (p.base) y += p.cb * *reinterpret_cast<const mgp_v4f*>(p.base + gr * C + c0);
        *reinterpret_cast<mgp_v4f*>(p.Y + gr * C + c0) = y;
        if (p.dot_partials) {
          const mgp_v4f dw = (p.dotw == p.X) ? ex[r] : *reinterpret_cast<const mgp_v4f*>(p.dotw + gr * C + c0);
          ds += dw * y;
        }
      }
    }
  }
#ifdef MGP_MT_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MT_STAMP(3);
  }   // t < m.T
  if (p.dot_partials) {
    // per workgroup and column: lanes kq = 1..3 onto kq = 0 (fixed order), then the workgroup's waves of the column's block in
    // wave order.  Any four consecutive waves hold every column block (NCB <= 4), except past the last tile: zeros there.
    __shared__ float red[kBlock / 64][64];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = ds[e];
      v += __shfl_down(v, 16, 64);
      v += __shfl_down(v, 32, 64);
      if (kq == 0) red[wave][4 * j + e] = v;
    }
    __syncthreads();
    const int w0 = lb * (kBlock / 64);
    for (int c = threadIdx.x; c < C; c += kBlock) {
      float tsum = 0.f;
#pragma unroll
      for (int wv = 0; wv < kBlock / 64; ++wv)
        if ((w0 + wv) % m.NCB == c / 64 && (w0 + wv) / m.NCB < m.T) tsum += red[wv][c & 63];
      p.dot_partials[(int64_t)lb * C + c] = tsum;
    }
  }
}

// the image and the padded column list of the tiles: one thread per row scatters its entries
__global__ __launch_bounds__(kBlock) void spmm_mt_fill_kernel(int64_t n, const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ vals, const uint16_t* __restrict__ lid,
                                                               const int32_t* __restrict__ tile_ptr, const int32_t* __restrict__ tile_cols,
                                                               const int32_t* __restrict__ sptr, int32_t* __restrict__ dcol,
                                                               float* __restrict__ img) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // over 16 ceil(n / 16) positions: the last tile may be short
  if (r >= ((n + 15) >> 4 << 4)) return;
  const int64_t t = r >> 4;
  const int i = (int)(r & 15);
  const int64_t s0 = sptr[t];
  if (r < n) {
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const float v = vals[e];
      if (v == 0.f) continue;                       // padding entries
      const int slot = lid[e];
      img[(s0 + (slot >> 2)) * 64 + (slot & 3) * 16 + i] += v;
    }
  }
  // this row's share of the tile's padded column list: entries i, i + 16, ...
  const int d0 = tile_ptr[t], D = tile_ptr[t + 1] - d0;
  const int total = 4 * (sptr[t + 1] - (int)s0);
  for (int q = i; q < total; q += 16) dcol[4 * s0 + q] = D > 0 ? tile_cols[d0 + (q < D ? q : D - 1)] : 0;
}

// In-solve duration of the C = 1 tile kernel, measured live (bench.py `roofline`): between mgp_spmm_timing_begin and
// mgp_spmm_timing_end every launch of spmv_tile_kernel is made with hipExtLaunchKernelGGL and its own start / stop event
// pair -- the events take the dispatch's begin / end timestamps (what rocprofv3 --kernel-trace reports), not the gaps
// between launches.  Eager launches only (the events of a captured launch belong to the graph node): bench.py times a
// plan created with use_graph = 0.
struct SpmvTimer {
  std::vector<hipEvent_t> ev;
  int used = 0;
  bool on = false;
};
SpmvTimer g_spmv_timer;

struct Plan {
  int grid;
  int64_t rows_per_block;
};

Plan make_plan(int64_t n, int rows_per_pass) {
  // contiguous row range per workgroup (a multiple of one pass of its lane groups), grid <= kMaxGrid
  int64_t rpb = rows_per_pass;
  int64_t grid = mgp_cdiv(n, rpb);
  if (grid > kMaxGrid) {
    rpb = mgp_cdiv(mgp_cdiv(n, kMaxGrid), rows_per_pass) * rows_per_pass;
    grid = mgp_cdiv(n, rpb);
  }
  if (grid < 1) grid = 1;
  return Plan{(int)grid, rpb};
}

// C == 1 shape: lanes per row (4 entries per lane per pass) and rows in flight per lane group.
// The host wrapper sets the lanes from the mean padded row length of the graph it built
// (mgp_spmm_set_group_hint); measured best on the 60k and 500k graphs: 8 lanes x 1 row (tools/tune_spmv.py).
std::atomic<int> g_row_group_hint{16};
std::atomic<int> g_rows_in_flight{1};

}  // namespace

// The lab switches below are process-wide atomics; a call reads them ONCE, at its entry point (knobs_snap), into a thread-local
// snapshot that every helper of the call consults: a switch flipped by another thread in the middle of a call cannot make the
// call's shape tests disagree with each other.  (Two CALLS -- sizing the dot partials, then launching -- can still see different
// settings: the header says the switches are not to be flipped while another thread is inside this family of calls.)
struct Knobs { int hint, rif, tile, tile_small, tile_wide, v4, dict, mt; };
thread_local Knobs tl_knobs{16, 1, 1, 1, 1, 1, 1, 1};
static void knobs_snap();

extern "C" int mgp_spmm_set_group_hint(int lanes) {
  if (lanes != 4 && lanes != 8 && lanes != 16 && lanes != 32 && lanes != 64) return MGP_ERR_ARG;
  g_row_group_hint = lanes;
  return MGP_OK;
}

extern "C" int mgp_spmm_set_rows_in_flight(int rows) {
  if (rows != 1 && rows != 2 && rows != 4 && rows != 8) return MGP_ERR_ARG;
  g_rows_in_flight = rows;
  return MGP_OK;
}

static int spmv_rows_in_flight() {
  int r = tl_knobs.rif;
  if (r > tl_knobs.hint) r = tl_knobs.hint;   // lane t finishes row t: needs R <= G
  return r;
}

static int spmm_cols_group(int C) {
  int g = 4;
  while (g < C && g < 64) g <<= 1;
  return g;
}

// rows a workgroup covers per pass (the grid / dot-partial count follows from it).  C in {4,8,12,16}
// uses 64 whichever kernel runs (the row16 kernel needs 16-byte aligned X, the generic one does not)
static int spmm_rows_per_pass(int C) {
  if (C == 1) return (kBlock / tl_knobs.hint) * spmv_rows_in_flight();
  if (C <= 16 && (C & 3) == 0) return (kBlock / 16) * 4;
  return kBlock / spmm_cols_group(C) * 4;
}

#ifdef MGP_STAMP
std::atomic<int> g_stamp_enable{0};
extern "C" int mgp_stamp_enable(int on) { g_stamp_enable = on ? 1 : 0; return 0; }
#endif
std::atomic<int> g_tile_mode{1};

extern "C" int mgp_spmm_set_tile_mode(int on) {
  g_tile_mode = on ? 1 : 0;
  return MGP_OK;
}

int mgp_tile_plan(const mgp_csr_t* L, int C, int* grid, int* tiles_per_block, size_t* lds_bytes);

static size_t tile_lds_bytes(const mgp_csr_t* L) {
  return ((size_t)L->tile_max_cols + (size_t)(L->tile_max_entries >> 2)) * sizeof(float);
}

static bool use_tiles(const mgp_csr_t* L, int C) {
  const int ord = (L->tile_rowptr != nullptr) + (L->tile_vals != nullptr) + (L->tile_rowid != nullptr);
  if (ord != 0 && ord != 3) return false;      // an ordered tile view is all three arrays or none
  return C == 1 && tl_knobs.tile && L->lid && L->tile_ptr && L->tile_cols &&
         (L->tile_rows == 32 || L->tile_rows == 64 || L->tile_rows == 128) && tile_lds_bytes(L) <= 65536 - 64;
}

static int tile_grid(const mgp_csr_t* L, int* tiles_per_block) {
  const int64_t ntiles = mgp_cdiv(L->n, L->tile_rows);
  const int64_t tpb = mgp_cdiv(ntiles, kMaxGrid);
  if (tiles_per_block) *tiles_per_block = (int)tpb;
  return (int)mgp_cdiv(ntiles, tpb);
}

int mgp_tile_plan(const mgp_csr_t* L, int C, int* grid, int* tiles_per_block, size_t* lds_bytes) {
  knobs_snap();
  if (!L || !use_tiles(L, C)) return 0;
  const int g = tile_grid(L, tiles_per_block);
  if (grid) *grid = g;
  if (lds_bytes) *lds_bytes = tile_lds_bytes(L);
  return 1;
}

// C in {4, 8, 12, 16} on 64-row tiles whose staged data (dictionary rows + matrix stream) fits the LDS budget
std::atomic<int> g_tile_small_mode{1};
// quads of partial sums staged at once: what is left of a quarter of the CU's LDS (160 KB, four workgroups) behind the
// dictionary, in steps of 256, at least 1024 (the quads a workgroup holds in registers), at most the largest tile
static int tile_small_window(const mgp_csr_t* L) {
  const int max_q = L->tile_max_entries >> 2;
  const long budget = 40448 - (long)L->tile_max_cols * 16;
  long w = budget > 0 ? budget / 16 / 256 * 256 : 0;
  if (w < 1024) w = 1024;
  if (w > max_q) w = (max_q + 255) / 256 * 256;
  return (int)(w > 0 ? w : 256);
}
static size_t tile_small_lds_bytes(const mgp_csr_t* L, int C) {
  const size_t max_q = (size_t)(L->tile_max_entries >> 2), w = (size_t)tile_small_window(L);
  size_t b = ((size_t)L->tile_max_cols + (max_q < w ? max_q : w)) * 16;
  const size_t red = (size_t)64 * C * sizeof(float);        // dot-partial staging reuses the same LDS
  return b > red ? b : red;
}
static bool use_tiles_small(const mgp_csr_t* L, int C) {
  const int ord = (L->tile_rowptr != nullptr) + (L->tile_vals != nullptr) + (L->tile_rowid != nullptr);
  if (ord != 0 && ord != 3) return false;
  if (!(C == 4 || C == 8 || C == 12 || C == 16) || !tl_knobs.tile || !tl_knobs.tile_small) return false;
  if (!L->lid || !L->tile_ptr || !L->tile_cols || L->tile_rows != 64) return false;
  if ((L->tile_max_entries & 3) != 0) return false;
  return tile_small_lds_bytes(L, C) <= 65536 - 64;
}

// 16 < C <= 256, C % 4 == 0 on 64-row tiles: the wide tile kernel (mgp_spmm_set_tile_wide_mode(0): spmm_kernel)
std::atomic<int> g_tile_wide_mode{1};
std::atomic<int> g_spmm_v4_mode{1};     // float4-lane gather kernel for 16 < C <= 256 (mgp_spmm_set_v4_mode(0): spmm_kernel)
static int tile_wide_cap(const mgp_csr_t* L) {
  int cap = (L->tile_max_cols + 63) / 64 * 64;
  if (cap > kWideCap) cap = kWideCap;
  if (cap < 64) cap = 64;
  return cap;
}
static size_t tile_wide_lds_bytes(const mgp_csr_t* L) {
  const size_t b = (size_t)tile_wide_cap(L) * 64;
  return b > 4096 ? b : 4096;                                // (dot-partial staging: 64 x 16 floats)
}
static bool use_tiles_wide(const mgp_csr_t* L, int C) {
  const int ord = (L->tile_rowptr != nullptr) + (L->tile_vals != nullptr) + (L->tile_rowid != nullptr);
  if (ord != 0 && ord != 3) return false;
  if (C <= 16 || C > 256 || (C & 3) != 0 || !tl_knobs.tile || !tl_knobs.tile_wide) return false;
  if (!L->lid || !L->tile_ptr || !L->tile_cols || L->tile_rows != 64) return false;
  if ((L->tile_max_entries & 3) != 0) return false;
  // measured (tools/lab/time_spmm_wide.py): on the 60k graph the per-entry gather kernel reads its X rows out of L2 and
  // wins from 64 columns up (63 vs 84 us at C = 64, 91 vs 158 us at C = 128; 47 vs 59 us at C = 32 the other way, but see below); on
  // the 1M graph, whose X block does not fit the caches, the dictionary kernel is 1.5-2.4x faster at every width
  // (1.48 vs 3.04 ms at C = 128).  mode 2 forces it at any size (tests, A/B).
  // (and up to 64 columns the float4-lane gather kernel beats both on a cache-resident X block: 33 us at C = 32)
  if (tl_knobs.tile_wide == 2) return true;
  return (size_t)L->n * (size_t)C * sizeof(float) >= ((size_t)96 << 20);
}

extern "C" int mgp_spmm_set_v4_mode(int on) {
  g_spmm_v4_mode = on == 2 ? 2 : (on ? 1 : 0);
  return MGP_OK;
}

extern "C" int mgp_spmm_set_tile_wide_mode(int on) {
  g_tile_wide_mode = on == 2 ? 2 : (on ? 1 : 0);
  return MGP_OK;
}

extern "C" int mgp_spmm_set_tile_small_mode(int on) {
  g_tile_small_mode = on ? 1 : 0;
  return MGP_OK;
}

// 16 < C <= 256 on 64-row tiles with lanes over columns (spmm_dict_kernel): slots per slice from what the stream leaves
// of the CU's LDS.  mgp_spmm_set_dict_mode: 0 never, 1 (default) where it was measured to win, 2 wherever the shape allows.
// Measured (tools/lab/time_spmm_wide.py, round 3; us: this kernel | chunked dictionary kernel | float4 gather | per-column
// gather):  60k graph  C = 64: 52 | 84 | 54 | 63   C = 128: 90 | 160 | 101 | 91   C = 256: 176 | 319 | 204 | 186
//           1M graph   C = 64: 704 | 777 | 1020 | 1188   C = 128: 1201 | 1458 | 2084 | 3037   C = 256: 2380 | 2871 | 6515 | 6828
// -> taken from 64 columns up when the X block (n x C floats) is 96 MB or more, i.e. does not sit in the caches.
std::atomic<int> g_dict_mode{1};
static int dict_nv(int C) { return (C + 63) / 64; }
static int dict_stream_cap(const mgp_csr_t* L) { return (L->tile_max_entries + 7) / 8 * 8; }
static int dict_slots(const mgp_csr_t* L, int C) {
  const int nv = dict_nv(C);
  const long left = (long)kDictLdsBudget - (long)dict_stream_cap(L) * 6 - (long)nv * 256;      // (the zero slot)
  long s = left / (nv * 256L);
  const long cap = (long)dict_max_u(nv) * kDictThreads / (16 * nv);      // staged float4 per thread <= dict_max_u
  if (s > cap) s = cap;
  s = s / 16 * 16;
  // no point in slices larger than the largest dictionary
  const long need = ((long)L->tile_max_cols + 15) / 16 * 16;
  if (s > need) s = need;
  return (int)s;
}
static size_t dict_lds_bytes(const mgp_csr_t* L, int C) {
  const size_t d = (size_t)(dict_slots(L, C) + 1) * dict_nv(C) * 256 + (size_t)dict_stream_cap(L) * 6;
  const size_t red = (size_t)64 * dict_nv(C) * 256;
  return d > red ? d : red;
}
static int dict_grid(const mgp_csr_t* L) {
  const int64_t ntiles = mgp_cdiv(L->n, L->tile_rows);
  return (int)(ntiles < kMaxGrid ? ntiles : kMaxGrid);
}
static bool dict_shape_ok(const mgp_csr_t* L, int C) {
  const int ord = (L->tile_rowptr != nullptr) + (L->tile_vals != nullptr) + (L->tile_rowid != nullptr);
  if (ord != 0 && ord != 3) return false;
  if (C <= 16 || C > 256 || (C & 3) != 0 || !tl_knobs.tile || !tl_knobs.dict) return false;
  if (!L->lid || !L->tile_ptr || !L->tile_cols || L->tile_rows != 64) return false;
  if ((L->tile_max_entries & 3) != 0) return false;
  if (dict_slots(L, C) < 32) return false;
  if (tl_knobs.dict == 2) return true;
  return C >= 64 && (size_t)L->n * (size_t)C * sizeof(float) >= ((size_t)96 << 20);
}
// the kernel moves 16 bytes per lane: X, Y, base and dotw rows must be 16-byte aligned (C % 4 == 0 makes every row so
// once the block is); the planned kernel (dot-partial blocks) and the launched one must never disagree, so a
// misaligned operand is an argument error at launch instead of a silent fall-through to another kernel
static bool aligned16(const void* a, const void* b, const void* c, const void* d) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

// 48 <= C <= 256 on the matrix cores (spmm_mt_kernel): taken when the CSR carries the dense 16-row tile image (mgp_spmm_mt_fill;
// the host wrapper builds it for graphs in natural row order whose tiles are at least 1/8 full) and the call has no row offset
// (weighted dot-product partials included: one row of partials per workgroup).  mgp_spmm_set_mt_mode(0) = never (A/B runs, tests).
std::atomic<int> g_mt_mode{1};
static void knobs_snap() {
  tl_knobs = Knobs{g_row_group_hint.load(), g_rows_in_flight.load(), g_tile_mode.load(), g_tile_small_mode.load(), g_tile_wide_mode.load(),
                   g_spmm_v4_mode.load(), g_dict_mode.load(), g_mt_mode.load()};
}
constexpr int kMtMinCols = 48;
static bool mt_shape_ok(const mgp_csr_t* L, int C) {
  if (!tl_knobs.mt || !L->mt_img || !L->mt_sptr || !L->mt_dcol || L->mt_tiles <= 0 || L->mt_steps <= 0) return false;
  // below 48 columns most lanes of a wave's 64-column block idle: the gather kernel is faster there (C = 32: 34 us against 41)
  if (C < kMtMinCols || C > 256 || (C & 3) != 0 || L->tile_rowid) return false;
  if (L->ncols != 0 && L->ncols != L->n) return false;        // a row slice of a partitioned operator never carries an image
  if ((int64_t)L->n * C * 4 >= (int64_t(1) << 31) || ((int64_t)L->mt_steps + 32) * 256 >= (int64_t(1) << 31)) return false;
  if ((L->mt_steps & 15) != 0) return false;           // every tile a whole number of bodies of four blocks
  return L->mt_tiles == (int32_t)mgp_cdiv(L->n, 16);
}

#ifdef MGP_MT_STAMP
extern "C" int mgp_mt_set_stamp_buffer(void* buf) {
  g_mt_stamps = static_cast<unsigned long long*>(buf);
  return MGP_OK;
}
#endif
extern "C" int mgp_spmm_set_mt_mode(int on) {
  const int prev = g_mt_mode;
  g_mt_mode = on ? 1 : 0;
  return prev;
}

extern "C" int mgp_spmm_mt_fill(int64_t n, const int32_t* rowptr, const float* vals, const uint16_t* lid16,
                                const int32_t* tile_ptr16, const int32_t* tile_cols16, const int32_t* sptr, int64_t steps,
                                int32_t* dcol, float* img, void* stream) {
  if (n <= 0 || !rowptr || !vals || !lid16 || !tile_ptr16 || !tile_cols16 || !sptr || !dcol || !img || steps <= 0 || (steps & 15))
    return MGP_ERR_ARG;
  hipStream_t st = mgp_stream(stream);
  MGP_HIP_TRY(hipMemsetAsync(img, 0, (size_t)(steps + 32) * 256, st));
  MGP_HIP_TRY(hipMemsetAsync(dcol + 4 * steps, 0, 192 * sizeof(int32_t), st));
  hipLaunchKernelGGL(spmm_mt_fill_kernel, dim3((unsigned)mgp_cdiv(n + 15, kBlock)), dim3(kBlock), 0, st, n, rowptr, vals, lid16,
                     tile_ptr16, tile_cols16, sptr, dcol, img);
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

extern "C" int mgp_spmm_set_dict_mode(int on) {
  g_dict_mode = on == 2 ? 2 : (on ? 1 : 0);
  return MGP_OK;
}

int mgp_spmm_dot_blocks_for(const mgp_csr_t* L, int C) {
  if (!L) return MGP_ERR_ARG;
  knobs_snap();
  if (use_tiles(L, C) || use_tiles_small(L, C)) return tile_grid(L, nullptr);
  if (mt_shape_ok(L, C)) return (int)mgp_cdiv((int64_t)L->mt_tiles * ((C + 63) / 64), kBlock / 64);
  if (dict_shape_ok(L, C)) return dict_grid(L);
  if (use_tiles_wide(L, C)) return tile_grid(L, nullptr);
  return mgp_spmm_dot_blocks(L->n, C);
}

// which kernel a call of mgp_spmm_fused with this CSR / width would launch (tests, docs): 0 = gather (C == 1: row groups),
// 1 = C == 1 tile kernel, 2 = small-C tile kernel, 3 = matrix-core tiles, 5 = lanes-over-columns dictionary, 6 = chunked
// dictionary (4 was round 4's persistent 8-lanes-per-row dictionary kernel: measured slower, removed in round 5)
extern "C" int mgp_spmm_kernel_choice(const mgp_csr_t* L, int C, int with_dot, int64_t row_offset) {
  if (!L || L->n <= 0 || C <= 0 || C > 256) return MGP_ERR_ARG;
  knobs_snap();
  static const float one = 1.f;
  if (use_tiles(L, C)) return 1;
  if (use_tiles_small(L, C)) return 2;
  (void)one;
  if (mt_shape_ok(L, C) && row_offset == 0) return 3;
  if (mt_shape_ok(L, C) && with_dot) return MGP_ERR_UNSUPPORTED;      // (as mgp_spmm_fused_rows: partials were sized for kernel 3)
  if (dict_shape_ok(L, C)) return 5;
  if (use_tiles_wide(L, C)) return 6;
  return 0;
}

extern "C" int mgp_spmm_dot_blocks_csr(const mgp_csr_t* L, int C) {
  if (!L || L->n <= 0 || C <= 0) return MGP_ERR_ARG;
  return mgp_spmm_dot_blocks_for(L, C);
}

extern "C" int mgp_spmm_dot_blocks(int64_t n, int C) {
  if (n <= 0 || C <= 0) return MGP_ERR_ARG;
  knobs_snap();
  return make_plan(n, spmm_rows_per_pass(C)).grid;
}

template <int G, int R, bool PRE>
static void launch_spmv(const SpmmArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((spmv_kernel<G, R, PRE>), dim3(grid), dim3(kBlock), 0, st, a);
}

template <int G, bool PRE>
static int launch_spmv_r(const SpmmArgs& a, int R, int grid, hipStream_t st) {
  switch (R) {
    case 1: launch_spmv<G, 1, PRE>(a, grid, st); return MGP_OK;
    case 2: launch_spmv<G, 2, PRE>(a, grid, st); return MGP_OK;
    case 4: launch_spmv<G, 4, PRE>(a, grid, st); return MGP_OK;
    case 8: if (G >= 8) { launch_spmv<G, (G >= 8 ? 8 : 4), PRE>(a, grid, st); return MGP_OK; }
  }
  return MGP_ERR_ARG;
}

template <int G, int NACC, bool PRE>
static void launch_spmm(const SpmmArgs& a, int grid, hipStream_t st) {
  hipLaunchKernelGGL((spmm_kernel<G, NACC, PRE>), dim3(grid), dim3(kBlock), 0, st, a);
}

extern "C" int mgp_spmm_fused(const mgp_csr_t* L, const float* X, int C, float* Y, float a, float b,
                              const float* pre, const float* post, const float* base, float cb,
                              float co, const float* dotw, float* dot_partials, void* stream) {
  return mgp_spmm_fused_ex(L, X, C, Y, a, b, pre, post, base, cb, co, dotw, dot_partials, nullptr, nullptr,
                           stream);
}

int mgp_spmm_fused_ex(const mgp_csr_t* L, const float* X, int C, float* Y, float a, float b,
                      const float* pre, const float* post, const float* base, float cb, float co,
                      const float* dotw, float* dot_partials, const int* skip, int* tick, void* stream) {
  return mgp_spmm_fused_part(L, 0, X, C, Y, a, b, pre, post, base, cb, co, dotw, dot_partials, skip, tick, stream);
}

int mgp_spmm_fused_part(const mgp_csr_t* L, int64_t row_offset, const float* X, int C, float* Y, float a, float b,
                        const float* pre, const float* post, const float* base, float cb, float co,
                        const float* dotw, float* dot_partials, const int* skip, int* tick, void* stream) {
  return mgp_spmm_fused_first(L, row_offset, X, C, Y, a, b, pre, post, base, cb, co, dotw, dot_partials, skip, tick,
                              nullptr, stream);
}

namespace {
// what a tile-kernel launch was made with: kept by the CG plan for the one graph node whose input pointer
// changes from solve to solve (mgp_spmm_patch_node)
struct TileLaunchRecord {
  SpmmArgs p;
  TileArgs t;
};
static_assert(sizeof(TileLaunchRecord) <= MGP_SPMM_RECORD_BYTES, "MGP_SPMM_RECORD_BYTES too small");
}  // namespace

// Re-point a captured tile-SpMV node at another input: every pointer operand of the recorded launch that equals
// `old_ptr` (X, base, dotw) becomes `new_ptr` in the executable graph; the record keeps the original.
int mgp_spmm_patch_node(void* exec, void* node, const void* record, const float* old_ptr, const float* new_ptr) {
  if (!exec || !node || !record) return MGP_ERR_ARG;
  TileLaunchRecord rec;
  memcpy(&rec, record, sizeof(rec));
  if (rec.p.X == old_ptr) rec.p.X = new_ptr;
  if (rec.p.base == old_ptr) rec.p.base = new_ptr;
  if (rec.p.dotw == old_ptr) rec.p.dotw = new_ptr;
  hipKernelNodeParams np;
  memset(&np, 0, sizeof(np));
  MGP_HIP_TRY(hipGraphKernelNodeGetParams(static_cast<hipGraphNode_t>(node), &np));
  void* kp[2] = {(void*)&rec.p, (void*)&rec.t};
  np.kernelParams = kp;
  np.extra = nullptr;
  MGP_HIP_TRY(hipGraphExecKernelNodeSetParams(static_cast<hipGraphExec_t>(exec), static_cast<hipGraphNode_t>(node), &np));
  return MGP_OK;
}

int mgp_spmm_fused_first(const mgp_csr_t* L, int64_t row_offset, const float* X, int C, float* Y, float a, float b,
                         const float* pre, const float* post, const float* base, float cb, float co,
                         const float* dotw, float* dot_partials, const int* skip, int* tick,
                         const MgpFirst* first, void* stream) {
  if (!L || !L->rowptr || !L->col || !L->vals || !L->diag || !X || !Y) return MGP_ERR_ARG;
  if (L->n <= 0 || C <= 0 || C > 256) return C > 256 ? MGP_ERR_UNSUPPORTED : MGP_ERR_ARG;
  if (X == Y) return MGP_ERR_ARG;  // rows gather other rows of X: never in place
  knobs_snap();
  hipStream_t st = mgp_stream(stream);
  SpmmArgs p{L->n, L->rowptr, L->col, L->vals, L->diag, X, Y, C, a, b, pre, post, base, cb, co,
             dotw, dotw ? dot_partials : nullptr, 0, skip, tick, row_offset, nullptr, nullptr, 0};
#ifdef MGP_STAMP
  p.stamp_on = g_stamp_enable;
#endif
  if (first) {
    if (!use_tiles(L, C)) return MGP_ERR_UNSUPPORTED;
    p.copy_x = first->copy_x;
    p.dot2_partials = (dotw && dot_partials) ? first->dot2_partials : nullptr;
    p.tick_reset = first->tick_reset;
  }
  if (use_tiles(L, C)) {
    TileArgs ta{L->tile_ptr, L->tile_cols, L->lid, mgp_cdiv(L->n, L->tile_rows), 1, L->tile_max_cols,
                L->tile_rowptr, L->tile_vals, L->tile_rowid, L->tile_max_entries, 0};
    const int grid = tile_grid(L, &ta.tiles_per_block);
    const size_t lds = tile_lds_bytes(L);
#define MGP_TILE_LAUNCH_K(KERNEL, BS)                                                                   \
  do {                                                                                                   \
    if (g_spmv_timer.on && 2 * g_spmv_timer.used + 1 < (int)g_spmv_timer.ev.size()) {                    \
      hipEvent_t e0 = g_spmv_timer.ev[2 * g_spmv_timer.used], e1 = g_spmv_timer.ev[2 * g_spmv_timer.used + 1]; \
      ++g_spmv_timer.used;                                                                               \
      hipExtLaunchKernelGGL(KERNEL, dim3(grid), dim3(BS), lds, st, e0, e1, 0, p, ta);                    \
    } else hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(BS), lds, st, p, ta);                             \
  } while (0)
#define MGP_TILE_LAUNCH(BS)                                                                              \
  do {                                                                                                   \
    if (pre) MGP_TILE_LAUNCH_K((spmv_tile_kernel<true, BS>), BS);                                        \
    else MGP_TILE_LAUNCH_K((spmv_tile_kernel<false, BS>), BS);                                           \
  } while (0)
    if (first && first->record) {
      TileLaunchRecord rec{p, ta};
      memcpy(first->record, &rec, sizeof(rec));
    }
    if (L->tile_rows == 32) MGP_TILE_LAUNCH(128);
    else if (L->tile_rows == 64) MGP_TILE_LAUNCH(256);
    else MGP_TILE_LAUNCH(512);
#undef MGP_TILE_LAUNCH_K
#undef MGP_TILE_LAUNCH
  } else if (use_tiles_small(L, C)) {
    if (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(base) |
          reinterpret_cast<uintptr_t>(dotw)) & 15) != 0)
      return MGP_ERR_ARG;                  // [n, C] blocks with C a multiple of 4 are 16-byte aligned row by row
    TileArgs ta{L->tile_ptr, L->tile_cols, L->lid, mgp_cdiv(L->n, L->tile_rows), 1, L->tile_max_cols,
                L->tile_rowptr, L->tile_vals, L->tile_rowid, L->tile_max_entries, tile_small_window(L)};
    const int grid = tile_grid(L, &ta.tiles_per_block);
    const size_t lds = tile_small_lds_bytes(L, C);
#define MGP_TILE_SMALL_LAUNCH(C4)                                                                                 \
  do {                                                                                                            \
    if (pre) hipLaunchKernelGGL((spmm_tile_q_kernel<C4, true>), dim3(grid), dim3(256), lds, st, p, ta);        \
    else hipLaunchKernelGGL((spmm_tile_q_kernel<C4, false>), dim3(grid), dim3(256), lds, st, p, ta);           \
  } while (0)
    if (C == 4) MGP_TILE_SMALL_LAUNCH(1);
    else if (C == 8) MGP_TILE_SMALL_LAUNCH(2);
    else if (C == 12) MGP_TILE_SMALL_LAUNCH(3);
    else MGP_TILE_SMALL_LAUNCH(4);
#undef MGP_TILE_SMALL_LAUNCH
  } else if (mt_shape_ok(L, C) && row_offset != 0 && dotw && dot_partials) {
    // a CSR that carries the tile image, a row offset AND dot partials: mgp_spmm_dot_blocks_csr sized them for the matrix-core
    // kernel, which takes no row offset -- the one combination that is refused (mgp_spmm_kernel_choice reports it too)
    return MGP_ERR_UNSUPPORTED;
  } else if (mt_shape_ok(L, C) && row_offset == 0) {
    if (!aligned16(X, Y, base, dotw)) return MGP_ERR_ARG;
    MtArgs ma{L->mt_sptr, L->mt_dcol, L->mt_img, L->mt_tiles, (C + 63) / 64, (int)(((int64_t)L->mt_steps + 32) * 256),
              (int)(((int64_t)L->mt_steps * 4 + 192) * 4)};
    const int grid = (int)mgp_cdiv((int64_t)ma.T * ma.NCB, kBlock / 64);
#if defined(MGP_MT_LAB) && (MGP_MT_LAB & 1)     // lab: every image request out of range -> zeros, no memory traffic
    ma.img_bytes = 0;
#endif
#ifdef MGP_MT_STAMP
    ma.stamps = g_mt_stamps;
#endif
    if (pre) hipLaunchKernelGGL((spmm_mt_kernel<true>), dim3(grid), dim3(kBlock), 0, st, p, ma);
    else hipLaunchKernelGGL((spmm_mt_kernel<false>), dim3(grid), dim3(kBlock), 0, st, p, ma);
  } else if (dict_shape_ok(L, C)) {
    if (!aligned16(X, Y, base, dotw)) return MGP_ERR_ARG;   // (the plan counted this kernel's dot-partial blocks)
    TileArgs ta{L->tile_ptr, L->tile_cols, L->lid, mgp_cdiv(L->n, L->tile_rows), 1, L->tile_max_cols,
                L->tile_rowptr, L->tile_vals, L->tile_rowid, L->tile_max_entries, 0};
    const int grid = dict_grid(L);
    const size_t lds = dict_lds_bytes(L, C);
    const int S = dict_slots(L, C), cap = dict_stream_cap(L);
#define MGP_DICT_LAUNCH(NV)                                                                                         \
  do {                                                                                                              \
    /* the attribute is per DEVICE and the call is cheap next to a launch: set every time (a process that drives a  \
       second GPU, or two host threads, must not depend on a process-wide flag) */                                  \
    if (pre) {                                                                                                      \
      MGP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_dict_kernel<NV, true>),                   \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kDictLdsBudget));                 \
      hipLaunchKernelGGL((spmm_dict_kernel<NV, true>), dim3(grid), dim3(kDictThreads), lds, st, p, ta, S, cap);     \
    } else {                                                                                                        \
      MGP_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_dict_kernel<NV, false>),                  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kDictLdsBudget));                 \
      hipLaunchKernelGGL((spmm_dict_kernel<NV, false>), dim3(grid), dim3(kDictThreads), lds, st, p, ta, S, cap);    \
    }                                                                                                               \
  } while (0)
    switch (dict_nv(C)) {
      case 1: MGP_DICT_LAUNCH(1); break;
      case 2: MGP_DICT_LAUNCH(2); break;
      case 3: MGP_DICT_LAUNCH(3); break;
      default: MGP_DICT_LAUNCH(4); break;
    }
#undef MGP_DICT_LAUNCH
  } else if (use_tiles_wide(L, C)) {
    if (!aligned16(X, Y, base, dotw)) return MGP_ERR_ARG;   // planned and launched kernel must not disagree
    TileArgs ta{L->tile_ptr, L->tile_cols, L->lid, mgp_cdiv(L->n, L->tile_rows), 1, L->tile_max_cols,
                L->tile_rowptr, L->tile_vals, L->tile_rowid, L->tile_max_entries, 0};
    const int grid = tile_grid(L, &ta.tiles_per_block);
    const size_t lds = tile_wide_lds_bytes(L);
    const int cap = tile_wide_cap(L);
    if (pre) hipLaunchKernelGGL((spmm_tile_wide_kernel<true>), dim3(grid), dim3(256), lds, st, p, ta, cap);
    else hipLaunchKernelGGL((spmm_tile_wide_kernel<false>), dim3(grid), dim3(256), lds, st, p, ta, cap);
  } else if (C == 1) {
    const int G = tl_knobs.hint;
    const int R = spmv_rows_in_flight();
    Plan pl = make_plan(L->n, (kBlock / G) * R);
    p.rows_per_block = pl.rows_per_block;
    int rc = MGP_OK;
#define MGP_SPMV_CASE(GG)                                                                     \
  case GG:                                                                                    \
    rc = pre ? launch_spmv_r<GG, true>(p, R, pl.grid, st) : launch_spmv_r<GG, false>(p, R, pl.grid, st); \
    break;
    switch (G) {
      MGP_SPMV_CASE(4)
      MGP_SPMV_CASE(8)
      MGP_SPMV_CASE(16)
      MGP_SPMV_CASE(32)
      MGP_SPMV_CASE(64)
      default: return MGP_ERR_ARG;
    }
#undef MGP_SPMV_CASE
    MGP_TRY(rc);
  } else if (C <= 16 && (C & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15) == 0) {
    Plan pl = make_plan(L->n, spmm_rows_per_pass(C));
    p.rows_per_block = pl.rows_per_block;
#define MGP_ROW16_LAUNCH(C4)                                                                                   \
  do {                                                                                                         \
    if (pre) hipLaunchKernelGGL((spmm_row16_kernel<C4, true>), dim3(pl.grid), dim3(kBlock), 0, st, p);          \
    else hipLaunchKernelGGL((spmm_row16_kernel<C4, false>), dim3(pl.grid), dim3(kBlock), 0, st, p);             \
  } while (0)
    if (C == 4) MGP_ROW16_LAUNCH(1);
    else if (C == 8) MGP_ROW16_LAUNCH(2);
    else if (C == 12) MGP_ROW16_LAUNCH(3);
    else MGP_ROW16_LAUNCH(4);
#undef MGP_ROW16_LAUNCH
  } else if (tl_knobs.v4 && C > 16 && C <= 256 && (C & 3) == 0 &&
             // measured (tools/lab/time_spmm_wide.py): 33 vs 59 us at C = 32 and 54 vs 63 us at C = 64 on the 60k graph, but
             // 101 vs 91 us at C = 128 (the X block no longer sits in L2 and the per-column kernel's whole-line pieces use
             // the Infinity Cache path better); on the 1M graph, when the dictionaries are not there, it wins at every width
             (tl_knobs.v4 == 2 || C <= 64 || (size_t)L->n * (size_t)C * sizeof(float) >= ((size_t)96 << 20)) &&
             ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(base) |
               reinterpret_cast<uintptr_t>(dotw) | reinterpret_cast<uintptr_t>(L->col) | reinterpret_cast<uintptr_t>(L->vals)) & 15) == 0 &&
             (L->tile_max_entries & 3) == 0 && L->lid != nullptr) {
    // (quad-padded rows are what the tile builder guarantees: the CSR of a graph without dictionaries keeps the
    // per-column kernel below)
    Plan pl = make_plan(L->n, spmm_rows_per_pass(C));        // the same row ranges / dot-partial blocks as spmm_kernel
    p.rows_per_block = pl.rows_per_block;
#define MGP_V4_LAUNCH(LPR)                                                                          \
  do {                                                                                              \
    if (pre) hipLaunchKernelGGL((spmm_v4_kernel<LPR, true>), dim3(pl.grid), dim3(kBlock), 0, st, p);  \
    else hipLaunchKernelGGL((spmm_v4_kernel<LPR, false>), dim3(pl.grid), dim3(kBlock), 0, st, p);     \
  } while (0)
    if (C <= 32) MGP_V4_LAUNCH(8);
    else if (C <= 64) MGP_V4_LAUNCH(16);
    else if (C <= 128) MGP_V4_LAUNCH(32);
    else MGP_V4_LAUNCH(64);
#undef MGP_V4_LAUNCH
  } else {
    const int G = spmm_cols_group(C);
    const int nacc = (int)mgp_cdiv(C, G);
    Plan pl = make_plan(L->n, spmm_rows_per_pass(C));
    p.rows_per_block = pl.rows_per_block;
#define MGP_SPMM_LAUNCH(GG, NA)                                    \
  do {                                                             \
    if (pre) launch_spmm<GG, NA, true>(p, pl.grid, st);            \
    else launch_spmm<GG, NA, false>(p, pl.grid, st);               \
  } while (0)
    if (G == 4) MGP_SPMM_LAUNCH(4, 1);
    else if (G == 8) MGP_SPMM_LAUNCH(8, 1);
    else if (G == 16) MGP_SPMM_LAUNCH(16, 1);
    else if (G == 32) MGP_SPMM_LAUNCH(32, 1);
    else if (nacc == 1) MGP_SPMM_LAUNCH(64, 1);
    else if (nacc == 2) MGP_SPMM_LAUNCH(64, 2);
    else MGP_SPMM_LAUNCH(64, 4);
#undef MGP_SPMM_LAUNCH
  }
  MGP_LAUNCH_CHECK();
  return MGP_OK;
}

// elementwise y = s[i] * x[i,:]
__global__ void scale_rows_kernel(const float* __restrict__ s, const float* __restrict__ x,
                                  float* __restrict__ y, int64_t n, int C) {
  int64_t total = n * C;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = s[i / C] * x[i];
}

extern "C" int mgp_laplacian_matmul(const mgp_csr_t* L, const float* dsqrt, const float* dinvsqrt,
                                    int mode, const float* X, int C, float* Y, float* work_nc,
                                    void* stream) {
  (void)work_nc;
  if (mode < 0 || mode > 2) return MGP_ERR_ARG;
  if (mode != 0 && (!dsqrt || !dinvsqrt)) return MGP_ERR_ARG;
  // graph_laplacian_operator.py:111-122: vec = rhs * D^{1/2} (rw) or rhs / D^{1/2} (rw^T);
  // out = L_sym vec; out *= D^{-1/2} (rw) or D^{1/2} (rw^T)
  const float* pre = mode == 0 ? nullptr : (mode == 1 ? dsqrt : dinvsqrt);
  const float* post = mode == 0 ? nullptr : (mode == 1 ? dinvsqrt : dsqrt);
  return mgp_spmm_fused(L, X, C, Y, 0.f, 1.f, pre, post, nullptr, 0.f, 1.f, nullptr, nullptr, stream);
}

// Row-partitioned form of mgp_spmm_fused: L_local holds rows [row_offset, row_offset + L_local->n) of
// the operator (column ids global), every vector has the global length; only the local rows of Y are
// written.  The multi-GPU path gathers the slices with RCCL (operator.hip); callers with their own
// exchange layer can use this entry directly.
extern "C" int mgp_spmm_fused_rows(const mgp_csr_t* L_local, int64_t row_offset, const float* X, int C, float* Y,
                                   float a, float b, const float* pre, const float* post, const float* base,
                                   float cb, float co, const float* dotw, float* dot_partials, void* stream) {
  if (row_offset < 0) return MGP_ERR_ARG;
  return mgp_spmm_fused_part(L_local, row_offset, X, C, Y, a, b, pre, post, base, cb, co, dotw, dot_partials, nullptr,
                             nullptr, stream);
}

// Measurement helper (bench.py / tools): `reps` back-to-back launches of Y = L X enqueued from C as ONE
// hipGraph (captured on a private stream, replayed on `stream`), so that neither Python nor the
// eager launch path (~2.7 us per launch, host-bound) paces the kernels -- the same way the CG
// iteration graph issues them.  Falls back to eager launches if the capture fails.
// elapsed_ms (nullable): HIP-event time of the `reps` launches on `stream` (graph build excluded).
extern "C" int mgp_spmm_timing_begin(int max_launches) {
  if (max_launches <= 0 || max_launches > 65536 || g_spmv_timer.on) return MGP_ERR_ARG;
  g_spmv_timer.ev.assign((size_t)2 * max_launches, nullptr);
  for (auto& e : g_spmv_timer.ev) MGP_HIP_TRY(hipEventCreate(&e));
  g_spmv_timer.used = 0;
  g_spmv_timer.on = true;
  return MGP_OK;
}

extern "C" int mgp_spmm_timing_end(float* total_ms, int* launches) {
  if (!g_spmv_timer.on) return MGP_ERR_ARG;
  g_spmv_timer.on = false;
  double sum = 0.0;
  int rc = MGP_OK;
  for (int i = 0; i < g_spmv_timer.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(g_spmv_timer.ev[2 * i + 1]) != hipSuccess ||
        hipEventElapsedTime(&ms, g_spmv_timer.ev[2 * i], g_spmv_timer.ev[2 * i + 1]) != hipSuccess)
      rc = MGP_ERR_ARG;
    sum += ms;
  }
  if (total_ms) *total_ms = (float)sum;
  if (launches) *launches = g_spmv_timer.used;
  for (auto e : g_spmv_timer.ev) (void)hipEventDestroy(e);
  g_spmv_timer.ev.clear();
  g_spmv_timer.used = 0;
  return rc;
}

extern "C" int mgp_spmm_repeat(const mgp_csr_t* L, const float* X, int C, float* Y, int reps, float* elapsed_ms,
                               void* stream) {
  hipStream_t cap = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  bool ok = reps >= 4 && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess;
  if (ok) ok = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    int rc = MGP_OK;
    for (int i = 0; i < reps && rc == MGP_OK; ++i)
      rc = mgp_spmm_fused_ex(L, X, C, Y, 0.f, 1.f, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr, nullptr, nullptr,
                             nullptr, cap);
    ok = hipStreamEndCapture(cap, &graph) == hipSuccess && rc == MGP_OK && graph != nullptr;
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
  }
  (void)hipGetLastError();
  int rc = MGP_OK;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (elapsed_ms) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, mgp_stream(stream)); }
  if (ok) {
    rc = (int)hipGraphLaunch(exec, mgp_stream(stream));
  } else {
    for (int i = 0; i < reps && rc == MGP_OK; ++i)
      rc = mgp_spmm_fused_ex(L, X, C, Y, 0.f, 1.f, nullptr, nullptr, nullptr, 0.f, 1.f, nullptr, nullptr, nullptr,
                             nullptr, stream);
  }
  if (elapsed_ms) {
    (void)hipEventRecord(e1, mgp_stream(stream));
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(elapsed_ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  } else if (rc == MGP_OK) {
    rc = (int)hipStreamSynchronize(mgp_stream(stream));
  }
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (cap) (void)hipStreamDestroy(cap);
  return rc;
}
