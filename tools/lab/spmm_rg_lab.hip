// Lab prototype: SpMM over ROW GROUPS.  R consecutive rows share one stream of their DISTINCT columns
// (batches of 8 columns, R values per column, zero where a row does not hold the column): one X-row
// load feeds R rows' accumulators.  A wave owns a group; the stream is wave-uniform (scalar loads).
// Y = diag .* X - W X  (what mgp_spmm_repeat times).  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xcd_block(int pb, int grid) {
  const int per = grid / 8, rem = grid % 8;
  const int x = pb % 8, i = pb / 8;
  return x * per + (x < rem ? x : rem) + i;
}

typedef int v2i __attribute__((ext_vector_type(2)));

// X-row pieces through a buffer descriptor: voffset = the lane's column pair, soffset = column id x row bytes (one
// s_mul per column instead of a 64-bit address); inline asm, so the waits are placed by hand (vmcnt retires in order)
template <int R, int U>
__device__ __forceinline__ void rg_issue(v2f (&x)[U], float& vl, __amdgpu_buffer_rsrc_t rs, int voff, const int* __restrict__ cb, int rowbytes,
                                         const float* vp) {
  asm volatile("global_load_dword %0, %1, off" : "=v"(vl) : "v"(vp));
  int so[U];
#pragma unroll
  for (int u = 0; u < U; ++u) so[u] = cb[u] * rowbytes;
#pragma unroll
  for (int u = 0; u < U; ++u) asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(x[u]) : "v"(voff), "s"(rs), "s"(so[u]));
}
__device__ __forceinline__ void rg_wait8(v2f (&x)[8], float& vl) {
  asm volatile("s_waitcnt vmcnt(9)" : "+v"(vl), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void rg_wait0(v2f (&x)[8], float& vl) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(vl), "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}

template <int R, int U>
__device__ __forceinline__ void rg_fma(v2f (&acc)[R], const v2f (&x)[U], float vl) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vl), (u * R + r) & 63));
      acc[r].x = fmaf(v, x[u].x, acc[r].x);
      acc[r].y = fmaf(v, x[u].y, acc[r].y);
    }
  }
}

template <int R, int U>
__global__ __launch_bounds__(256) void rg_kernel(const int* __restrict__ rg_ptr, const int* __restrict__ rg_cols,
                                                 const float* __restrict__ rg_vals, const float* __restrict__ diag,
                                                 const float* __restrict__ X, float* __restrict__ Y, int n, int C,
                                                 int ngroups) {
  const int lb = xcd_block(blockIdx.x, gridDim.x);
  const int w = __builtin_amdgcn_readfirstlane(lb * 4 + (threadIdx.x >> 6));
  if (w >= ngroups) return;
  const int lane = threadIdx.x & 63;
  const int cl = 2 * lane < C ? 2 * lane : 0;
  const int b0 = rg_ptr[w], b1 = rg_ptr[w + 1];
  v2f acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = v2f{0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), (short)0, (int)((size_t)n * C * 4), 0x00020000);
  const int voff = cl * 4, rowbytes = C * 4;
  v2f xa[U], xb[U];
  float va = 0.f, vb = 0.f;
  static_assert(U * R <= 64, "one value per lane and batch");
  const float* vlane = rg_vals + (lane < U * R ? lane : 0);
  if (b0 < b1) rg_issue<R, U>(xa, va, rs, voff, rg_cols + (size_t)b0 * U, rowbytes, vlane + (size_t)b0 * U * R);
  for (int b = b0; b < b1; b += 2) {
    const bool two = b + 1 < b1;
    const int bb = two ? b + 1 : b;
    rg_issue<R, U>(xb, vb, rs, voff, rg_cols + (size_t)bb * U, rowbytes, vlane + (size_t)bb * U * R);
    rg_wait8(xa, va);
    rg_fma<R, U>(acc, xa, va);
    const int bn = b + 2 < b1 ? b + 2 : b;
    rg_issue<R, U>(xa, va, rs, voff, rg_cols + (size_t)bn * U, rowbytes, vlane + (size_t)bn * U * R);
    rg_wait8(xb, vb);
    if (two) rg_fma<R, U>(acc, xb, vb);
  }
  rg_wait0(xa, va);
  if (2 * lane < C) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = (int64_t)w * R + r;
      if (row < n) {
        const v2f xs = *reinterpret_cast<const v2f*>(X + row * C + 2 * lane);
        const float d = diag[row];
        v2f y;
        y.x = d * xs.x - acc[r].x;
        y.y = d * xs.y - acc[r].y;
        *reinterpret_cast<v2f*>(Y + row * C + 2 * lane) = y;
      }
    }
  }
}

extern "C" int lab_spmm_rg(int R, const int* rg_ptr, const int* rg_cols, const float* rg_vals, const float* diag,
                           const float* X, float* Y, int n, int C, int reps, float* ms, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int ngroups = (n + R - 1) / R;
  const int grid = (ngroups + 3) / 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) {
    if (R == 4) hipLaunchKernelGGL((rg_kernel<4, 8>), dim3(grid), dim3(256), 0, st, rg_ptr, rg_cols, rg_vals, diag, X, Y, n, C, ngroups);
    else if (R == 8) hipLaunchKernelGGL((rg_kernel<8, 8>), dim3(grid), dim3(256), 0, st, rg_ptr, rg_cols, rg_vals, diag, X, Y, n, C, ngroups);
    else return -1;
  }
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  hipEventElapsedTime(ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return (int)hipGetLastError();
}
