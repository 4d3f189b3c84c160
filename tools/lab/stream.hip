// Lab: read bandwidth of a cache-resident buffer re-read by consecutive launches (not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ int xcd_block(int pb, int grid) {
  int per = grid / 8, rem = grid % 8, x = pb % 8, i = pb / 8;
  return x * per + (x < rem ? x : rem) + i;
}

// each block reads a contiguous chunk; UNROLL float4 loads in flight per thread
template <int UNROLL, bool XCD>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, float* __restrict__ out, long n4, long per_block) {
  const int lb = XCD ? xcd_block(blockIdx.x, gridDim.x) : blockIdx.x;
  long b0 = lb * per_block, b1 = b0 + per_block;
  if (b1 > n4) b1 = n4;
  float acc = 0.f;
  for (long i = b0 + threadIdx.x; i < b1; i += 256 * UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      long j = i + u * 256;
      v[u] = j < b1 ? in[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
float time_it(F f, int reps, hipStream_t st) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(b, st);
  (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  hipStream_t st; (void)hipStreamCreate(&st);
  float* out; (void)hipMalloc(&out, 1 << 22);
  for (double mb : {3.5, 7.0, 14.0, 22.0, 29.4, 60.0, 120.0, 480.0}) {
    long n4 = (long)(mb * 1e6 / 16);
    float4* buf; (void)hipMalloc(&buf, n4 * 16);
    (void)hipMemset(buf, 0, n4 * 16);
    for (int grid : {256, 1024, 2048}) {
      long per = (n4 + grid - 1) / grid;
      float t1 = time_it([&] { hipLaunchKernelGGL((read_kernel<1, false>), dim3(grid), dim3(256), 0, st, buf, out, n4, per); }, 100, st);
      float t4 = time_it([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(256), 0, st, buf, out, n4, per); }, 100, st);
      float t4x = time_it([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(grid), dim3(256), 0, st, buf, out, n4, per); }, 100, st);
      float t8x = time_it([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, st, buf, out, n4, per); }, 100, st);
      printf("%6.1f MB grid %4d: u1 %.2f us (%.0f GB/s) | u4 %.2f (%.0f) | u4+xcd %.2f (%.0f) | u8+xcd %.2f (%.0f)\n", mb, grid,
             t1, mb * 1e3 / t1, t4, mb * 1e3 / t4, t4x, mb * 1e3 / t4x, t8x, mb * 1e3 / t8x);
    }
    (void)hipFree(buf);
  }
  return 0;
}
