// Lab: fixed cost of short dependent-load kernels on MI355X (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -o latency latency.hip && ./latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int DEPTH>
__global__ void chain_kernel(const int* __restrict__ idx, float* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int j = i;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) j = idx[j];
  out[i] = (float)j;
}

__global__ void empty_kernel(float* out) { if (out == nullptr) out[0] = 1.f; }

template <typename F>
float time_it(F f, int reps, hipStream_t st) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  hipStreamSynchronize(st);
  hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3f;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int n : {60000, 240000}) {
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 13) % n);   // permutation-ish, scattered
    int* d_idx; float* d_out;
    CK(hipMalloc(&d_idx, n * sizeof(int))); CK(hipMalloc(&d_out, n * sizeof(float)));
    CK(hipMemcpy(d_idx, h.data(), n * sizeof(int), hipMemcpyHostToDevice));
    int grid = (n + 255) / 256;
    printf("n=%d grid=%d\n", n, grid);
    printf("  empty            %.2f us\n", time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(256), 0, st, d_out); }, 200, st));
    printf("  store only       %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<0>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    printf("  1 dependent load %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<1>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    printf("  2 dependent      %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<2>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    printf("  3 dependent      %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<3>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    printf("  4 dependent      %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<4>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    printf("  8 dependent      %.2f us\n", time_it([&] { hipLaunchKernelGGL(chain_kernel<8>, dim3(grid), dim3(256), 0, st, d_idx, d_out, n); }, 200, st));
    hipFree(d_idx); hipFree(d_out);
  }
  return 0;
}
