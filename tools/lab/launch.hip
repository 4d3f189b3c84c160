// Lab: what a kernel boundary inside a hipGraph costs on MI355X, and what kernel arguments add to it
// (not part of the product).  hipcc --offload-arch=gfx950 -O3 -o launch launch.hip && ./launch
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Big { float* p[31]; int n; float* last; };   // 264 bytes

__global__ void k_noarg() {}
__global__ void k_store(float* out) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = 1.f; }
__global__ void k_store_all(float* out) { out[blockIdx.x * blockDim.x + threadIdx.x] = 1.f; }
__global__ void k_big_last(Big b) { b.last[blockIdx.x * blockDim.x + threadIdx.x] = (float)b.n; }
__global__ void k_big_all(Big b) {
  float s = 0.f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int k = 0; k < 31; ++k) s += b.p[k][i];
  b.last[i] = s;
}
__global__ void k_flat8(float* a0, float* a1, float* a2, float* a3, float* a4, float* a5, float* a6, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = a0[i] + a1[i] + a2[i] + a3[i] + a4[i] + a5[i] + a6[i];
}
__global__ void k_flag_then(const int* flag, float* a0, float* out) {   // dependent: flag load gates the rest
  if (*flag) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = a0[i];
}

template <typename F>
float graph_time(F enqueue, int reps, hipStream_t st) {
  hipStream_t cap; hipStreamCreateWithFlags(&cap, hipStreamNonBlocking);
  hipGraph_t g; hipGraphExec_t ex;
  hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < reps; ++i) enqueue(cap);
  hipStreamEndCapture(cap, &g);
  hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  hipGraphLaunch(ex, st); hipStreamSynchronize(st);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a, st); hipGraphLaunch(ex, st); hipEventRecord(b, st); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  hipGraphExecDestroy(ex); hipGraphDestroy(g); hipStreamDestroy(cap);
  return best / reps * 1e3f;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int N = 4096 * 256;
  float* buf[33];
  for (int i = 0; i < 33; ++i) { CK(hipMalloc(&buf[i], N * sizeof(float))); CK(hipMemset(buf[i], 0, N * sizeof(float))); }
  int* flag; CK(hipMalloc(&flag, 4)); CK(hipMemset(flag, 0, 4));
  Big b; for (int k = 0; k < 31; ++k) b.p[k] = buf[k]; b.n = 3; b.last = buf[32];
  for (int grid : {1, 256, 938, 4096}) {
    printf("grid=%d x 256\n", grid);
    printf("  no args, empty           %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_noarg, dim3(grid), dim3(256), 0, s); }, 200, st));
    printf("  1 ptr, one lane stores   %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, s, buf[0]); }, 200, st));
    printf("  1 ptr, all lanes store   %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_store_all, dim3(grid), dim3(256), 0, s, buf[0]); }, 200, st));
    printf("  264-B struct, last field %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_big_last, dim3(grid), dim3(256), 0, s, b); }, 200, st));
    printf("  264-B struct, 31 loads   %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_big_all, dim3(grid), dim3(256), 0, s, b); }, 200, st));
    printf("  8 flat ptrs, 7 loads     %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_flat8, dim3(grid), dim3(256), 0, s, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], buf[7]); }, 200, st));
    printf("  flag gate + 1 load       %.2f us\n", graph_time([&](hipStream_t s) { hipLaunchKernelGGL(k_flag_then, dim3(grid), dim3(256), 0, s, flag, buf[0], buf[1]); }, 200, st));
  }
  return 0;
}
