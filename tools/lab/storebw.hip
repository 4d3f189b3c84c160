// Lab (not product): rate at which 128 x 128 fp32 tiles of a row-major [n1, n2] matrix can be WRITTEN from registers, by store
// pattern -- what kernel_block's epilogue is bounded by.  One workgroup of 4 waves per tile, each wave a 64 x 64 quarter.
//   0  dword per lane, one instruction = 2 rows x 128 B (the MFMA 32x32 C layout as it falls)          [nt]
//   1  same, default cache policy
//   2  16 B per lane, one instruction = 32 rows x 32 B (operands swapped, straight from the registers)   [nt]
//   3  same, default cache policy
//   4  16 B per lane, one instruction = 4 rows x 256 B (after a transpose through LDS)                  [nt]
//   5  same, default cache policy
//   6  8 B per lane, one instruction = 2 rows x 256 B                                                   [nt]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ K, long n1, long n2, int nct) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const long row0 = (long)(blockIdx.x / nct) * 128 + wr * 64, col0 = (long)(blockIdx.x % nct) * 128 + wc * 64;
  const float v = (float)blockIdx.x;
  if (MODE == 0 || MODE == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = col0 + j * 32 + (lane & 31);
          if (row < n1 && col < n2) {
            if (MODE == 0) __builtin_nontemporal_store(v, K + row * n2 + col);
            else K[row * n2 + col] = v;
          }
        }
  } else if (MODE == 2 || MODE == 3) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const long row = row0 + i * 32 + (lane & 31), col = col0 + j * 32 + 8 * g + 4 * (lane >> 5);
          f4 w = {v, v, v, v};
          if (row < n1 && col < n2) {
            if (MODE == 2) __builtin_nontemporal_store(w, (f4*)(K + row * n2 + col));
            else *(f4*)(K + row * n2 + col) = w;
          }
        }
  } else if (MODE == 4 || MODE == 5) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long row = row0 + q * 4 + (lane >> 4), col = col0 + 4 * (lane & 15);
      f4 w = {v, v, v, v};
      if (row < n1 && col < n2) {
        if (MODE == 4) __builtin_nontemporal_store(w, (f4*)(K + row * n2 + col));
        else *(f4*)(K + row * n2 + col) = w;
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const long row = row0 + q * 2 + (lane >> 5), col = col0 + 2 * (lane & 31);
      f2 w = {v, v};
      if (row < n1 && col < n2) __builtin_nontemporal_store(w, (f2*)(K + row * n2 + col));
    }
  }
}

template <int MODE>
float run(float* K, long n1, long n2) {
  const int nct = (int)((n2 + 127) / 128), nrt = (int)((n1 + 127) / 128);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nct * nrt), dim3(256), 0, 0, K, n1, n2, nct);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nct * nrt), dim3(256), 0, 0, K, n1, n2, nct);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 20 * 1e3f;
}

int main() {
  const long shapes[3][2] = {{600, 60000}, {4096, 60000}, {8192, 8192}};
  for (auto& sh : shapes) {
    const long n1 = sh[0], n2 = sh[1];
    float* K; hipMalloc(&K, n1 * n2 * 4);
    const float us[7] = {run<0>(K, n1, n2), run<1>(K, n1, n2), run<2>(K, n1, n2), run<3>(K, n1, n2), run<4>(K, n1, n2), run<5>(K, n1, n2), run<6>(K, n1, n2)};
    printf("%ld x %ld (%.0f MB):", n1, n2, n1 * n2 * 4 / 1e6);
    for (int m = 0; m < 7; ++m) printf("  mode %d: %.1f us = %.2f TB/s", m, us[m], n1 * n2 * 4 / us[m] / 1e6);
    printf("\n");
    hipFree(K);
  }
  return 0;
}
