// Lab: cost of a software grid barrier across all co-resident workgroups on MI355X (not part of the product).
// hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned nblocks, unsigned& target, int* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    target += nblocks;
    __threadfence();
    atomicAdd(counter, 1u);
    long spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 20000000) { *fail = 1; break; }     // never hang the box
    }
  }
  __syncthreads();
  return true;
}

// every phase: each block writes one value per thread, barrier, reads a value written by ANOTHER block
__global__ __launch_bounds__(256) void k_bar(unsigned* counter, int* fail, float* buf, int rounds, int with_data) {
  unsigned target = 0;
  const int nb = gridDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (with_data) buf[(size_t)blockIdx.x * 256 + threadIdx.x] = (float)(r + blockIdx.x);
    grid_barrier(counter, nb, target, fail);
    if (with_data) {
      const int ob = (blockIdx.x + 97) % nb;
      acc += __builtin_nontemporal_load(&buf[(size_t)ob * 256 + threadIdx.x]);
    }
  }
  if (with_data && acc == -1.f) buf[0] = acc;
}

int main(int argc, char** argv) {
  int blocks = argc > 1 ? atoi(argv[1]) : 938;
  unsigned* counter; int* fail; float* buf;
  CK(hipMalloc(&counter, 256)); CK(hipMalloc(&fail, 256)); CK(hipMalloc(&buf, (size_t)4096 * 256 * 4));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bar, 256, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("occupancy %d blocks/CU x %d CUs = %d co-resident; launching %d\n", occ, prop.multiProcessorCount, occ * prop.multiProcessorCount, blocks);
  if (blocks > occ * prop.multiProcessorCount) { printf("would not be co-resident: abort\n"); return 1; }
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int with_data = 0; with_data < 2; ++with_data)
    for (int rounds : {1, 101, 401}) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(counter, 0, 256)); CK(hipMemset(fail, 0, 256));
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_bar, dim3(blocks), dim3(256), 0, 0, counter, fail, buf, rounds, with_data);
        hipEventRecord(b, 0); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
      }
      int hf = 0; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
      printf("data %d rounds %4d: %.2f us total, fail %d\n", with_data, rounds, best * 1e3, hf);
    }
  return 0;
}
