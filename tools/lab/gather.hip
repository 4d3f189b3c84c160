// Lab: cost of a 4-byte gather wave-instruction vs address pattern (not product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

template <int MODE>
__device__ __forceinline__ float gload(const float* p) {
  float v;
  if (MODE == 0) return *p;
  if (MODE == 1) return __builtin_nontemporal_load(p);
  if (MODE == 2) { asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
  if (MODE == 3) { asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
  return 0.f;
}

typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i make_rsrc(const void* p, unsigned bytes) {
  unsigned long long a = (unsigned long long)p;
  v4i r;
  r.x = (int)(a & 0xffffffffu);
  r.y = (int)((a >> 32) & 0xffffu);
  r.z = (int)bytes;
  r.w = 0x00020000;      // DATA_FORMAT = 32 (raw buffer)
  return r;
}

// gathers through the buffer path with a cache-policy immediate (aux): 0 plain, 1 sc0, 2 nt, 16 sc1, 17 sc0+sc1
template <int NG, int AUX>
__global__ __launch_bounds__(256) void gather_buf_kernel(const int* __restrict__ idx, const float* __restrict__ x, float* __restrict__ out, long n, int nx) {
  long i = (long)blockIdx.x * 256 * NG + threadIdx.x;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, nx * 4, 0x00020000);
  float acc = 0.f;
  int c[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) c[g] = idx[i + g * 256];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, c[g] * 4, 0, AUX));
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NG, int MODE = 0>
__global__ __launch_bounds__(256) void gather_kernel(const int* __restrict__ idx, const float* __restrict__ x, float* __restrict__ out, long n) {
  long i = (long)blockIdx.x * 256 * NG + threadIdx.x;
  float acc = 0.f;
  int c[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) c[g] = idx[i + g * 256];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc += gload<MODE>(x + c[g]);
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
float time_it(F f, int reps, hipStream_t st) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 10; ++i) f();
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
  const int NX = 60000;          // x vector (240 KB)
  const long NE = 3670016;       // gathers per launch (= padded nnz of the bench graph), multiple of 256*16
  std::vector<int> h(NE);
  float *x, *out; int* idx;
  (void)hipMalloc(&x, NX * 4); (void)hipMemset(x, 0, NX * 4);
  (void)hipMalloc(&out, NE * 4); (void)hipMalloc(&idx, NE * 4);
  std::mt19937 rng(1);
  const char* names[] = {"uniform random over x", "random within +-64 of a per-wave base", "4 adjacent lanes share a 16-B slot (sorted, gap 1)",
                         "16 adjacent lanes share a 64-B line", "lane-consecutive (fully coalesced)", "all lanes of a wave the same address",
                         "stride 16 floats (64 lanes -> 64 distinct lines, sequential)", "same-100-block random (61%) + uniform (39%)",
                         "same-100-block (61%) + one of 8 fixed other blocks per row block (39%)"};
  for (int pat = 0; pat < 9; ++pat) {
    for (long e = 0; e < NE; ++e) {
      long wave = e / 64; int lane = e % 64;
      int base = (int)((wave * 97) % (NX - 2048)) + 1024;
      int v;
      switch (pat) {
        case 0: v = rng() % NX; break;
        case 1: v = base + (int)(rng() % 128) - 64; break;
        case 2: v = base + (lane / 4) * 16 + (lane % 4); break;
        case 3: v = base + (lane / 16) * 64 + (lane % 16); break;
        case 4: v = base + lane; break;
        case 5: v = base; break;
        case 6: v = (base + lane * 16) % NX; break;
        case 7: v = (rng() % 100 < 61) ? (base / 100) * 100 + (int)(rng() % 100) : (int)(rng() % NX); break;
        default: {
          long row = e / 61; int B = (int)((row / 100) % 600);
          if (rng() % 100 < 61) v = B * 100 + (int)(rng() % 100);
          else { unsigned hb = (unsigned)(B * 2654435761u + (rng() % 8) * 40503u); v = (int)(hb % 600) * 100 + (int)(rng() % 100); }
          break;
        }
      }
      h[e] = v;
    }
    (void)hipMemcpy(idx, h.data(), NE * 4, hipMemcpyHostToDevice);
    float t1 = time_it([&] { hipLaunchKernelGGL((gather_kernel<1>), dim3(NE / 256), dim3(256), 0, st, idx, x, out, NE); }, 50, st);
    float t4 = time_it([&] { hipLaunchKernelGGL((gather_kernel<4>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE); }, 50, st);
    float t16 = time_it([&] { hipLaunchKernelGGL((gather_kernel<16>), dim3(NE / 4096), dim3(256), 0, st, idx, x, out, NE); }, 50, st);
    float tnt = time_it([&] { hipLaunchKernelGGL((gather_kernel<4, 1>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE); }, 50, st);
    float b0 = time_it([&] { hipLaunchKernelGGL((gather_buf_kernel<4, 0>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE, NX); }, 50, st);
    float b1 = time_it([&] { hipLaunchKernelGGL((gather_buf_kernel<4, 1>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE, NX); }, 50, st);
    float b16 = time_it([&] { hipLaunchKernelGGL((gather_buf_kernel<4, 16>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE, NX); }, 50, st);
    float b17 = time_it([&] { hipLaunchKernelGGL((gather_buf_kernel<4, 17>), dim3(NE / 1024), dim3(256), 0, st, idx, x, out, NE, NX); }, 50, st);
    printf("%-62s NG=4 %.2f | nt %.2f | buffer plain %.2f sc0 %.2f sc1 %.2f sc0sc1 %.2f us\n", names[pat], t4, tnt, b0, b1, b16, b17);
    (void)t1; (void)t16;
  }
  return 0;
}
