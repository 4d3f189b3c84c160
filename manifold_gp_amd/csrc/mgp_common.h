// Shared helpers for libmgp_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "mgp_hip.h"

#define MGP_WAVE 64
#define MGP_NXCD 8

#define MGP_HIP_TRY(expr)                                  \
  do {                                                     \
    hipError_t _e = (expr);                                \
    if (_e != hipSuccess) return (int)_e;                  \
  } while (0)

#define MGP_TRY(expr)                                      \
  do {                                                     \
    int _r = (expr);                                       \
    if (_r != 0) return _r;                                \
  } while (0)

#define MGP_LAUNCH_CHECK() MGP_HIP_TRY(hipGetLastError())

static inline hipStream_t mgp_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t mgp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t mgp_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct MgpArena {
  char* base;
  size_t cap;
  size_t off;
  MgpArena(void* p, size_t bytes) : base(static_cast<char*>(p)), cap(bytes), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = mgp_align(count * sizeof(T));
    if (off + bytes > cap) { off = cap + 1; return nullptr; }
    T* r = reinterpret_cast<T*>(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= cap; }
};

// XCD-aware logical block id: physical blocks are dealt round-robin over the 8 XCDs, so
// blocks b and b+8 share an L2.  Map them to CONTIGUOUS logical ids so that one XCD streams a
// contiguous row range (its slice of x / y stays in its own L2).  Bijective for any grid size.
__device__ __forceinline__ int mgp_xcd_block(int pb, int grid) {
  int per = grid / MGP_NXCD, rem = grid % MGP_NXCD;
  int x = pb % MGP_NXCD, i = pb / MGP_NXCD;
  // XCD x owns per (+1 if x < rem) logical blocks, laid out consecutively
  int start = x * per + (x < rem ? x : rem);
  return start + i;
}

__device__ __forceinline__ float mgp_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int G>
__device__ __forceinline__ float mgp_group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ double mgp_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
