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

// Wait for a stream that carries RCCL collectives of a job with more than one rank: hipStreamSynchronize never returns
// when a peer rank has died or a collective hangs.  Polls hipStreamQuery (sleeping 50 us between polls after the first
// millisecond) for at most MGP_DIST_TIMEOUT_S seconds (default 300), then returns MGP_ERR_TIMEOUT: the process is expected
// to exit non-zero (a fresh launch of the job, never a re-exec).  Single-rank callers keep hipStreamSynchronize.
#include <chrono>
#include <cstdlib>
#include <thread>
static inline int mgp_stream_wait_bounded(hipStream_t st) {
  static const double limit_s = [] {
    const char* e = getenv("MGP_DIST_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 300.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return MGP_OK;
    if (q != hipErrorNotReady) return (int)q;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt > limit_s) return MGP_ERR_TIMEOUT;
    if (dt > 1e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
#define MGP_STREAM_WAIT(st, multi_rank)                                   \
  do {                                                                    \
    if (multi_rank) { MGP_TRY(mgp_stream_wait_bounded(st)); }             \
    else { MGP_HIP_TRY(hipStreamSynchronize(st)); }                       \
  } while (0)
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

// Cross-lane sums on the VALU (DPP) instead of __shfl_xor, which hipcc lowers to ds_bpermute_b32: a
// 6-level xor tree is 6 dependent LDS-pipe round trips (~0.1 us each in a latency-bound kernel).
//   row_shr:1,2,4,8 build 16-lane row totals in lane 15 of every row, row_bcast15 / row_bcast31 carry
//   them across the four rows, lane 63 ends up with the wave total, readlane broadcasts it.
// Fixed order, every lane gets the same value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mgp_dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float mgp_wave_sum(float v) {
  v = mgp_dpp_add<0x111, 0xf>(v);   // row_shr:1
  v = mgp_dpp_add<0x112, 0xf>(v);   // row_shr:2
  v = mgp_dpp_add<0x114, 0xf>(v);   // row_shr:4
  v = mgp_dpp_add<0x118, 0xf>(v);   // row_shr:8  -> lane 15 of each row holds the row total
  v = mgp_dpp_add<0x142, 0xa>(v);   // row_bcast15 into rows 1 and 3
  v = mgp_dpp_add<0x143, 0xc>(v);   // row_bcast31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over the 4 lanes of a quad, result in all 4 (same order as xor 1 then xor 2)
__device__ __forceinline__ float mgp_quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
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
