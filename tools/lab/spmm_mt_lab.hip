// Lab: wide SpMM (C = 128 / 64 columns) on the fp32 matrix cores over 16-row tiles made dense in their own distinct columns.
// A tile of 16 rows touches D distinct columns (C3 graph: mean 291); its rows are stored as a dense 16 x D block in the operand
// layout of v_mfma_f32_16x16x4_f32 (step s: lane (i, kq) holds A[i][d(4 s + kq)]), the distinct X rows are gathered ONCE per
// tile (16 bytes per lane: lane (j, kq) loads X[d(4 s + kq)][64 cb + 4 j .. + 3], whose four components are the B operands of
// four MFMAs that produce output columns 64 cb + 4 j + e).  No LDS, no barriers, one wave per tile.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libspmm_mt_lab.so spmm_mt_lab.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// A wave owns (tile, 64-column block).  A tile's steps go in BLOCKS of four (16 dictionary entries, 16 MFMAs = 512 cycles of the
// pipe); a wave keeps THREE blocks of operands in flight ahead of the one it multiplies (ring of four buffer sets, the loop is
// unrolled by four blocks = one 64-entry dictionary batch): with one block in flight the launch was bound by its longest tile
// (54 blocks) times the memory latency (67 us at C = 128).  All loads are inline asm into fixed registers; the waits are
// `s_waitcnt vmcnt(N)` with N = the loads younger than the block waited for (vector memory operations of a wave complete in
// order).  Order of a body's memory operations, body = blocks k .. k + 3 (k a multiple of 4), R(i) = the 8 loads of block i,
// D(b) = the dictionary batch of blocks 4 b .. 4 b + 3:
//     ... R(k) R(k+1) R(k+2) | D(k/4 + 2) . wait(k) R(k+3) . wait(k+1) R(k+4) . wait(k+2) R(k+5) . wait(k+3) R(k+6) | ...
// so wait(k), wait(k+1), wait(k+2) leave 17 operations in flight and wait(k+3) 16.
struct MtBuf {
  float a[4];
  f32x4 b[4];
};

__device__ __forceinline__ void mt_request(MtBuf& nb, int dq, int p0, int kq, int joff, int rowbytes, int so,
                                           __amdgpu_buffer_rsrc_t rimg, __amdgpu_buffer_rsrc_t rx, int lane4) {
  int off[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) off[p] = __builtin_amdgcn_ds_bpermute((4 * (p0 + p) + kq) * 4, dq) * rowbytes + joff;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:%4" : "=v"(nb.a[p]) : "v"(lane4), "s"(rimg), "s"(so), "n"(256 * p));
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(nb.b[p]) : "v"(off[p]), "s"(rx));
  }
  // the four addresses stay live (= in registers of their own) until the last load of the group has been issued: the compiler
  // believes a load's destination is written AT the asm statement and is free to compute a later address in an earlier load's
  // destination, which the hardware may overwrite first when the issue of the later load stalls
  asm volatile("" :: "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]));
}

template <int N>
__device__ __forceinline__ void mt_wait(MtBuf& cb) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(cb.a[0]) : "n"(N));
#pragma unroll
  for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(cb.a[p]), "+v"(cb.b[p]));
}

__device__ __forceinline__ void mt_mfma(const MtBuf& cb, int s0, int S, f32x4 (&acc)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float av = s0 + p < S ? cb.a[p] : 0.f;     // a step past the tile's end holds the next tile's operands
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, cb.b[p][e], acc[e], 0, 0, 0);
  }
}

// sptr [T + 1]: steps before tile t, every tile's count a multiple of 4 (a block); dcol [4 * steps + 192]: distinct columns,
// padded with a valid column (its image entries are 0); img [64 * (steps + 32)]
__global__ __launch_bounds__(256) void spmm_mt_kernel(const int* __restrict__ sptr, const int* __restrict__ dcol,
                                                      const float* __restrict__ img, const float* __restrict__ X,
                                                      float* __restrict__ Y, int n, int T, int C, int NCB, int img_bytes, int dic_bytes) {
  const int lane = threadIdx.x & 63, j = lane & 15, kq = lane >> 4;
  // consecutive tiles (locality order: they share X rows) on ONE XCD, so that its L2 holds its slice of X: workgroup pb runs on
  // XCD pb % 8; XCD x takes the x-th run of gridDim / 8 logical workgroups
  const int per = (int)gridDim.x / 8, rem = (int)gridDim.x % 8, xcd = (int)blockIdx.x % 8;
  const int lb = xcd * per + (xcd < rem ? xcd : rem) + (int)blockIdx.x / 8;
  const int w = __builtin_amdgcn_readfirstlane(lb * 4 + (int)(threadIdx.x >> 6));
  const int t = w / NCB, cb = w % NCB;               // the column blocks of a tile side by side: they share its image
  if (t >= T) return;
  const int base = __builtin_amdgcn_readfirstlane(sptr[t]), S = __builtin_amdgcn_readfirstlane(sptr[t + 1]) - base;
  const int blk0 = base >> 2, NB = (S + 3) >> 2;       // host side: base is a multiple of 4
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), (short)0, n * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rimg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), (short)0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rdic = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(dcol), (short)0, dic_bytes, 0x00020000);
  const int lane4 = lane * 4, rowbytes = C * 4, joff = cb * 256 + j * 16;
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  MtBuf buf0, buf1, buf2, buf3;
  int dq0, dq1, dq2;          // dictionary batches of the body in hand, the next one, and the one in flight
  const int dic0 = blk0 * 64, img0 = blk0 * 1024;      // byte offsets of the tile's first block
  // prologue: batches 0 and 1, then blocks 0, 1, 2
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dq0) : "v"(lane4), "s"(rdic), "s"(dic0));
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:256" : "=v"(dq1) : "v"(lane4), "s"(rdic), "s"(dic0));
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq0), "+v"(dq1));
  mt_request(buf0, dq0, 0, kq, joff, rowbytes, img0, rimg, rx, lane4);
  mt_request(buf1, dq0, 4, kq, joff, rowbytes, img0 + 1024, rimg, rx, lane4);
  mt_request(buf2, dq0, 8, kq, joff, rowbytes, img0 + 2048, rimg, rx, lane4);
  for (int k = 0; k < NB; k += 4) {
    const int so = img0 + k * 1024, sd = dic0 + k * 64;
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:512" : "=v"(dq2) : "v"(lane4), "s"(rdic), "s"(sd));
    mt_wait<17>(buf0);
    mt_request(buf3, dq0, 12, kq, joff, rowbytes, so + 3 * 1024, rimg, rx, lane4);
    __builtin_amdgcn_sched_barrier(0);      // requests stay in front of the block's MFMAs (left alone, the scheduler sinks them)
    mt_mfma(buf0, 4 * k, S, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<17>(buf1);
    mt_request(buf0, dq1, 0, kq, joff, rowbytes, so + 4 * 1024, rimg, rx, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma(buf1, 4 * k + 4, S, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<17>(buf2);
    mt_request(buf1, dq1, 4, kq, joff, rowbytes, so + 5 * 1024, rimg, rx, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma(buf2, 4 * k + 8, S, acc);
    __builtin_amdgcn_sched_barrier(0);
    mt_wait<16>(buf3);
    mt_request(buf2, dq1, 8, kq, joff, rowbytes, so + 6 * 1024, rimg, rx, lane4);
    __builtin_amdgcn_sched_barrier(0);
    mt_mfma(buf3, 4 * k + 12, S, acc);
    __builtin_amdgcn_sched_barrier(0);
    dq0 = dq1;
    // the batch requested at the top of this body is older than R(k+3), which wait(k+3) has seen land: 24 = R(k+4..k+6) waits
    // for nothing new, it only tells the compiler where dq2 becomes readable
    asm volatile("s_waitcnt vmcnt(24)" : "+v"(dq2));
    dq1 = dq2;
  }
  // The last requests (blocks past the tile's end) are still in flight INTO the buffers: they must land before the compiler may
  // reuse those registers -- without this wait the epilogue's store addresses lived in registers a late load then overwrote (a
  // memory fault on the first run of the pipelined version).
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(dq0), "+v"(dq1));
  mt_wait<0>(buf0); mt_wait<0>(buf1); mt_wait<0>(buf2); mt_wait<0>(buf3);
  // acc[e][r]: row 16 t + 4 kq + r, column 64 cb + 4 j + e
  if (64 * cb + 4 * j < C) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + 4 * kq + r;
      if (row < n) {
        f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        *reinterpret_cast<f32x4*>(Y + (int64_t)row * C + 64 * cb + 4 * j) = v;
      }
    }
  }
}

extern "C" int lab_spmm_mt(const int* sptr, const int* dcol, const float* img, const float* X, float* Y, int n, int T, int C,
                           int reps, float* ms, void* stream, int img_bytes, int dic_bytes) {
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int NCB = (C + 63) / 64;
  const int grid = (T * NCB + 3) / 4;
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(spmm_mt_kernel, dim3(grid), dim3(256), 0, st, sptr, dcol, img, X, Y, n, T, C, NCB, img_bytes, dic_bytes);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) return 2;
  hipEventElapsedTime(ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
