// Lab: what it costs to re-point a captured CG-chunk-sized graph at new kernel arguments.
//   (a) capture (stream capture of 30 launches)  (b) hipGraphInstantiate  (c) hipGraphExecUpdate from a re-captured graph
//   (d) hipGraphExecKernelNodeSetParams on every node  (e) hipGraphExecDestroy.     hipcc --offload-arch=gfx950 -O2 graph_update.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { float* p[16]; float s[8]; int n; };
__global__ void k(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0][a.n] = a.s[0]; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  float* buf; CK(hipMalloc(&buf, 4096));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const int nodes = 30;
  auto capture = [&](float scale, hipGraph_t* g) -> int {
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) { Args a{}; a.p[0] = buf; a.s[0] = scale; a.n = i; hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, a); }
    CK(hipStreamEndCapture(st, g));
    return 0;
  };
  for (int rep = 0; rep < 4; ++rep) {
    hipGraph_t g1, g2; hipGraphExec_t ex;
    double t0 = now(); if (capture(1.f, &g1)) return 1; double t1 = now();
    CK(hipGraphInstantiate(&ex, g1, nullptr, nullptr, 0)); double t2 = now();
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st)); double t3 = now();
    if (capture(2.f, &g2)) return 1; double t4 = now();
    hipGraphNode_t err; hipGraphExecUpdateResult res;
    hipError_t e = hipGraphExecUpdate(ex, g2, &err, &res); double t5 = now();
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st)); double t6 = now();
    float h[32]; CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    // per-node parameter update
    size_t nn = 0; CK(hipGraphGetNodes(g1, nullptr, &nn)); std::vector<hipGraphNode_t> ns(nn); CK(hipGraphGetNodes(g1, ns.data(), &nn));
    double t7 = now();
    for (size_t i = 0; i < nn; ++i) {
      hipKernelNodeParams p; CK(hipGraphKernelNodeGetParams(ns[i], &p));
      Args a = *reinterpret_cast<Args*>(p.kernelParams[0]); a.s[0] = 3.f;
      void* kp[1] = {&a}; p.kernelParams = kp;
      CK(hipGraphExecKernelNodeSetParams(ex, ns[i], &p));
    }
    double t8 = now();
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    float h2[32]; CK(hipMemcpy(h2, buf, sizeof(h2), hipMemcpyDeviceToHost));
    double t9 = now(); CK(hipGraphExecDestroy(ex)); double t10 = now();
    CK(hipGraphDestroy(g1)); CK(hipGraphDestroy(g2));
    printf("rep %d: capture %.0f us, instantiate %.0f us, launch+sync %.0f us, re-capture %.0f us, ExecUpdate %.0f us (%s, result %d, value %.0f), "
           "launch+sync %.0f us, %zu x NodeSetParams %.0f us (value %.0f), ExecDestroy %.0f us\n", rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4,
           hipGetErrorString(e), (int)res, h[3], t6 - t5, nn, t8 - t7, h2[3], t10 - t9);
  }
  return 0;
}
