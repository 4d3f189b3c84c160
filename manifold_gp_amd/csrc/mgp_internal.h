// Internal (non-exported) entry points shared between the translation units of libmgp_hip.
#pragma once
#include "mgp_hip.h"

// mgp_spmm_fused plus: `skip` (device flag: the launch is a no-op when non-zero) and `tick`
// (device counter incremented once per non-skipped launch) -- used by the CG iteration graph.
int mgp_spmm_fused_ex(const mgp_csr_t* L, const float* X, int C, float* Y, float a, float b,
                      const float* pre, const float* post, const float* base, float cb, float co,
                      const float* dotw, float* dot_partials, const int* skip, int* tick, void* stream);

// operator chain with the same hooks on its LAST SpMM
int mgp_operator_apply_ex(const mgp_operator_t* op, const float* X, int C, float* Y, const float* dotw,
                          float* dot_partials, const int* skip, int* tick, void* work, size_t work_bytes,
                          void* stream);
