// Library / device probes.
#include "mgp_common.h"

extern "C" int mgp_version(void) { return 100; }

extern "C" int mgp_device_info(int* cu_count, int* wave_size, size_t* hbm_bytes) {
  int dev = 0;
  MGP_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t p;
  MGP_HIP_TRY(hipGetDeviceProperties(&p, dev));
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (wave_size) *wave_size = p.warpSize;
  if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
  return MGP_OK;
}
