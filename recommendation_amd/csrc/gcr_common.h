// Shared helpers for the libgcr HIP sources (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gcr.h"

#define GCR_WAVE 64

#define GCR_CHECK_ARG(cond)       \
  do {                            \
    if (!(cond)) return GCR_EINVAL; \
  } while (0)

static inline int32_t gcr_hip_status(hipError_t e) {
  return e == hipSuccess ? GCR_OK : (int32_t)(GCR_HIP_ERROR_BASE - (int32_t)e);
}

#define GCR_LAUNCH_STATUS() gcr_hip_status(hipGetLastError())

__device__ __forceinline__ int gcr_readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float gcr_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// full-wave butterfly sum: every lane ends with the total
__device__ __forceinline__ float gcr_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, GCR_WAVE);
  return v;
}
__device__ __forceinline__ float gcr_wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, GCR_WAVE));
  return v;
}
