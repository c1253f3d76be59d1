// Memory-system probes for bench.py's roofline block ("confirm with a device-memcpy /
// STREAM probe on the box and report that too").  Three access shapes, all with 16 B per lane:
//   copy    dst[i] = src[i]                        1 read + 1 write per byte
//   read    sink += src[i]                         read only
//   gather  out[k] = sum of 64 table rows idx[64k..64k+63], 256-B rows, 16 loads in flight per
//           wave — the access shape of spmm_parts without its CSR streams
// They are measurement aids, not part of the recommender path.
#include "gcr_common.h"

namespace {

__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                         int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {       // four 16-B loads in flight per lane before the stores
    const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    dst[i] = v0;
    dst[i + stride] = v1;
    dst[i + 2 * stride] = v2;
    dst[i + 3 * stride] = v3;
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void probe_read_kernel(const float4* __restrict__ src, int64_t n4,
                                                         float* __restrict__ sink) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    a.x += v0.x + v1.x + v2.x + v3.x;
    a.y += v0.y + v1.y + v2.y + v3.y;
    a.z += v0.z + v1.z + v2.z + v3.z;
    a.w += v0.w + v1.w + v2.w + v3.w;
  }
  for (; i < n4; i += stride) {
    const float4 v = src[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const float s = gcr_wave_sum(a.x + a.y + a.z + a.w);
  // data-dependent store the compiler cannot drop; practically never taken on random data
  if (s == 1.2345678e33f && (threadIdx.x & 63) == 0) sink[0] = s;
}

__global__ __launch_bounds__(256) void probe_gather_kernel(const float* __restrict__ table,
                                                           const int32_t* __restrict__ idx, int64_t n_groups,
                                                           int n_rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= n_groups) return;
  const int my = min(max(idx[g * 64 + lane], 0), n_rows - 1);   // never dereference a bad id
  float acc = 0.f;
#pragma unroll
  for (int b = 0; b < 64; b += 16) {
    float r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = table[(int64_t)gcr_readlane_i(my, b + u) * 64 + lane];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += r[u];
  }
  out[g * 64 + lane] = acc;
}

}  // namespace

extern "C" int32_t gcr_probe_copy_f32(const float* src, float* dst, int64_t n_floats, void* stream) {
  GCR_CHECK_ARG(src != nullptr && dst != nullptr && n_floats >= 0 && (n_floats & 3) == 0);
  GCR_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  if (n_floats == 0) return GCR_OK;
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                     (float4*)dst, n_floats / 4);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_probe_read_f32(const float* src, int64_t n_floats, float* sink, void* stream) {
  GCR_CHECK_ARG(src != nullptr && sink != nullptr && n_floats >= 0 && (n_floats & 3) == 0);
  GCR_CHECK_ARG(((uintptr_t)src & 15) == 0);
  if (n_floats == 0) return GCR_OK;
  hipLaunchKernelGGL(probe_read_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                     n_floats / 4, sink);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_probe_gather_rows_f32(const float* table, int64_t n_rows, const int32_t* idx, int64_t n_idx,
                                             float* out, void* stream) {
  GCR_CHECK_ARG(table != nullptr && idx != nullptr && out != nullptr);
  GCR_CHECK_ARG(n_rows > 0 && n_rows < (1ll << 31) && n_idx >= 0 && (n_idx & 63) == 0 && n_idx / 256 < (1ll << 31));
  if (n_idx == 0) return GCR_OK;
  const int64_t n_groups = n_idx / 64;
  hipLaunchKernelGGL(probe_gather_kernel, dim3((unsigned)((n_groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     table, idx, n_groups, (int)n_rows, out);
  return GCR_LAUNCH_STATUS();
}
