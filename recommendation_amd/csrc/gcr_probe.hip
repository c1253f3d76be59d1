// Memory-system probes for bench.py's roofline block ("confirm with a device-memcpy /
// STREAM probe on the box and report that too").  Three access shapes, all with 16 B per lane:
//   copy    dst[i] = src[i]                        1 read + 1 write per byte
//   read    sink += src[i]                         read only
//   gather  out[k] = sum of 64 table rows idx[64k..64k+63], 256-B rows, 16 loads in flight per
//           wave — the access shape of spmm_parts without its CSR streams
// They are measurement aids, not part of the recommender path.
#include "gcr_common.h"

namespace {

// every block owns one contiguous slice; a thread keeps eight 16-B loads in flight, 4 KB apart
__global__ __launch_bounds__(256) void probe_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                         int64_t n4) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(n4, lo + per);
  int64_t i = lo + threadIdx.x;
  for (; i + 7 * 256 < hi; i += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) dst[i + 256 * u] = v[u];
  }
  for (; i < hi; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void probe_read_kernel(const float4* __restrict__ src, int64_t n4,
                                                         float* __restrict__ sink) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(n4, lo + per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t i = lo + threadIdx.x;
  for (; i + 7 * 256 < hi; i += 8 * 256) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[i + 256 * u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
    }
  }
  for (; i < hi; i += 256) {
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
  hipLaunchKernelGGL(probe_copy_kernel, dim3(256 * 32), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
                     (float4*)dst, n_floats / 4);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_probe_read_f32(const float* src, int64_t n_floats, float* sink, void* stream) {
  GCR_CHECK_ARG(src != nullptr && sink != nullptr && n_floats >= 0 && (n_floats & 3) == 0);
  GCR_CHECK_ARG(((uintptr_t)src & 15) == 0);
  if (n_floats == 0) return GCR_OK;
  hipLaunchKernelGGL(probe_read_kernel, dim3(256 * 32), dim3(256), 0, (hipStream_t)stream, (const float4*)src,
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
