// Sustained whole-chip rate of v_mfma_f32_32x32x16_{bf16,f16} on random operands (4 independent accumulators per wave,
// 4 waves per SIMD, every CU busy): is the f16 flavour slower under the power cap than the bf16 one?
// hipcc --offload-arch=gfx950 -O2 scripts/exp/mfma_rate_probe.hip -o scripts/exp/mfma_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ __launch_bounds__(256) void burn(const u32x4* __restrict__ in, float* __restrict__ out, int iters) {
  u32x4 a = in[threadIdx.x], b = in[256 + threadIdx.x];
  f32x16 c[4];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (F16)
        c[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c[k], 0, 0, 0);
      else
        c[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c[k], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += c[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int blocks = 256 * 4, iters = 20000;
  u32x4* in; float* out;
  (void)hipMalloc(&in, 512 * 16); (void)hipMalloc(&out, blocks * 256 * 4);
  unsigned h[2048];
  srand(1);
  // random 16-bit patterns with a moderate exponent for both formats (sign 0/1, exponent near bias, random mantissa)
  for (int flavour = 0; flavour < 2; ++flavour) {
    for (int i = 0; i < 2048; ++i) {
      unsigned lo, hi;
      if (flavour == 0) { lo = 0x3f00 | (rand() & 0x80ff); hi = 0x3f00 | (rand() & 0x80ff); }      // bf16 ~ [0.5, 1)
      else { lo = 0x3800 | (rand() & 0x83ff); hi = 0x3800 | (rand() & 0x83ff); }                    // f16 ~ [0.5, 1)
      h[i] = lo | (hi << 16);
    }
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0, 0);
      if (flavour == 0) hipLaunchKernelGGL(burn<false>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(burn<true>, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double flops = 2.0 * 32 * 32 * 16 * 4.0 * iters * 4.0 * blocks;
      printf("%s: %.2f ms, %.0f TFLOP/s\n", flavour == 0 ? "bf16" : "f16 ", ms, flops / ms / 1e9);
    }
  }
  return 0;
}
