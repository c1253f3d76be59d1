// How many plain VALU instructions hide behind one v_mfma_f32_32x32x16_f16 on gfx950, with one and with two waves
// per SIMD?  Loop body: 1 MFMA (accumulator chain) + K independent v_fma_f32; cycles per iteration from s_memtime.
// hipcc --offload-arch=gfx950 -O2 scripts/exp/coissue_probe.hip -o scripts/exp/coissue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int NM>
__global__ __launch_bounds__(512) void body(const float* __restrict__ in, float* __restrict__ out, long long* cyc, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[(threadIdx.x & 63) + i]; b[i] = (_Float16)in[64 + (threadIdx.x & 63) + i]; }
  f32x16 c[NM];
  for (int m = 0; m < NM; ++m)
    for (int r = 0; r < 16; ++r) c[m][r] = 0.f;
  float v[12];
  for (int i = 0; i < 12; ++i) v[i] = in[(threadIdx.x & 63) + i];
  const float k0 = in[200], k1 = in[201];
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      // one asm block per slot: the MFMA, then K independent v_fma_f32 in program order (the compiler would otherwise
      // group the MFMAs and pack the FMAs)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[m]) : "v"(a), "v"(b));
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k % 12]) : "v"(k0), "v"(k1));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int m = 0; m < NM; ++m)
    for (int r = 0; r < 16; ++r) s += c[m][r];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  // the older wave of a SIMD pair wins the issue arbitration: report the older (wave 0), the younger (wave 4) and the span
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int K>
void run(int blocks_per_cu, float* in, float* out, long long* cyc) {
  const int iters = 2000, NM = 4;
  // one block per CU: 256 threads = one wave per SIMD, 512 threads = two waves per SIMD
  hipLaunchKernelGGL((body<K, NM>), dim3(256), dim3(256 * blocks_per_cu), 0, 0, in, out, cyc, iters);
  long long h[16];
  (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  const int nw = 4 * blocks_per_cu;
  long long lo = h[0], hi = h[1];
  for (int w = 0; w < nw; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
  const double per = 1.0 / (iters * NM);
  printf("K=%2d VALU per MFMA, %d wave(s) per SIMD: wave 0 %6.1f", K, blocks_per_cu, (h[1] - h[0]) * per);
  if (blocks_per_cu == 2) printf("  wave 4 %6.1f", (h[9] - h[8]) * per);
  printf("  SIMD time per MFMA slot (span / MFMAs of one wave) %6.1f\n", (hi - lo) * per);
}

int main() {
  float *in, *out; long long* cyc;
  (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 256 * 2 * 256 * 4); (void)hipMalloc(&cyc, 128);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.5f + 0.001f * i;
  h[200] = 0.999f; h[201] = 0.001f;
  (void)hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, in, out, cyc); run<2>(w, in, out, cyc); run<4>(w, in, out, cyc); run<6>(w, in, out, cyc);
    run<8>(w, in, out, cyc); run<10>(w, in, out, cyc); run<12>(w, in, out, cyc);
  }
  return 0;
}
