// Does v_mfma_f32_32x32x16_f16 honour f16 sub-normal inputs, and does the f32 -> f16 pack produce them (RNE)?
// hipcc --offload-arch=gfx950 -O2 scripts/exp/f16_denorm_probe.hip -o /tmp/f16probe && /tmp/f16probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float* in, float* out, unsigned* bits) {
  const float a = in[0], b = in[1];
  const f16x2 pa = __builtin_convertvector((f32x2){a, a}, f16x2);
  const f16x2 pb = __builtin_convertvector((f32x2){b, b}, f16x2);
  f16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = pa[0]; vb[i] = pb[0]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, c, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = c[0];
    bits[0] = __builtin_bit_cast(unsigned, pa);
    // RNE check: 1 + 2^-11 (tie) -> 1.0 (even); 1 + 3*2^-11 (tie) -> 1 + 2^-9 (even)
    const f16x2 t = __builtin_convertvector((f32x2){in[2], in[3]}, f16x2);
    bits[1] = __builtin_bit_cast(unsigned, t);
    out[1] = (float)pa[0];
  }
}

int main() {
  float h_in[4] = {0x1p-20f, 0x1p10f, 1.0f + 0x1p-11f, 1.0f + 3 * 0x1p-11f};
  float *d_in, *d_out; unsigned* d_bits;
  hipMalloc(&d_in, 16); hipMalloc(&d_out, 8); hipMalloc(&d_bits, 8);
  hipMemcpy(d_in, h_in, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_bits);
  float out[2]; unsigned bits[2];
  hipMemcpy(out, d_out, 8, hipMemcpyDeviceToHost);
  hipMemcpy(bits, d_bits, 8, hipMemcpyDeviceToHost);
  printf("mfma(16 x 2^-20 * 2^10) = %g (expect 0.015625 if sub-normals are honoured, 0 if flushed)\n", out[0]);
  printf("f16(2^-20) bits = 0x%04x (expect 0x0010), back to f32 = %g\n", bits[0] & 0xffff, out[1]);
  printf("f16(1+2^-11) = 0x%04x (RNE: 0x3c00), f16(1+3*2^-11) = 0x%04x (RNE: 0x3c02)\n", bits[1] & 0xffff, bits[1] >> 16);
  return 0;
}
