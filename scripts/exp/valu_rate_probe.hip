// Issue rate of the VALU instructions the operand split is made of (gfx950): cycles per wave-instruction, measured
// with s_memtime around a loop of 8 independent chains x 64 iterations, one wave per SIMD.
// hipcc --offload-arch=gfx950 -O2 scripts/exp/valu_rate_probe.hip -o scripts/exp/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(OP)                                                                                         \
  asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                            \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
               : "v"(c))

#define OP_CVT_PK_F16(i) "v_cvt_pk_f16_f32 %" #i ", %" #i ", %8\n"
#define OP_CVT_PK_BF16(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n"
#define OP_CVT_PKRTZ(i) "v_cvt_pkrtz_f16_f32 %" #i ", %" #i ", %8\n"
#define OP_CVT_F32_F16(i) "v_cvt_f32_f16_e32 %" #i ", %" #i "\n"
#define OP_CVT_F32_F16_SDWA(i) "v_cvt_f32_f16_sdwa %" #i ", %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define OP_SUB(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n"
#define OP_EXP(i) "v_exp_f32_e32 %" #i ", %" #i "\n"
#define OP_LSHL(i) "v_lshlrev_b32_e32 %" #i ", 16, %" #i "\n"
#define OP_AND(i) "v_and_b32_e32 %" #i ", 0xffff0000, %" #i "\n"

template <int WHICH>
__global__ void rate(float* out, long long* cyc) {
  float r[8];
  for (int i = 0; i < 8; ++i) r[i] = out[threadIdx.x + 64 * i];
  const float c = out[1000];
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
    if (WHICH == 0) CHAIN8(OP_CVT_PK_F16);
    if (WHICH == 1) CHAIN8(OP_CVT_PK_BF16);
    if (WHICH == 2) CHAIN8(OP_CVT_PKRTZ);
    if (WHICH == 3) CHAIN8(OP_CVT_F32_F16);
    if (WHICH == 4) CHAIN8(OP_CVT_F32_F16_SDWA);
    if (WHICH == 5) CHAIN8(OP_SUB);
    if (WHICH == 6) CHAIN8(OP_EXP);
    if (WHICH == 7) CHAIN8(OP_LSHL);
    if (WHICH == 8) CHAIN8(OP_AND);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += r[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[WHICH] = t1 - t0;
}

int main() {
  float* d; long long* c;
  (void)hipMalloc(&d, 8192); (void)hipMalloc(&c, 128);
  (void)hipMemset(d, 0, 8192);
  const char* names[9] = {"v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32", "v_cvt_pkrtz_f16_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa",
                          "v_sub_f32", "v_exp_f32", "v_lshlrev_b32", "v_and_b32"};
  hipLaunchKernelGGL(rate<0>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<3>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<5>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<6>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<7>, dim3(1), dim3(64), 0, 0, d, c);
  hipLaunchKernelGGL(rate<8>, dim3(1), dim3(64), 0, 0, d, c);
  long long h[16];
  (void)hipMemcpy(h, c, 128, hipMemcpyDeviceToHost);
  for (int i = 0; i < 9; ++i) printf("%-22s %6.2f counter ticks per wave-instruction\n", names[i], (double)h[i] / (256.0 * 8));
  return 0;
}
