// What does ds_read_b64_tr_b16 deliver?  LDS image [16 rows][64 cols] of 16-bit ids (row * 64 + col); every lane of a
// 16-lane group supplies the address of (row q = (l & 15) >> 2 [+ 4 * group], cols 4p .. 4p+3, p = l & 3); print what
// each lane receives.  (experiment; not part of the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ short lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int q = l >> 2, p = l & 3;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + (q + 4 * g) * 64 + 4 * p));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  short* d;
  short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / 64, h[l * 4 + e] % 64);
    printf("\n");
  }
  return 0;
}
