// Ablation harness for the InfoNCE forward tile loop (not part of libgcr): which of
// {epilogue VALU, LDS staging + barrier, occupancy} costs the MFMA pipe its idle time?
//   hipcc --offload-arch=gfx950 -O3 scripts/exp_infonce.hip -o /tmp/exp_infonce && /tmp/exp_infonce
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 64, KH = 32, STRIDE = D + 4, NT = 2, TILE = 32;

template <bool EPI, bool STAGE, bool BARRIER, int MINW>
__global__ __launch_bounds__(256, MINW) void k(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows,
                                               int64_t tiles, float* __restrict__ out) {
  __shared__ __align__(16) float lds[2][TILE * STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i32 = lane & 31, h = lane >> 5;
  float bfrag[NT][KH];
  for (int t = 0; t < NT; ++t)
    for (int s = 0; s < KH; ++s) bfrag[t][s] = a[((blockIdx.x * 4 + wave) * 64 + 32 * t + i32) % 2048 * D + h * KH + s];
  float m_run[NT] = {-1e30f, -1e30f}, l_run[NT] = {0.f, 0.f};
  for (int i = tid; i < 2 * TILE * STRIDE; i += 256) (&lds[0][0])[i] = b[i % (TILE * D)];
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * tiles;
  for (int64_t tt = 0; tt < tiles; ++tt) {
    const int cur = tt & 1;
    float4 regs[2];
    if (STAGE) {
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u, row = idx / 16, c4 = idx % 16;
        regs[u] = *reinterpret_cast<const float4*>(b + ((t0 + tt + 1) * TILE + row) % n_rows * D + 4 * c4);
      }
    }
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* base = lds[cur] + i32 * STRIDE + h * KH;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(base + 4 * q);
      const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
    }
    if (EPI) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        float tmax = acc[t][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, acc[t][r]);
        const float m_new = fmaxf(m_run[t], tmax);
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(acc[t][r] - m_new);
        l_run[t] = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m_new) + sum;
        m_run[t] = m_new;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        asm volatile("" ::"v"(acc[t]));
        l_run[t] += acc[t][0];
      }
    }
    if (STAGE) {
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u, row = idx / 16, c4 = idx % 16;
        *reinterpret_cast<float4*>(lds[cur ^ 1] + row * STRIDE + 4 * c4) = regs[u];
      }
    }
    if (BARRIER) __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = m_run[0] + l_run[0] + m_run[1] + l_run[1];
}


// software-pipelined variant: epilogue of the previous tile interleaved with this tile's MFMAs;
// staged rows are multiplied/written only after the MFMA stream (late vmcnt wait)
template <int VALU_PER_MFMA, int MINW>
__global__ __launch_bounds__(256, MINW) void kp(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows,
                                                int64_t tiles, float* __restrict__ out) {
  __shared__ __align__(16) float lds[2][TILE * STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i32 = lane & 31, h = lane >> 5;
  float bfrag[NT][KH];
  for (int t = 0; t < NT; ++t)
    for (int s = 0; s < KH; ++s) bfrag[t][s] = a[((blockIdx.x * 4 + wave) * 64 + 32 * t + i32) % 2048 * D + h * KH + s];
  float m_run[NT] = {-1e30f, -1e30f}, l_run[NT] = {0.f, 0.f};
  for (int i = tid; i < 2 * TILE * STRIDE; i += 256) (&lds[0][0])[i] = b[i % (TILE * D)];
  __syncthreads();
  const int64_t t0 = (int64_t)blockIdx.x * tiles;
  f32x16 accs[2][NT];
  for (int p = 0; p < 2; ++p)
    for (int t = 0; t < NT; ++t)
      for (int r = 0; r < 16; ++r) accs[p][t][r] = 0.f;
  auto step = [&](int64_t tt, const int cur, f32x16 (&acc)[NT], const f32x16 (&old)[NT]) {
    float4 regs[2];
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u, row = idx / 16, c4 = idx % 16;
      regs[u] = *reinterpret_cast<const float4*>(b + ((t0 + tt + 1) * TILE + row) % n_rows * D + 4 * c4);
    }
    for (int t = 0; t < NT; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* base = lds[cur] + i32 * STRIDE + h * KH;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(base + 4 * q);
      const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float tmax = old[t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, old[t][r]);
      const float m_new = fmaxf(m_run[t], tmax);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(old[t][r] - m_new);
      l_run[t] = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m_new) + sum;
      m_run[t] = m_new;
    }
    if (VALU_PER_MFMA > 0) {
#pragma unroll
      for (int i = 0; i < KH * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, VALU_PER_MFMA, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    for (int u = 0; u < 2; ++u) {
      const int idx = tid + 256 * u, row = idx / 16, c4 = idx % 16;
      *reinterpret_cast<float4*>(lds[cur ^ 1] + row * STRIDE + 4 * c4) = regs[u];
    }
    __syncthreads();
  };
  for (int64_t tt = 0; tt + 1 < tiles; tt += 2) {
    step(tt, 0, accs[0], accs[1]);
    step(tt + 1, 1, accs[1], accs[0]);
  }
  out[blockIdx.x * 256 + tid] = m_run[0] + l_run[0] + m_run[1] + l_run[1] + accs[1][0][0];
}

// LDS-DMA staged variant (timing only; measured 123.6 TF vs 126.1 TF register-staged at 3 blocks/CU,
// 120 vs 109 TF at 2 blocks/CU: staging is not what limits the 3-blocks/CU configuration the library
// ships): global_load_lds_dwordx4 straight into an UNPADDED, XOR-swizzled tile
// (16-B chunk c of row j lives at chunk c ^ (j & 15)); no staging VGPRs, no ds_write, no multiply.
template <int MINW>
__global__ __launch_bounds__(256, MINW) void kdma(const float* __restrict__ a, const float* __restrict__ b, int64_t n_rows,
                                                  int64_t tiles, float* __restrict__ out) {
  __shared__ __align__(16) float lds[2][TILE * D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i32 = lane & 31, h = lane >> 5;
  float bfrag[NT][KH];
  for (int t = 0; t < NT; ++t)
    for (int s = 0; s < KH; ++s) bfrag[t][s] = a[((blockIdx.x * 4 + wave) * 64 + 32 * t + i32) % 2048 * D + h * KH + s];
  float m_run[NT] = {-1e30f, -1e30f}, l_run[NT] = {0.f, 0.f};
  const int64_t t0 = (int64_t)blockIdx.x * tiles;
  auto dma_tile = [&](int64_t tile, float* buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ii = wave * 2 + u;                       // 1-KiB piece = rows 4*ii .. 4*ii+3
      const int row = 4 * ii + (lane >> 4), phys = lane & 15, logical = phys ^ (row & 15);
      const float* src = b + ((tile * TILE + row) % n_rows) * D + logical * 4;
      __builtin_amdgcn_global_load_lds(src, buf + ii * 256, 16, 0, 0);
    }
  };
  dma_tile(t0, lds[0]);
  __syncthreads();
  for (int64_t tt = 0; tt < tiles; ++tt) {
    const int cur = tt & 1;
    dma_tile(t0 + tt + 1, lds[cur ^ 1]);
    f32x16 acc[NT];
    for (int t = 0; t < NT; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* rowp = lds[cur] + i32 * D;
#pragma unroll
    for (int q = 0; q < KH / 4; ++q) {
      const int phys = (8 * h + q) ^ (i32 & 15);
      const float4 av = *reinterpret_cast<const float4*>(rowp + 4 * phys);
      const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float tmax = acc[t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, acc[t][r]);
      const float m_new = fmaxf(m_run[t], tmax);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(acc[t][r] - m_new);
      l_run[t] = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m_new) + sum;
      m_run[t] = m_new;
    }
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = m_run[0] + l_run[0] + m_run[1] + l_run[1];
}

template <class K>
float run(K kern, int blocks, const float* a, const float* b, int64_t n, int64_t tiles, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, b, n, tiles, out);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, a, b, n, tiles, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  const int64_t n = 1000000;
  std::vector<float> ha(2048 * D), hb(n * D);
  for (auto& v : ha) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
  float *a, *b, *out;
  hipMalloc(&a, ha.size() * 4);
  hipMalloc(&b, hb.size() * 4);
  hipMalloc(&out, 4096 * 256 * 4);
  hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  const int64_t tiles = 488;
#define RUN(name, EPI, STAGE, BAR, MINW, blocks)                                                         \
  {                                                                                                      \
    float ms = run(k<EPI, STAGE, BAR, MINW>, blocks, a, b, n, tiles, out);                               \
    double fl = 2.0 * blocks * 256.0 * tiles * 32 * 64;                                                   \
    printf("%-44s blocks %4d: %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);                      \
  }
  RUN("full (epi+stage+barrier), 2 blk/CU", true, true, true, 2, 512);
  RUN("no epilogue", false, true, true, 2, 512);
  RUN("no stage, barrier only", true, false, true, 2, 512);
  RUN("no stage, no barrier", true, false, false, 2, 512);
  RUN("MFMA only (no epi/stage/barrier)", false, false, false, 2, 512);
  RUN("full, 1 blk/CU", true, true, true, 2, 256);
  RUN("MFMA only, 1 blk/CU", false, false, false, 2, 256);
  RUN("full, 3 blk/CU (768)", true, true, true, 2, 768);
#define RUNP(name, V, MINW, blocks)                                                                      \
  {                                                                                                      \
    float ms = run(kp<V, MINW>, blocks, a, b, n, tiles, out);                                            \
    double fl = 2.0 * blocks * 256.0 * tiles * 32 * 64;                                                   \
    printf("%-44s blocks %4d: %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);                      \
  }
  RUNP("pipelined, sched 3 VALU/MFMA, 2 blk/CU", 3, 2, 512);
  RUNP("pipelined, sched 2 VALU/MFMA, 2 blk/CU", 2, 2, 512);
  RUNP("pipelined, no sched hint, 2 blk/CU", 0, 2, 512);
  RUNP("pipelined, sched 3, 1 blk/CU", 3, 2, 256);
  RUNP("pipelined, sched 3, 768 blocks", 3, 2, 768);
  RUN("full, 4 blk/CU (1024)", true, true, true, 2, 1024);
#define RUND(name, MINW, blocks)                                                                         \
  {                                                                                                      \
    float ms = run(kdma<MINW>, blocks, a, b, n, tiles, out);                                             \
    double fl = 2.0 * blocks * 256.0 * tiles * 32 * 64;                                                   \
    printf("%-44s blocks %4d: %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);                      \
  }
  RUN("full, bounds(256,3) 768 blocks (again)", true, true, true, 3, 768);
  RUND("LDS-DMA swizzled, 768 blocks (3/CU)", 3, 768);
  RUND("LDS-DMA swizzled, 512 blocks", 3, 512);
  RUND("LDS-DMA swizzled, 1024 blocks (4/CU)", 4, 1024);
  RUN("full, bounds(256,3) 768 blocks (again)", true, true, true, 3, 768);
  RUND("LDS-DMA swizzled, 768 blocks (3/CU)", 3, 768);
  RUN("full, bounds(256,4) 1024 blocks", true, true, true, 4, 1024);
  RUN("full, bounds(256,4) 768 blocks", true, true, true, 4, 768);
  RUN("full, bounds(256,3) 768 blocks", true, true, true, 3, 768);
  RUN("full, bounds(256,4) 2048 blocks", true, true, true, 4, 2048);
  return 0;
}
