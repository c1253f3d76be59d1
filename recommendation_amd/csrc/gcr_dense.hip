// Row-wise glue of the SEPT / MHCN encoders' backward pass as single-pass kernels:
//   normalize_bwd_n   d(A x) of `F.normalize(A x)` from the SAVED NORMALISED rows (sept.py:223-224, mhcn.py:440-457):
//                         dz = (g_n - n <n, g_n>) * inv + g_raw
//                     one read of each input, one write (torch ran it as 4-6 element-wise / reduction passes over [U, d]
//                     per operator and layer);
//   rows_dot_vec      out[r] = <x[r], v> (rocBLAS' gemv of this row-major shape ran 187 us for 64 MB: 0.34 TB/s);
//   weighted_colsum   out[c] = sum_r w[r] x[r][c]: the gradient of v in `em @ v` (the channel-attention logits,
//                     mhcn.py:414 reassociated as em @ (attention_mat attention^T)); the library's transposed GEMV ran
//                     1.1 ms at n = 250K, d = 64;
//   gram_tn           out[dx, dg] = X^T G for tall X [n, dx], G [n, dg] — the weight gradient of every `em @ W` of
//                     MHCN's gating / attention (mhcn.py:404-420: W is d x d, n = #users).  The library GEMM gives the
//                     d x d output to one or two workgroups that walk all n rows (0.52 ms at n = 250K, d = 64:
//                     6.7 of the 18 ms of config 5's fwd + bwd); here the rows are split over the whole chip on the f32
//                     MFMA (exact f32 products) and the per-workgroup partials are summed in a fixed order.
//   gate              out = em * sigmoid(z + bias) and its backward (mhcn.py:404-411 self_gating / self_supervised_gating
//                     around the library GEMM z = em W): d em, d z and the bias gradient's partial column sums in one pass
//                     each (torch: add, sigmoid, multiply forward; eight element-wise / reduction launches backward);
//   channel_mix       mhcn.py:413-420 channel_attention on the logits e_k . v: per row the three logits, their softmax,
//                     mixed = sum_k score_k e_k (+ scale * extra: the `+ simple / 2` of mhcn.py:443) in one pass; the
//                     backward gives d e_k = score_k g + dlogit_k v, d extra and the partial sums of d v in one pass
//                     (torch: ~30 launches per call, three calls per step).
#include "gcr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16 lanes per row, float4 per lane and 64 columns (d % 4 == 0); out may alias g_n or g_raw.
// FROM_RAW: `nrm` holds the RAW rows z = A x (the forward did not keep the normalised copy): n = z * inv on the fly.
template <bool FROM_RAW>
__global__ __launch_bounds__(256) void normalize_bwd_n_kernel(const float* __restrict__ nrm, const float* __restrict__ inv,
                                                              const float* g_n, const float* g_raw, int64_t rows, int d,
                                                              float* out) {
  const int l16 = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < rows; r += (int64_t)gridDim.x * 16) {
    const float s = inv[r];
    float dot = 0.f;
    float4 nv[4], gv[4];                                     // d <= 256: at most four float4 per lane
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = l16 * 4 + 64 * q;
      if (c < d) {
        nv[q] = *reinterpret_cast<const float4*>(nrm + r * d + c);
        if (FROM_RAW) { nv[q].x *= s; nv[q].y *= s; nv[q].z *= s; nv[q].w *= s; }
        gv[q] = *reinterpret_cast<const float4*>(g_n + r * d + c);
        dot += nv[q].x * gv[q].x + nv[q].y * gv[q].y + nv[q].z * gv[q].z + nv[q].w * gv[q].w;
      }
    }
    dot += __shfl_xor(dot, 8, 16);
    dot += __shfl_xor(dot, 4, 16);
    dot += __shfl_xor(dot, 2, 16);
    dot += __shfl_xor(dot, 1, 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = l16 * 4 + 64 * q;
      if (c < d) {
        float4 o = make_float4((gv[q].x - nv[q].x * dot) * s, (gv[q].y - nv[q].y * dot) * s,
                               (gv[q].z - nv[q].z * dot) * s, (gv[q].w - nv[q].w * dot) * s);
        if (g_raw != nullptr) {
          const float4 a = *reinterpret_cast<const float4*>(g_raw + r * d + c);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *reinterpret_cast<float4*>(out + r * d + c) = o;
      }
    }
  }
}

// out[r] = <x[r, :], v>: 16 lanes per row, float4 loads (d % 4 == 0) or scalar ones
__global__ __launch_bounds__(256) void rows_dot_vec_kernel(const float* __restrict__ x, const float* __restrict__ v, int64_t n,
                                                           int d, float* __restrict__ out) {
  const int l16 = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += (int64_t)gridDim.x * 16) {
    float dot = 0.f;
    if ((d & 3) == 0) {
      for (int c = l16 * 4; c < d; c += 64) {
        const float4 a = *reinterpret_cast<const float4*>(x + r * d + c);
        const float4 b = *reinterpret_cast<const float4*>(v + c);
        dot += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
    } else {
      for (int c = l16; c < d; c += 16) dot += x[r * d + c] * v[c];
    }
    dot += __shfl_xor(dot, 8, 16);
    dot += __shfl_xor(dot, 4, 16);
    dot += __shfl_xor(dot, 2, 16);
    dot += __shfl_xor(dot, 1, 16);
    if (l16 == 0) out[r] = dot;
  }
}

// One workgroup = four waves over one chunk of rows; wave w owns output tiles w, w + 4, ... (32 x 32 each, TX x TG of
// them).  v_mfma_f32_32x32x2_f32: A[m][k] from lane (m = lane & 31, k = lane >> 5), B[k][n] likewise — with k = the row
// pair (r, r + 1) both operands are plain coalesced 128-B loads of two consecutive rows.
template <int TX, int TG>
__global__ __launch_bounds__(256) void gram_tn_kernel(const float* __restrict__ x, const float* __restrict__ g, int64_t n,
                                                      int64_t rows_per_block, float* __restrict__ part) {
  constexpr int DX = 32 * TX, DG = 32 * TG, NTILE = TX * TG, PER_WAVE = (NTILE + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c32 = lane & 31, h = lane >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  f32x16 acc[PER_WAVE];
#pragma unroll
  for (int t = 0; t < PER_WAVE; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  constexpr int U = 8;                                       // row pairs in flight
  for (int64_t r = r0; r < r1; r += 2 * U) {
    float av[PER_WAVE][U], bv[PER_WAVE][U];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
      const int tile = wave + 4 * t;
      const int tm = tile / TG, tn = tile % TG;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t row = r + 2 * u + h;
        const bool ok = row < r1 && tile < NTILE;
        av[t][u] = ok ? x[row * DX + 32 * tm + c32] : 0.f;
        bv[t][u] = ok ? g[row * DG + 32 * tn + c32] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][u], bv[t][u], acc[t], 0, 0, 0);
  }
  float* dst = part + (int64_t)blockIdx.x * (DX * DG);
#pragma unroll
  for (int t = 0; t < PER_WAVE; ++t) {
    const int tile = wave + 4 * t;
    if (tile < NTILE) {
      const int tm = tile / TG, tn = tile % TG;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;      // C layout of the 32 x 32 accumulator
        dst[(32 * tm + row) * DG + 32 * tn + c32] = acc[t][r];
      }
    }
  }
}

// out[k] = sum over the workgroups' partials, four lanes per output (each a quarter of the partials, loads eight deep),
// combined in a fixed order: bitwise reproducible.  (One thread per output walking all 1024 partials was a chain of 1024
// dependent memory round trips: 229 us for the 64 x 64 case, ten times the product itself.)
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ part, int nblocks, int elems,
                                                          float* __restrict__ out) {
  const int k = blockIdx.x * 64 + (threadIdx.x >> 2);
  const int q = threadIdx.x & 3;
  float s = 0.f;
  if (k < elems) {
    const int per = (nblocks + 3) / 4;
    const int b0 = q * per, b1 = min(nblocks, b0 + per);
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + u) * elems + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < b1; ++b) s += part[(int64_t)b * elems + k];
  }
  s += __shfl_xor(s, 1, 4);
  s += __shfl_xor(s, 2, 4);
  if (k < elems && q == 0) out[k] = s;
}

// part[block][c] = sum over the block's rows of w[r] x[r][c]  (d <= 256: thread t owns column t % DP of every
// (256 / DP)-th row; DP = d rounded up to a power of two)
__global__ __launch_bounds__(256) void weighted_colsum_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              int64_t n, int d, int dp, int64_t rows_per_block,
                                                              float* __restrict__ part) {
  __shared__ float red[256];
  const int c = threadIdx.x % dp, lane_row = threadIdx.x / dp, step = 256 / dp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  float s = 0.f;
  if (c < d) {
    int64_t r = r0 + lane_row;
    for (; r + 3 * step < r1; r += 4 * step) {
      const float a0 = x[r * d + c] * w[r], a1 = x[(r + step) * d + c] * w[r + step];
      const float a2 = x[(r + 2 * step) * d + c] * w[r + 2 * step], a3 = x[(r + 3 * step) * d + c] * w[r + 3 * step];
      s += (a0 + a1) + (a2 + a3);
    }
    for (; r < r1; r += step) s += x[r * d + c] * w[r];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < dp) {
    float t = 0.f;
    for (int k = 0; k < step; ++k) t += red[threadIdx.x + k * dp];
    if (c < d) part[(int64_t)blockIdx.x * d + c] = t;
  }
}

__device__ __forceinline__ float sigmoid_f32(float t) { return 1.0f / (1.0f + expf(-t)); }

// thread t owns column t % DP of every (256 / DP)-th row (DP = d rounded up to a power of two <= 256): coalesced rows
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ em, const float* __restrict__ z,
                                                       const float* __restrict__ bias, int64_t n, int d, int dp,
                                                       float* __restrict__ out) {
  const int c = threadIdx.x % dp, step = 256 / dp;
  if (c >= d) return;
  const float b = bias != nullptr ? bias[c] : 0.f;
  for (int64_t r = (int64_t)blockIdx.x * step + threadIdx.x / dp; r < n; r += (int64_t)gridDim.x * step) {
    const int64_t k = r * d + c;
    out[k] = em[k] * sigmoid_f32(z[k] + b);
  }
}

// d_em = g sig, d_z = g em sig (1 - sig), part[block][c] = sum over the block's rows of d_z (the bias gradient)
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ g, const float* __restrict__ em,
                                                       const float* __restrict__ z, const float* __restrict__ bias, int64_t n,
                                                       int d, int dp, int64_t rows_per_block, float* __restrict__ d_em,
                                                       float* __restrict__ d_z, float* __restrict__ part) {
  __shared__ float red[256];
  const int c = threadIdx.x % dp, lane_row = threadIdx.x / dp, step = 256 / dp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  float s = 0.f;
  if (c < d) {
    const float b = bias != nullptr ? bias[c] : 0.f;
    for (int64_t r = r0 + lane_row; r < r1; r += step) {
      const int64_t k = r * d + c;
      const float sg = sigmoid_f32(z[k] + b), gv = g[k], e = em[k];
      d_em[k] = gv * sg;
      const float dz = gv * e * sg * (1.0f - sg);
      d_z[k] = dz;
      s += dz;
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < dp) {
    float t = 0.f;
    for (int k = 0; k < step; ++k) t += red[threadIdx.x + k * dp];
    if (c < d) part[(int64_t)blockIdx.x * d + c] = t;
  }
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LPR lanes per row, float4 per lane (d = 4 LPR in {32, 64, 128, 256}); score [3, n]
template <int LPR>
__global__ __launch_bounds__(256) void channel_mix_fwd_kernel(const float* __restrict__ e1, const float* __restrict__ e2,
                                                              const float* __restrict__ e3, const float* __restrict__ v,
                                                              const float* __restrict__ extra, float extra_scale, int64_t n,
                                                              float* __restrict__ mixed, float* __restrict__ score) {
  constexpr int D = 4 * LPR, GROUPS = 256 / LPR;
  const int gl = threadIdx.x % LPR;
  const float4 v4 = *reinterpret_cast<const float4*>(v + 4 * gl);
  for (int64_t r = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR; r < n; r += (int64_t)gridDim.x * GROUPS) {
    const int64_t k = r * D + 4 * gl;
    const float4 a = *reinterpret_cast<const float4*>(e1 + k), b = *reinterpret_cast<const float4*>(e2 + k),
                 c = *reinterpret_cast<const float4*>(e3 + k);
    const float l1 = group_sum<LPR>(dot4(a, v4)), l2 = group_sum<LPR>(dot4(b, v4)), l3 = group_sum<LPR>(dot4(c, v4));
    const float m = fmaxf(l1, fmaxf(l2, l3));
    const float p1 = expf(l1 - m), p2 = expf(l2 - m), p3 = expf(l3 - m);
    const float inv = 1.0f / (p1 + p2 + p3);
    const float s1 = p1 * inv, s2 = p2 * inv, s3 = p3 * inv;
    float4 o = make_float4(s1 * a.x + s2 * b.x + s3 * c.x, s1 * a.y + s2 * b.y + s3 * c.y, s1 * a.z + s2 * b.z + s3 * c.z,
                           s1 * a.w + s2 * b.w + s3 * c.w);
    if (extra != nullptr) {
      const float4 x = *reinterpret_cast<const float4*>(extra + k);
      o.x += extra_scale * x.x; o.y += extra_scale * x.y; o.z += extra_scale * x.z; o.w += extra_scale * x.w;
    }
    *reinterpret_cast<float4*>(mixed + k) = o;
    if (gl == 0) {
      score[r] = s1;
      score[n + r] = s2;
      score[2 * n + r] = s3;
    }
  }
}

// d e_k = score_k g + dl_k v with dl_k = score_k (<g, e_k> - sum_m score_m <g, e_m>);  d extra = scale g;
// part[block][c] = sum over the block's rows of sum_k dl_k e_k[c]  (the gradient of v)
template <int LPR>
__global__ __launch_bounds__(256) void channel_mix_bwd_kernel(const float* __restrict__ g, const float* __restrict__ e1,
                                                              const float* __restrict__ e2, const float* __restrict__ e3,
                                                              const float* __restrict__ v, const float* __restrict__ score,
                                                              float extra_scale, int64_t n, int64_t rows_per_block,
                                                              float* __restrict__ d1, float* __restrict__ d2,
                                                              float* __restrict__ d3, float* __restrict__ d_extra,
                                                              float* __restrict__ part) {
  constexpr int D = 4 * LPR, GROUPS = 256 / LPR;
  __shared__ float4 red[256];
  const int gl = threadIdx.x % LPR;
  const float4 v4 = *reinterpret_cast<const float4*>(v + 4 * gl);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(n, r0 + rows_per_block);
  float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t r = r0 + threadIdx.x / LPR; r < r1; r += GROUPS) {
    const int64_t k = r * D + 4 * gl;
    const float4 gv = *reinterpret_cast<const float4*>(g + k);
    const float4 a = *reinterpret_cast<const float4*>(e1 + k), b = *reinterpret_cast<const float4*>(e2 + k),
                 c = *reinterpret_cast<const float4*>(e3 + k);
    const float s1 = score[r], s2 = score[n + r], s3 = score[2 * n + r];
    const float a1 = group_sum<LPR>(dot4(gv, a)), a2 = group_sum<LPR>(dot4(gv, b)), a3 = group_sum<LPR>(dot4(gv, c));
    const float abar = s1 * a1 + s2 * a2 + s3 * a3;
    const float q1 = s1 * (a1 - abar), q2 = s2 * (a2 - abar), q3 = s3 * (a3 - abar);
    *reinterpret_cast<float4*>(d1 + k) = make_float4(s1 * gv.x + q1 * v4.x, s1 * gv.y + q1 * v4.y, s1 * gv.z + q1 * v4.z, s1 * gv.w + q1 * v4.w);
    *reinterpret_cast<float4*>(d2 + k) = make_float4(s2 * gv.x + q2 * v4.x, s2 * gv.y + q2 * v4.y, s2 * gv.z + q2 * v4.z, s2 * gv.w + q2 * v4.w);
    *reinterpret_cast<float4*>(d3 + k) = make_float4(s3 * gv.x + q3 * v4.x, s3 * gv.y + q3 * v4.y, s3 * gv.z + q3 * v4.z, s3 * gv.w + q3 * v4.w);
    if (d_extra != nullptr)
      *reinterpret_cast<float4*>(d_extra + k) = make_float4(extra_scale * gv.x, extra_scale * gv.y, extra_scale * gv.z, extra_scale * gv.w);
    dv.x += q1 * a.x + q2 * b.x + q3 * c.x;
    dv.y += q1 * a.y + q2 * b.y + q3 * c.y;
    dv.z += q1 * a.z + q2 * b.z + q3 * c.z;
    dv.w += q1 * a.w + q2 * b.w + q3 * c.w;
  }
  red[threadIdx.x] = dv;
  __syncthreads();
  if (threadIdx.x < LPR) {
    float4 t = red[threadIdx.x];
    for (int k = 1; k < GROUPS; ++k) {
      const float4 u = red[threadIdx.x + k * LPR];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * D + 4 * threadIdx.x) = t;
  }
}

int gram_blocks(int64_t n) {
  const int64_t want = (n + 255) / 256;                      // >= 256 rows per workgroup
  return (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
}

bool gram_dim_ok(int d) { return d == 32 || d == 64 || d == 96 || d == 128; }

}  // namespace

extern "C" int32_t gcr_normalize_bwd_n_f32(const float* n_rows_normalised, const float* inv_norm, const float* g_n,
                                           const float* g_raw, int64_t rows, int32_t d, float* out, void* stream) {
  GCR_CHECK_ARG(rows >= 0 && d >= 4 && d <= 256 && (d & 3) == 0);
  if (rows == 0) return GCR_OK;
  GCR_CHECK_ARG(n_rows_normalised != nullptr && inv_norm != nullptr && g_n != nullptr && out != nullptr);
  const int64_t want = (rows + 15) / 16;
  hipLaunchKernelGGL(normalize_bwd_n_kernel<false>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0,
                     (hipStream_t)stream, n_rows_normalised, inv_norm, g_n, g_raw, rows, d, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_normalize_bwd_raw_f32(const float* z_raw, const float* inv_norm, const float* g_n, const float* g_raw,
                                             int64_t rows, int32_t d, float* out, void* stream) {
  GCR_CHECK_ARG(rows >= 0 && d >= 4 && d <= 256 && (d & 3) == 0);
  if (rows == 0) return GCR_OK;
  GCR_CHECK_ARG(z_raw != nullptr && inv_norm != nullptr && g_n != nullptr && out != nullptr);
  const int64_t want = (rows + 15) / 16;
  hipLaunchKernelGGL(normalize_bwd_n_kernel<true>, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0,
                     (hipStream_t)stream, z_raw, inv_norm, g_n, g_raw, rows, d, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_gram_tn_workspace_bytes(int64_t n, int32_t dx, int32_t dg) {
  if (n <= 0 || !gram_dim_ok(dx) || !gram_dim_ok(dg)) return 0;
  return (int64_t)gram_blocks(n) * dx * dg * (int64_t)sizeof(float);
}

extern "C" int32_t gcr_gram_tn_f32(const float* x, const float* g, int64_t n, int32_t dx, int32_t dg, float* out,
                                   void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0);
  if (!gram_dim_ok(dx) || !gram_dim_ok(dg)) return GCR_EUNSUPPORTED;
  GCR_CHECK_ARG(out != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    hipError_t err = hipMemsetAsync(out, 0, sizeof(float) * (size_t)dx * dg, s);
    return gcr_hip_status(err);
  }
  GCR_CHECK_ARG(x != nullptr && g != nullptr && workspace != nullptr);
  const int nb = gram_blocks(n);
  const int64_t per = ((n + nb - 1) / nb + 1) & ~(int64_t)1;   // even: a row pair never straddles two workgroups
  float* part = reinterpret_cast<float*>(workspace);
#define GCR_GRAM(TX, TG) \
  hipLaunchKernelGGL((gram_tn_kernel<TX, TG>), dim3((unsigned)nb), dim3(256), 0, s, x, g, n, per, part)
#define GCR_GRAM_ROW(TX)                 \
  switch (dg / 32) {                     \
    case 1: GCR_GRAM(TX, 1); break;      \
    case 2: GCR_GRAM(TX, 2); break;      \
    case 3: GCR_GRAM(TX, 3); break;      \
    default: GCR_GRAM(TX, 4); break;     \
  }
  switch (dx / 32) {
    case 1: GCR_GRAM_ROW(1); break;
    case 2: GCR_GRAM_ROW(2); break;
    case 3: GCR_GRAM_ROW(3); break;
    default: GCR_GRAM_ROW(4); break;
  }
#undef GCR_GRAM_ROW
#undef GCR_GRAM
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  const int elems = dx * dg;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, s, part, nb, elems, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_weighted_colsum_workspace_bytes(int64_t n, int32_t d) {
  if (n <= 0 || d < 1 || d > 256) return 0;
  return (int64_t)gram_blocks(n) * d * (int64_t)sizeof(float);
}

extern "C" int32_t gcr_weighted_colsum_f32(const float* x, const float* w, int64_t n, int32_t d, float* out, void* workspace,
                                           void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1 && d <= 256 && out != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return gcr_hip_status(hipMemsetAsync(out, 0, sizeof(float) * (size_t)d, s));
  GCR_CHECK_ARG(x != nullptr && w != nullptr && workspace != nullptr);
  const int nb = gram_blocks(n);
  const int64_t per = (n + nb - 1) / nb;
  int dp = 1;
  while (dp < d) dp <<= 1;
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(weighted_colsum_kernel, dim3((unsigned)nb), dim3(256), 0, s, x, w, n, (int)d, dp, per, part);
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, part, nb, (int)d, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_rows_dot_vec_f32(const float* x, const float* v, int64_t n, int32_t d, float* out, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && v != nullptr && out != nullptr);
  const int64_t want = (n + 15) / 16;
  hipLaunchKernelGGL(rows_dot_vec_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream,
                     x, v, n, (int)d, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_gate_fwd_f32(const float* em, const float* z, const float* bias, int64_t n, int32_t d, float* out,
                                    void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1 && d <= 256);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(em != nullptr && z != nullptr && out != nullptr);
  int dp = 1;
  while (dp < d) dp <<= 1;
  const int64_t want = (n + (256 / dp) * 4 - 1) / ((256 / dp) * 4);
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, (hipStream_t)stream, em, z,
                     bias, n, (int)d, dp, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_gate_bwd_workspace_bytes(int64_t n, int32_t d) {
  if (n <= 0 || d < 1 || d > 256) return 0;
  return (int64_t)gram_blocks(n) * d * (int64_t)sizeof(float);
}

extern "C" int32_t gcr_gate_bwd_f32(const float* g, const float* em, const float* z, const float* bias, int64_t n, int32_t d,
                                    float* d_em, float* d_z, float* d_bias, void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1 && d <= 256 && d_bias != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return gcr_hip_status(hipMemsetAsync(d_bias, 0, sizeof(float) * (size_t)d, s));
  GCR_CHECK_ARG(g != nullptr && em != nullptr && z != nullptr && d_em != nullptr && d_z != nullptr && workspace != nullptr);
  const int nb = gram_blocks(n);
  const int64_t per = (n + nb - 1) / nb;
  int dp = 1;
  while (dp < d) dp <<= 1;
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, s, g, em, z, bias, n, (int)d, dp, per, d_em, d_z, part);
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, part, nb, (int)d, d_bias);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_channel_mix_supported(int32_t d) { return d == 32 || d == 64 || d == 128 || d == 256; }

extern "C" int32_t gcr_channel_mix_fwd_f32(const float* e1, const float* e2, const float* e3, const float* v, const float* extra,
                                           float extra_scale, int64_t n, int32_t d, float* mixed, float* score, void* stream) {
  GCR_CHECK_ARG(n >= 0);
  if (!gcr_channel_mix_supported(d)) return GCR_EUNSUPPORTED;
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(e1 && e2 && e3 && v && mixed && score);
  hipStream_t s = (hipStream_t)stream;
  const int groups = 256 / (d / 4);
  const int64_t want = (n + groups * 2 - 1) / (groups * 2);
  const dim3 grid((unsigned)(want > 16384 ? 16384 : want));
#define GCR_MIXF(LPR) \
  hipLaunchKernelGGL((channel_mix_fwd_kernel<LPR>), grid, dim3(256), 0, s, e1, e2, e3, v, extra, extra_scale, n, mixed, score)
  switch (d) {
    case 32: GCR_MIXF(8); break;
    case 64: GCR_MIXF(16); break;
    case 128: GCR_MIXF(32); break;
    default: GCR_MIXF(64); break;
  }
#undef GCR_MIXF
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_channel_mix_bwd_workspace_bytes(int64_t n, int32_t d) {
  if (n <= 0 || !gcr_channel_mix_supported(d)) return 0;
  return (int64_t)gram_blocks(n) * d * (int64_t)sizeof(float);
}

extern "C" int32_t gcr_channel_mix_bwd_f32(const float* g, const float* e1, const float* e2, const float* e3, const float* v,
                                           const float* score, float extra_scale, int64_t n, int32_t d, float* d_e1, float* d_e2,
                                           float* d_e3, float* d_extra, float* d_v, void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d_v != nullptr);
  if (!gcr_channel_mix_supported(d)) return GCR_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return gcr_hip_status(hipMemsetAsync(d_v, 0, sizeof(float) * (size_t)d, s));
  GCR_CHECK_ARG(g && e1 && e2 && e3 && v && score && d_e1 && d_e2 && d_e3 && workspace);
  const int nb = gram_blocks(n);
  const int64_t per = (n + nb - 1) / nb;
  float* part = reinterpret_cast<float*>(workspace);
#define GCR_MIXB(LPR)                                                                                                       \
  hipLaunchKernelGGL((channel_mix_bwd_kernel<LPR>), dim3((unsigned)nb), dim3(256), 0, s, g, e1, e2, e3, v, score, extra_scale, \
                     n, per, d_e1, d_e2, d_e3, d_extra, part)
  switch (d) {
    case 32: GCR_MIXB(8); break;
    case 64: GCR_MIXB(16); break;
    case 128: GCR_MIXB(32); break;
    default: GCR_MIXB(64); break;
  }
#undef GCR_MIXB
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((unsigned)((d + 63) / 64)), dim3(256), 0, s, part, nb, (int)d, d_v);
  return GCR_LAUNCH_STATUS();
}
