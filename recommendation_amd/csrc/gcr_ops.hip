// The stages either side of the hot path as streaming kernels:
//   adam_step          fused dense Adam on an embedding table (torch.optim.Adam of ncl.py:305, lightgcn.py:84,
//                      gcl.py:201 incl. its L2 weight_decay), up to three gradient pieces summed on the way in
//   mask_columns       PyGCL FeatureMasking / drop_feature (univariate/grace.py:261-278): whole feature columns zeroed
//   spgemm_expand      sparse x sparse products of MHCN's motif adjacency (univariate/mhcn.py:340-368), expand step
//                      of expand-sort-compress (the sort + duplicate sum is gcr_coo_to_csr with coalesce = 1)
//   csr_lookup         element-wise product with a sparse mask: value of (r, c) in another CSR (0 when absent)
// All HBM-bound: 16-B accesses, one pass.
#include "gcr_common.h"

namespace {

__global__ __launch_bounds__(256) void adam_step_kernel(float4* __restrict__ p, const float4* __restrict__ g1,
                                                        const float4* __restrict__ g2, const float4* __restrict__ g3,
                                                        float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                        int tail, float lr_over_bc1, float beta1, float beta2, float eps,
                                                        float inv_sqrt_bc2, float weight_decay, float grad_scale,
                                                        const int64_t* __restrict__ step_dev, float lr) {
  const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
  if (step_dev != nullptr) {      // step count kept on the device (a captured graph replays with the CURRENT count)
    const double st = (double)*step_dev;
    lr_over_bc1 = (float)((double)lr / (1.0 - pow((double)beta1, st)));
    inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)beta2, st)));
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp = p[i], gg = g1[i], mm = m[i], vv = v[i];
    if (g2 != nullptr) {
      const float4 t = g2[i];
      gg.x += t.x; gg.y += t.y; gg.z += t.z; gg.w += t.w;
    }
    if (g3 != nullptr) {
      const float4 t = g3[i];
      gg.x += t.x; gg.y += t.y; gg.z += t.z; gg.w += t.w;
    }
    float pe[4] = {pp.x, pp.y, pp.z, pp.w}, ge[4] = {gg.x, gg.y, gg.z, gg.w};
    float me[4] = {mm.x, mm.y, mm.z, mm.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g = fmaf(weight_decay, pe[e], ge[e] * grad_scale);        // torch Adam: grad + weight_decay * param
      me[e] = fmaf(omb1, g - me[e], me[e]);                                  // m += (1 - b1) (g - m)   (torch's lerp)
      ve[e] = fmaf(beta2, ve[e], omb2 * g * g);                              // v = b2 v + (1 - b2) g^2
      const float denom = sqrtf(ve[e]) * inv_sqrt_bc2 + eps;
      pe[e] -= lr_over_bc1 * (me[e] / denom);
    }
    p[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    m[i] = make_float4(me[0], me[1], me[2], me[3]);
    v[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) {        // the last n % 4 elements, one per thread
    const int64_t i = 4 * n4 + threadIdx.x;
    float* ps = reinterpret_cast<float*>(p);
    float* ms = reinterpret_cast<float*>(m);
    float* vs = reinterpret_cast<float*>(v);
    float g = reinterpret_cast<const float*>(g1)[i];
    if (g2 != nullptr) g += reinterpret_cast<const float*>(g2)[i];
    if (g3 != nullptr) g += reinterpret_cast<const float*>(g3)[i];
    g = fmaf(weight_decay, ps[i], g * grad_scale);
    const float mm = fmaf(omb1, g - ms[i], ms[i]);
    const float vv = fmaf(beta2, vs[i], omb2 * g * g);
    ps[i] -= lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
    ms[i] = mm;
    vs[i] = vv;
  }
}

__global__ __launch_bounds__(256) void mask_columns_kernel(const float* __restrict__ x, int64_t n, int d,
                                                           const uint32_t* __restrict__ keep_bits, float* __restrict__ out) {
  const int64_t total = n * (int64_t)d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    out[i] = ((keep_bits[c >> 5] >> (c & 31)) & 1u) ? x[i] : 0.f;
  }
}

// one 16-lane group per non-zero e = (i, k) of A: the entries (k, j) of B's row k go to out[offset[e] ...]
__global__ __launch_bounds__(256) void spgemm_expand_kernel(const int64_t* __restrict__ a_rowptr, const int32_t* __restrict__ a_col,
                                                            const float* __restrict__ a_val, int64_t a_rows, int64_t a_nnz,
                                                            const int64_t* __restrict__ b_rowptr, const int32_t* __restrict__ b_col,
                                                            const float* __restrict__ b_val, const int64_t* __restrict__ offset,
                                                            const int32_t* __restrict__ a_row_of, int64_t* __restrict__ out_row,
                                                            int64_t* __restrict__ out_col, float* __restrict__ out_val) {
  const int l16 = threadIdx.x & 15;
  for (int64_t e = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); e < a_nnz; e += (int64_t)gridDim.x * 16) {
    const int32_t k = a_col[e];
    const float av = a_val != nullptr ? a_val[e] : 1.0f;
    const int64_t i = a_row_of[e];
    const int64_t b0 = b_rowptr[k], b1 = b_rowptr[k + 1], o = offset[e];
    for (int64_t f = b0 + l16; f < b1; f += 16) {
      out_row[o + (f - b0)] = i;
      out_col[o + (f - b0)] = b_col[f];
      out_val[o + (f - b0)] = av * (b_val != nullptr ? b_val[f] : 1.0f);
    }
  }
}

// out[e] = value of (row(e), col[e]) in the CSR m (columns ascending inside a row), 0 when it is not stored
__global__ __launch_bounds__(256) void csr_lookup_kernel(const int32_t* __restrict__ row_of, const int32_t* __restrict__ col,
                                                         int64_t nnz, const int64_t* __restrict__ m_rowptr,
                                                         const int32_t* __restrict__ m_col, const float* __restrict__ m_val,
                                                         float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = row_of[e], c = col[e];
    int64_t lo = m_rowptr[r], hi = m_rowptr[r + 1];
    const int64_t end = hi;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (m_col[mid] < c) lo = mid + 1; else hi = mid;
    }
    out[e] = (lo < end && m_col[lo] == c) ? (m_val != nullptr ? m_val[lo] : 1.0f) : 0.f;
  }
}

// out[idx[i], :] += src[i, :]: one wave per source row, 256-B float-atomic row segments (duplicate ids add up; ids
// outside [0, n_rows) are skipped)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                               int64_t n, int d, int64_t n_rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    const int64_t r = idx[i];
    if (r < 0 || r >= n_rows) continue;
    for (int c = lane; c < d; c += 64) atomicAdd(out + r * d + c, src[i * d + c]);
  }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                          int64_t n, int d, int64_t n_rows, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    const int64_t r = idx[i];
    const bool ok = r >= 0 && r < n_rows;
    for (int c = lane; c < d; c += 64) out[i * d + c] = ok ? table[r * d + c] : 0.f;
  }
}

int ops_grid(int64_t n, int per_block) {
  const int64_t g = (n + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int32_t gcr_adam_step_f32(float* param, const float* grad, const float* grad2, const float* grad3,
                                     float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                     float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
  GCR_CHECK_ARG(n >= 0 && step >= 1);
  GCR_CHECK_ARG(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq);
  GCR_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)grad2 | (uintptr_t)grad3 | (uintptr_t)exp_avg |
                  (uintptr_t)exp_avg_sq) & 15) == 0);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3(ops_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                     (const float4*)grad, (const float4*)grad2, (const float4*)grad3, (float4*)exp_avg, (float4*)exp_avg_sq,
                     n / 4, (int)(n & 3), (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), weight_decay, grad_scale,
                     (const int64_t*)nullptr, lr);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_adam_step_dev_f32(float* param, const float* grad, const float* grad2, const float* grad3,
                                         float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                                         float eps, float weight_decay, const int64_t* step_dev, float grad_scale,
                                         void* stream) {
  GCR_CHECK_ARG(n >= 0 && step_dev != nullptr);
  GCR_CHECK_ARG(lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(param && grad && exp_avg && exp_avg_sq);
  GCR_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)grad2 | (uintptr_t)grad3 | (uintptr_t)exp_avg |
                  (uintptr_t)exp_avg_sq) & 15) == 0);
  hipLaunchKernelGGL(adam_step_kernel, dim3(ops_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                     (const float4*)grad, (const float4*)grad2, (const float4*)grad3, (float4*)exp_avg, (float4*)exp_avg_sq,
                     n / 4, (int)(n & 3), 0.f, beta1, beta2, eps, 0.f, weight_decay, grad_scale, step_dev, lr);
  return GCR_LAUNCH_STATUS();
}

__global__ void bitmap_set_kernel(const int64_t* __restrict__ idx, int64_t n, int64_t n_bits, uint32_t* __restrict__ bits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = idx[i];
    if (b >= 0 && b < n_bits) atomicOr(bits + (b >> 5), 1u << (b & 31));
  }
}

extern "C" int32_t gcr_bitmap_set(const int64_t* idx, int64_t n, int64_t n_bits, uint32_t* bits, void* stream) {
  GCR_CHECK_ARG(n >= 0 && n_bits >= 0);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(idx && bits);
  hipLaunchKernelGGL(bitmap_set_kernel, dim3(ops_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, idx, n, n_bits, bits);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_mask_columns_f32(const float* x, int64_t n, int32_t d, const uint32_t* keep_bits, float* out,
                                        void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x && keep_bits && out);
  hipLaunchKernelGGL(mask_columns_kernel, dim3(ops_grid(n * d, 256 * 4)), dim3(256), 0, (hipStream_t)stream, x, n, d,
                     keep_bits, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_spgemm_expand_f32(const int64_t* a_rowptr, const int32_t* a_col, const float* a_val, int64_t a_rows,
                                         int64_t a_nnz, const int32_t* a_row_of, const int64_t* b_rowptr,
                                         const int32_t* b_col, const float* b_val, const int64_t* offset, int64_t* out_row,
                                         int64_t* out_col, float* out_val, void* stream) {
  GCR_CHECK_ARG(a_rows >= 0 && a_nnz >= 0);
  if (a_nnz == 0) return GCR_OK;
  GCR_CHECK_ARG(a_rowptr && a_col && a_row_of && b_rowptr && b_col && offset && out_row && out_col && out_val);
  hipLaunchKernelGGL(spgemm_expand_kernel, dim3(ops_grid(a_nnz, 16)), dim3(256), 0, (hipStream_t)stream, a_rowptr, a_col,
                     a_val, a_rows, a_nnz, b_rowptr, b_col, b_val, offset, a_row_of, out_row, out_col, out_val);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_csr_lookup_f32(const int32_t* row_of, const int32_t* col, int64_t nnz, const int64_t* m_rowptr,
                                      const int32_t* m_col, const float* m_val, float* out, void* stream) {
  GCR_CHECK_ARG(nnz >= 0);
  if (nnz == 0) return GCR_OK;
  GCR_CHECK_ARG(row_of && col && m_rowptr && m_col && out);
  hipLaunchKernelGGL(csr_lookup_kernel, dim3(ops_grid(nnz, 256)), dim3(256), 0, (hipStream_t)stream, row_of, col, nnz,
                     m_rowptr, m_col, m_val, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_gather_rows_f32(const float* table, const int64_t* idx, int64_t n, int32_t d, int64_t n_rows,
                                       float* out, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1 && n_rows >= 0);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(table && idx && out);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ops_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, table, idx, n, d, n_rows,
                     out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_scatter_add_rows_f32(const float* src, const int64_t* idx, int64_t n, int32_t d, int64_t n_rows,
                                            float* out, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1 && n_rows >= 0);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(src && idx && out);
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(ops_grid(n, 4)), dim3(256), 0, (hipStream_t)stream, src, idx, n, d,
                     n_rows, out);
  return GCR_LAUNCH_STATUS();
}
