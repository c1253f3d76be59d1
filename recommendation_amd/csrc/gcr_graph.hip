// Graph ingest on the device: COO -> CSR (stable by-row order or (row,col)-coalesced), degree
// normalisation of the values, exact-count edge dropout.  Integer work, bit-exact with the
// reference's host-side construction:
//   Interaction._build_adj + convert_sparse_mat_to_tensor   ncl.py:74-85,203-209  (COO order kept
//       inside a row, duplicates kept: torch.sparse.mm on the uncoalesced COO sums them)
//   csr_matrix((data,(u,i))) ; adj + adj.T ; normalize_graph_mat   selfcf.py:291-306,240-255,
//       ssl4rec.py:79-88  (sorted by (row, col), duplicates summed, D^-1/2 A D^-1/2, inf -> 0)
//   gcn_norm(add_self_loops=False) of LGConv               lightgcn.py:17,25
//   GraphAugmentor.edge_dropout                             univariate/sept.py:55-61 (keeps exactly
//       floor(nnz * (1 - p)) entries, without replacement)
// Sorting / scanning / run-length reduction use rocPRIM's device primitives (stable LSD radix
// sort), the rest are small streaming kernels.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>

#include "gcr_common.h"
#include "gcr_philox.h"

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n) {
  const int64_t g = (n + kBlock - 1) / kBlock;
  return (unsigned)(g < 1 ? 1 : (g > 262144 ? 262144 : g));
}

inline int bits_for(uint64_t max_value) {
  int b = 1;
  while (b < 64 && (max_value >> b) != 0) ++b;
  return b;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__global__ void iota_and_check_kernel(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t nnz,
                                      int64_t n_rows, int64_t n_cols, int64_t mul, int64_t* __restrict__ key,
                                      int64_t* __restrict__ idx, unsigned long long* __restrict__ n_errors) {
  unsigned long long bad = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = row[e], c = col[e];
    const bool ok = r >= 0 && r < n_rows && c >= 0 && c < n_cols;
    bad += !ok;
    // invalid entries sort to the end (key = max) and are dropped by the fill kernels
    key[e] = ok ? (mul > 0 ? r * mul + c : r) : (mul > 0 ? n_rows * mul : n_rows);
    idx[e] = e;
  }
  if (bad) atomicAdd(n_errors, bad);
}

// rowptr[r] = first sorted position whose row is >= r   (rows of sorted keys: key / div)
__global__ void rowptr_fill_kernel(const int64_t* __restrict__ key, int64_t n, int64_t div, int64_t n_rows,
                                   int64_t* __restrict__ rowptr) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = e > 0 ? key[e - 1] / div : -1;
    int64_t cur = e < n ? key[e] / div : n_rows;
    if (prev >= n_rows) continue;               // inside the tail of invalid entries
    if (cur > n_rows) cur = n_rows;             // first invalid entry closes every remaining row
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = e;
  }
}

__global__ void gather_kernel(const int64_t* __restrict__ perm, const int64_t* __restrict__ col,
                              const float* __restrict__ val, int64_t n, int32_t* __restrict__ col_out,
                              float* __restrict__ val_out, int64_t* __restrict__ perm_out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = perm[e];
    col_out[e] = (int32_t)col[p];
    if (val_out != nullptr) val_out[e] = val != nullptr ? val[p] : 1.0f;
    if (perm_out != nullptr) perm_out[e] = p;
  }
}

__global__ void gather_val_kernel(const int64_t* __restrict__ perm, const float* __restrict__ val, int64_t n,
                                  float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = val != nullptr ? val[perm[e]] : 1.0f;
}

__global__ void split_key_kernel(const int64_t* __restrict__ ukey, const int64_t* __restrict__ n_unique, int64_t mul,
                                 int64_t max_valid, int32_t* __restrict__ col_out, int64_t* __restrict__ nnz_out) {
  const int64_t n = *n_unique;
  // unique keys are sorted: invalid ones (== max_valid) can only be the last
  const int64_t n_valid = (n > 0 && ukey[n - 1] >= max_valid) ? n - 1 : n;
  if (blockIdx.x == 0 && threadIdx.x == 0) *nnz_out = n_valid;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_valid; e += (int64_t)gridDim.x * blockDim.x)
    col_out[e] = (int32_t)(ukey[e] % mul);
}

// rowptr for the coalesced case: number of valid unique keys is on the device
__global__ void rowptr_fill_dev_n_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ n_dev,
                                         int64_t div, int64_t n_rows, int64_t* __restrict__ rowptr) {
  const int64_t n = *n_dev;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = e > 0 ? key[e - 1] / div : -1;
    const int64_t cur = e < n ? key[e] / div : n_rows;
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = e;
  }
}

// one thread per row: dinv[r] = (sum of the row's values)^-1/2, inf -> 0   (selfcf.py:243-245)
// (one 16-lane group per row with a fixed-shape tree sum: a thread per row spent 3.6 ms on the
// 50K-entry hub rows of cfg2; row sums of the 0/1/2-valued adjacency are exact in any order)
__global__ void row_dinv_kernel(const int64_t* __restrict__ rowptr, const float* __restrict__ val, int64_t n_rows,
                                float* __restrict__ dinv, bool inverse_not_rsqrt) {
  const int l16 = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * (blockDim.x / 16) + (threadIdx.x >> 4); r < n_rows;
       r += (int64_t)gridDim.x * (blockDim.x / 16)) {
    float s = 0.f;
    if (val != nullptr) {
      for (int64_t e = rowptr[r] + l16; e < rowptr[r + 1]; e += 16) s += val[e];
      s += __shfl_xor(s, 8, 16);
      s += __shfl_xor(s, 4, 16);
      s += __shfl_xor(s, 2, 16);
      s += __shfl_xor(s, 1, 16);
    } else {
      s = (float)(rowptr[r + 1] - rowptr[r]);
    }
    // s == 0 -> inf -> 0, like np.power(0, -0.5 | -1) followed by `d_inv[np.isinf(d_inv)] = 0`
    const float d = inverse_not_rsqrt ? 1.0f / s : 1.0f / sqrtf(s);
    if (l16 == 0) dinv[r] = isinf(d) ? 0.f : d;
  }
}

// val_out[e] = dinv_row[row(e)] * val[e] * dinv_col[col[e]]   (selfcf.py:246-249; lightgcn gcn_norm)
__global__ void scale_values_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                    const float* __restrict__ val, const float* __restrict__ dinv_row,
                                    const float* __restrict__ dinv_col, int64_t n_rows, float* __restrict__ val_out) {
  // one thread per non-zero; its row by binary search in rowptr (a thread per row spent 12.5 ms on
  // cfg2's hub rows)
  const int64_t nnz = rowptr[n_rows];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_rows;            // largest r with rowptr[r] <= e
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (rowptr[mid] <= e) lo = mid;
      else hi = mid;
    }
    val_out[e] = dinv_row[lo] * (val != nullptr ? val[e] : 1.0f) * (dinv_col != nullptr ? dinv_col[col[e]] : 1.0f);
  }
}

// 64-bit random key per edge (ties broken by the stable sort): ctr = (e, 1, 'EDGE')
__global__ void random_keys_kernel(int64_t nnz, uint64_t seed, uint64_t* __restrict__ key, int64_t* __restrict__ idx) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * blockDim.x) {
    const U4 r = philox4x32_10(U4{(uint32_t)e, (uint32_t)((uint64_t)e >> 32), 1u, 0x45444745u}, k0, k1);
    key[e] = ((uint64_t)r.x << 32) | r.y;
    idx[e] = e;
  }
}

__global__ void set_bits_kernel(const int64_t* __restrict__ idx, int64_t n_keep, uint32_t* __restrict__ bits) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_keep; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx[k];
    atomicOr(bits + (e >> 5), 1u << (e & 31));
  }
}

struct Carve {
  char* base;
  size_t off = 0;
  template <class T>
  T* take(size_t n) {
    T* p = reinterpret_cast<T*>(base + off);
    off += align256(n * sizeof(T));
    return p;
  }
};

size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t* k = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)n, 0, 64, (hipStream_t)0);
  return bytes;
}

size_t rbk_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t* k = nullptr;
  float* v = nullptr;
  (void)rocprim::reduce_by_key(nullptr, bytes, k, v, (size_t)n, k, v, k, rocprim::plus<float>(), rocprim::equal_to<int64_t>(),
                         (hipStream_t)0);
  return bytes;
}

}  // namespace

extern "C" int64_t gcr_coo_to_csr_workspace_bytes(int64_t nnz) {
  if (nnz < 0) return 0;
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  // keys in/out, idx in/out, sorted values, unique count, sort temp, reduce-by-key temp
  return (int64_t)(4 * align256(n * 8) + 2 * align256(n * 4) + 256 + align256(sort_temp_bytes(nnz)) +
                   align256(rbk_temp_bytes(nnz)) + 1024);
}

extern "C" int32_t gcr_coo_to_csr(const int64_t* row, const int64_t* col, const float* val, int64_t nnz,
                                  int64_t n_rows, int64_t n_cols, int32_t coalesce, int64_t* rowptr, int32_t* col_out,
                                  float* val_out, int64_t* perm_out, int64_t* nnz_out, int64_t* n_errors,
                                  void* workspace, void* stream) {
  GCR_CHECK_ARG(nnz >= 0 && n_rows >= 0 && n_cols >= 0 && n_rows < (1ll << 31) && n_cols < (1ll << 31));
  GCR_CHECK_ARG(rowptr != nullptr && nnz_out != nullptr && n_errors != nullptr);
  GCR_CHECK_ARG(nnz == 0 || (row && col && col_out && workspace));
  GCR_CHECK_ARG(!coalesce || val_out != nullptr);
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(n_errors, 0, sizeof(int64_t), s);
  if (err != hipSuccess) return gcr_hip_status(err);
  if (nnz == 0) {
    err = hipMemsetAsync(rowptr, 0, sizeof(int64_t) * (size_t)(n_rows + 1), s);
    if (err == hipSuccess) err = hipMemsetAsync(nnz_out, 0, sizeof(int64_t), s);
    return gcr_hip_status(err);
  }
  Carve ws{reinterpret_cast<char*>(workspace)};
  int64_t* key_in = ws.take<int64_t>(nnz);
  int64_t* key_out = ws.take<int64_t>(nnz);
  int64_t* idx_in = ws.take<int64_t>(nnz);
  int64_t* idx_out = ws.take<int64_t>(nnz);
  float* sval = ws.take<float>(nnz);
  float* uval = ws.take<float>(nnz);
  int64_t* n_unique = ws.take<int64_t>(1);
  size_t sort_bytes = sort_temp_bytes(nnz);
  void* sort_tmp = ws.take<char>(sort_bytes);
  size_t rbk_bytes = rbk_temp_bytes(nnz);
  void* rbk_tmp = ws.take<char>(rbk_bytes);

  const int64_t mul = coalesce ? n_cols : 0;
  const int64_t max_key = coalesce ? n_rows * n_cols : n_rows;
  hipLaunchKernelGGL(iota_and_check_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, row, col, nnz, n_rows, n_cols, mul,
                     key_in, idx_in, (unsigned long long*)n_errors);
  err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, key_in, key_out, idx_in, idx_out, (size_t)nnz, 0,
                                  bits_for((uint64_t)max_key), s);
  if (err != hipSuccess) return gcr_hip_status(err);
  if (!coalesce) {
    hipLaunchKernelGGL(rowptr_fill_kernel, dim3(grid_for(nnz + 1)), dim3(kBlock), 0, s, key_out, nnz, (int64_t)1, n_rows,
                       rowptr);
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_out, col, val, nnz, col_out, val_out,
                       perm_out);
    // entries with an invalid id sorted to the tail: rowptr[n_rows] is the count of valid ones
    err = hipMemcpyAsync(nnz_out, rowptr + n_rows, sizeof(int64_t), hipMemcpyDeviceToDevice, s);
    if (err != hipSuccess) return gcr_hip_status(err);
    return GCR_LAUNCH_STATUS();
  }
  hipLaunchKernelGGL(gather_val_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, idx_out, val, nnz, sval);
  // unique (row, col) keys -> key_in (reused), summed values -> uval
  err = rocprim::reduce_by_key(rbk_tmp, rbk_bytes, key_out, sval, (size_t)nnz, key_in, uval, n_unique,
                               rocprim::plus<float>(), rocprim::equal_to<int64_t>(), s);
  if (err != hipSuccess) return gcr_hip_status(err);
  hipLaunchKernelGGL(split_key_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, key_in, n_unique, n_cols, max_key,
                     col_out, nnz_out);
  hipLaunchKernelGGL(rowptr_fill_dev_n_kernel, dim3(grid_for(nnz + 1)), dim3(kBlock), 0, s, key_in, nnz_out, n_cols,
                     n_rows, rowptr);
  err = hipMemcpyAsync(val_out, uval, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToDevice, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_csr_sym_norm_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                        int64_t n_cols, const int64_t* rowptr_t, const float* val_t, float* dinv_row,
                                        float* dinv_col, float* val_out, void* stream) {
  GCR_CHECK_ARG(n_rows >= 0 && n_cols >= 0);
  if (n_rows == 0) return GCR_OK;
  GCR_CHECK_ARG(rowptr && col && dinv_row && val_out);
  GCR_CHECK_ARG(rowptr_t == nullptr || dinv_col != nullptr);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(row_dinv_kernel, dim3(grid_for(n_rows * 16)), dim3(kBlock), 0, s, rowptr, val, n_rows, dinv_row, false);
  const float* dcol = dinv_row;   // square symmetric operator: column scale = row scale
  if (rowptr_t != nullptr) {      // rectangular / asymmetric: column sums come from the transposed CSR
    hipLaunchKernelGGL(row_dinv_kernel, dim3(grid_for(n_cols * 16)), dim3(kBlock), 0, s, rowptr_t, val_t, n_cols, dinv_col, false);
    dcol = dinv_col;
  }
  hipLaunchKernelGGL(scale_values_kernel, dim3(16384), dim3(kBlock), 0, s, rowptr, col, val, dinv_row, dcol,
                     n_rows, val_out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_csr_row_norm_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                                        float* rinv, float* val_out, void* stream) {
  GCR_CHECK_ARG(n_rows >= 0);
  if (n_rows == 0) return GCR_OK;
  GCR_CHECK_ARG(rowptr && col && rinv && val_out);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(row_dinv_kernel, dim3(grid_for(n_rows * 16)), dim3(kBlock), 0, s, rowptr, val, n_rows, rinv, true);
  hipLaunchKernelGGL(scale_values_kernel, dim3(16384), dim3(kBlock), 0, s, rowptr, col, val, rinv,
                     (const float*)nullptr, n_rows, val_out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_edge_mask_exact_workspace_bytes(int64_t nnz) {
  if (nnz <= 0) return 256;
  return (int64_t)(4 * align256((size_t)nnz * 8) + align256(sort_temp_bytes(nnz)) + 1024);
}

extern "C" int32_t gcr_edge_mask_exact_bits(int64_t nnz, int64_t n_keep, uint64_t seed, uint32_t* bits,
                                            void* workspace, void* stream) {
  GCR_CHECK_ARG(nnz >= 0 && n_keep >= 0 && n_keep <= nnz);
  if (nnz == 0) return GCR_OK;
  GCR_CHECK_ARG(bits != nullptr && workspace != nullptr);
  hipStream_t s = (hipStream_t)stream;
  Carve ws{reinterpret_cast<char*>(workspace)};
  uint64_t* key_in = ws.take<uint64_t>(nnz);
  uint64_t* key_out = ws.take<uint64_t>(nnz);
  int64_t* idx_in = ws.take<int64_t>(nnz);
  int64_t* idx_out = ws.take<int64_t>(nnz);
  size_t sort_bytes = sort_temp_bytes(nnz);
  void* sort_tmp = ws.take<char>(sort_bytes);
  hipError_t err = hipMemsetAsync(bits, 0, sizeof(uint32_t) * (size_t)((nnz + 31) / 32), s);
  if (err != hipSuccess) return gcr_hip_status(err);
  hipLaunchKernelGGL(random_keys_kernel, dim3(grid_for(nnz)), dim3(kBlock), 0, s, nnz, seed, key_in, idx_in);
  err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, key_in, key_out, idx_in, idx_out, (size_t)nnz, 0, 64, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  if (n_keep > 0)
    hipLaunchKernelGGL(set_bits_kernel, dim3(grid_for(n_keep)), dim3(kBlock), 0, s, idx_out, n_keep, bits);
  return GCR_LAUNCH_STATUS();
}
