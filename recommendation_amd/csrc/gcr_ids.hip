// Raw id -> dense id maps on the device (the id maps of graph ingest).
//
// Replaces the Python sets / dicts of
//   Interaction._build            ncl.py:55-66 (= directau.py:111-122, sept.py:117-128): dense id = rank of the raw id in
//                                 SORTED order (`sorted(users)`)
//   Interaction.__generate_set    selfcf.py:279-288 (= ssl4rec.py:69-75): dense id = order of FIRST APPEARANCE in the
//                                 training file (dict insertion order)
// The raw ids arrive as 64-bit keys whose unsigned order is the order of the ids (the host encodes a column of id strings
// as big-endian bytes, 8 per key word; longer ids are folded word by word through the sorted form — encoders.py).  One
// stable LSD radix sort of (key, position), a head flag per run of equal keys and a scan give the sorted-rank ids; the
// first-appearance ids re-rank the runs by the position of their head (the stable sort leaves every run's earliest
// position at its head).  Integer work, bit-exact with the reference's maps (tests/golden/graph_build.npz).
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gcr_common.h"

namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n) {
  const int64_t g = (n + kBlock - 1) / kBlock;
  return (unsigned)(g < 1 ? 1 : (g > 262144 ? 262144 : g));
}
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

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
  uint64_t* k = nullptr;
  int64_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 64, (hipStream_t)0);
  return bytes;
}
size_t scan_temp_bytes(int64_t n) {
  size_t bytes = 0;
  int64_t* v = nullptr;
  (void)rocprim::inclusive_scan(nullptr, bytes, v, v, (size_t)n, rocprim::plus<int64_t>(), (hipStream_t)0);
  return bytes;
}

__global__ void iota_kernel(int64_t n, int64_t* __restrict__ idx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) idx[e] = e;
}

// head[e] = 1 where a run of equal keys starts
__global__ void head_kernel(const uint64_t* __restrict__ ks, int64_t n, int64_t* __restrict__ head) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    head[e] = (e == 0 || ks[e] != ks[e - 1]) ? 1 : 0;
}

// sorted-rank ids: run number = inclusive scan of the head flags - 1
__global__ void scatter_sorted_kernel(const uint64_t* __restrict__ ks, const int64_t* __restrict__ is,
                                      const int64_t* __restrict__ run, int64_t n, int64_t* __restrict__ dense,
                                      int64_t* __restrict__ first_pos, int64_t* __restrict__ n_unique) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = run[e] - 1;
    dense[is[e]] = r;
    if (e == 0 || ks[e] != ks[e - 1]) first_pos[r] = is[e];      // stable sort: the head is the earliest occurrence
    if (e == n - 1) *n_unique = r + 1;
  }
}

// first-appearance ids, step 1: key = head ? position of the head : +inf, value = run number
__global__ void head_pos_kernel(const uint64_t* __restrict__ ks, const int64_t* __restrict__ is,
                                const int64_t* __restrict__ run, int64_t n, uint64_t* __restrict__ key,
                                int64_t* __restrict__ val, int64_t* __restrict__ n_unique) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const bool head = e == 0 || ks[e] != ks[e - 1];
    key[e] = head ? (uint64_t)is[e] : ~0ull;
    val[e] = run[e] - 1;
    if (e == n - 1) *n_unique = run[e];
  }
}

// step 2: the j-th smallest head position belongs to run val[j]: that run gets id j
__global__ void rank_runs_kernel(const uint64_t* __restrict__ key_sorted, const int64_t* __restrict__ val_sorted,
                                 int64_t n, int64_t* __restrict__ id_of_run, int64_t* __restrict__ first_pos) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    if (key_sorted[j] == ~0ull) continue;
    id_of_run[val_sorted[j]] = j;
    first_pos[j] = (int64_t)key_sorted[j];
  }
}

__global__ void scatter_first_seen_kernel(const int64_t* __restrict__ is, const int64_t* __restrict__ run,
                                          const int64_t* __restrict__ id_of_run, int64_t n, int64_t* __restrict__ dense) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    dense[is[e]] = id_of_run[run[e] - 1];
}

}  // namespace

extern "C" int64_t gcr_dense_ids_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const size_t m = (size_t)(n > 0 ? n : 1);
  return (int64_t)(7 * align256(m * 8) + align256(sort_temp_bytes(n)) + align256(scan_temp_bytes(n)) + 1024);
}

extern "C" int32_t gcr_dense_ids_u64(const uint64_t* keys, int64_t n, int32_t order, int64_t* dense, int64_t* first_pos,
                                     int64_t* n_unique, void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0 && (order == 0 || order == 1) && n_unique != nullptr);
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) return gcr_hip_status(hipMemsetAsync(n_unique, 0, sizeof(int64_t), s));
  GCR_CHECK_ARG(keys && dense && first_pos && workspace);
  Carve ws{reinterpret_cast<char*>(workspace)};
  uint64_t* ks = ws.take<uint64_t>(n);
  int64_t* idx = ws.take<int64_t>(n);
  int64_t* is = ws.take<int64_t>(n);
  int64_t* run = ws.take<int64_t>(n);
  uint64_t* k2 = ws.take<uint64_t>(n);
  int64_t* v2 = ws.take<int64_t>(n);
  int64_t* id_of_run = ws.take<int64_t>(n);
  size_t sort_bytes = sort_temp_bytes(n);
  void* sort_tmp = ws.take<char>(sort_bytes);
  size_t scan_bytes = scan_temp_bytes(n);
  void* scan_tmp = ws.take<char>(scan_bytes);

  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, n, idx);
  hipError_t err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, ks, idx, is, (size_t)n, 0, 64, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  hipLaunchKernelGGL(head_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, ks, n, run);
  err = rocprim::inclusive_scan(scan_tmp, scan_bytes, run, run, (size_t)n, rocprim::plus<int64_t>(), s);
  if (err != hipSuccess) return gcr_hip_status(err);
  if (order == 0) {
    hipLaunchKernelGGL(scatter_sorted_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, ks, is, run, n, dense, first_pos,
                       n_unique);
    return GCR_LAUNCH_STATUS();
  }
  // the second sort's outputs reuse ks / idx (the sorted keys are only compared inside head_pos_kernel, before it)
  hipLaunchKernelGGL(head_pos_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, ks, is, run, n, k2, v2, n_unique);
  err = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, k2, ks, v2, idx, (size_t)n, 0, 64, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  hipLaunchKernelGGL(rank_runs_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, ks, idx, n, id_of_run, first_pos);
  hipLaunchKernelGGL(scatter_first_seen_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, is, run, id_of_run, n, dense);
  return GCR_LAUNCH_STATUS();
}
