// k-means E-step of NCL's prototype contrast (ncl.py:340-356), everything except the nearest-centroid search
// (gcr_kmeans_assign_f32, on the MFMA tile engine of gcr_infonce.hip): the centroid update and the re-seeding of empty
// clusters.  The reference delegates to faiss.Kmeans(d, k).train(x) — an un-vendored dependency, not installed: what is
// restated here is faiss' published Clustering::train loop (Clustering.cpp: compute_centroids + split_clusters),
// PARITY WITH FAISS UNPINNED (no fixture exists), checked against oracle_np.kmeans_lloyd.
//
// Nothing in here reads anything back to the host: an iteration is a fixed launch sequence (hipGraph-capturable), and the
// rare empty-cluster path runs inside a one-block kernel whose common case is one pass over the k counts.
#include "gcr_common.h"
#include "gcr_philox.h"

namespace {

// the three-bf16-plane operand split of the search kernels (gcr_b3.h wants these three names first)
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kTileJ = 32;
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
#include "gcr_b3.h"

// One fragment of the pre-split centroid image of gcr_kmeans_search_image_f32 (layout: kmeans_image_kernel in
// gcr_infonce.hip): the 8 features [8 * col8, + 8) of centroid `row`, three planes.
__device__ __forceinline__ void image_store_fragment(u32x4* __restrict__ image, int d, int64_t row, int col8, const float (&v)[8]) {
  const int kh = d / 2, kc = d / 16;
  const int feat = 8 * col8, h = feat / kh, c = (feat % kh) / 8;
  const int64_t T = row / kTileJ;
  const int lane = 32 * h + (int)(row % kTileJ);
  unsigned q[3][4];
  split3(v[0], v[1], q[0][0], q[1][0], q[2][0]);
  split3(v[2], v[3], q[0][1], q[1][1], q[2][1]);
  split3(v[4], v[5], q[0][2], q[1][2], q[2][2]);
  split3(v[6], v[7], q[0][3], q[1][3], q[2][3]);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) image[((T * kc + c) * 3 + pl) * 64 + lane] = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
}

constexpr uint32_t kStreamSplit = 0x4B4D5350u;  // 'KMSP'

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// sums[c] += x_i (256-B float-atomic rows), counts[c] += 1, one wave per point.  n_copies > 1: block b adds into private
// copy b % n_copies of (sums, counts) — with a few hundred clusters every point of a 77K-point training set would
// otherwise hit one of ~300 rows (the memory-side atomic unit serialises adds to one row); the finalize kernel sums the copies.
__global__ __launch_bounds__(256) void kmeans_accumulate_kernel(const float* __restrict__ x, int64_t n, int d,
                                                                const int64_t* __restrict__ assign, int64_t k,
                                                                float* __restrict__ sums, float* __restrict__ counts,
                                                                int n_copies = 1) {
  const int lane = threadIdx.x & 63;
  sums += (int64_t)(blockIdx.x % n_copies) * k * d;
  counts += (int64_t)(blockIdx.x % n_copies) * k;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * 4) {
    const int64_t c = assign[i];
    if (c < 0 || c >= k) continue;
    for (int col = lane; col < d; col += 64) atomicAdd(sums + c * d + col, x[i * d + col]);
    if (lane == 0) atomicAdd(counts + c, 1.0f);
  }
}

// The same sums from the points ORDERED by cluster (gcr_sort_index of `assign`): a wave walks 64 consecutive
// entries, adds every run of equal cluster ids in registers and issues one row atomic per run and chunk
// (1M x 64 points: 0.42 ms with one atomic row per point).
__global__ __launch_bounds__(256) void kmeans_accumulate_sorted_kernel(const float* __restrict__ x, int64_t n, int d,
                                                                       const uint32_t* __restrict__ keys,
                                                                       const int32_t* __restrict__ perm, int64_t k,
                                                                       float* __restrict__ sums,
                                                                       float* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  const int64_t n_chunks = (n + 63) / 64;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks; chunk += (int64_t)gridDim.x * 4) {
    const int64_t c0 = chunk * 64;
    const int cnt = (int)(n - c0 < 64 ? n - c0 : 64);
    uint32_t my_key = 0xFFFFFFFFu;
    int my_row = 0;
    if (lane < cnt) {
      my_key = keys[c0 + lane];
      my_row = perm[c0 + lane];
      if (my_key >= (uint32_t)k) my_key = 0xFFFFFFFFu;
    }
    uint32_t cur = 0xFFFFFFFFu;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float run = 0.f;
    auto flush = [&]() {
      if (cur != 0xFFFFFFFFu) {
        for (int v = 0; v < 4; ++v) {
          const int c = lane + 64 * v;
          if (c < d) atomicAdd(sums + (int64_t)cur * d + c, acc[v]);
        }
        if (lane == 0) atomicAdd(counts + cur, run);
      }
    };
    constexpr int kGather = 8;
    for (int e0 = 0; e0 < cnt; e0 += kGather) {
      uint32_t key[kGather];
      float row[kGather][4];
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        const int e = e0 + q < cnt ? e0 + q : cnt - 1;
        key[q] = e0 + q < cnt ? (uint32_t)__builtin_amdgcn_readlane((int)my_key, e) : 0xFFFFFFFFu;
        const int64_t r = key[q] != 0xFFFFFFFFu ? (int64_t)__builtin_amdgcn_readlane(my_row, e) : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int c = lane + 64 * v;
          row[q][v] = c < d ? x[r * d + c] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        if (key[q] == 0xFFFFFFFFu) continue;
        if (key[q] != cur) {
          flush();
          cur = key[q];
          run = 0.f;
#pragma unroll
          for (int v = 0; v < 4; ++v) acc[v] = 0.f;
        }
        run += 1.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[v] += row[q][v];
      }
    }
    flush();
  }
}

// centroid = sum / count (empty clusters keep their previous centroid); half_sq = 0.5 ||c||^2
__global__ __launch_bounds__(256) void kmeans_finalize_kernel(const float* __restrict__ sums,
                                                              const float* __restrict__ counts, int64_t k, int d,
                                                              float* __restrict__ cent, float* __restrict__ half_sq) {
  const int l16 = threadIdx.x & 15;
  for (int64_t c = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); c < k; c += (int64_t)gridDim.x * 16) {
    const float cnt = counts != nullptr ? counts[c] : 0.f;
    float ss = 0.f;
    for (int col = l16; col < d; col += 16) {
      float v = cent[c * d + col];
      if (cnt > 0.f) {
        v = sums[c * d + col] / cnt;
        cent[c * d + col] = v;
      }
      ss += v * v;
    }
    ss = group16_sum(ss);
    if (l16 == 0) half_sq[c] = 0.5f * ss;
  }
}


// centroid = sum / count (empty clusters keep their previous centroid); half_sq = 0.5 ||c||^2; the sums are CLEARED on the
// way out (the next iteration's accumulate adds into zeros: no memset launches between iterations)
__global__ __launch_bounds__(256) void kmeans_finalize_clear_kernel(float* __restrict__ sums, float* __restrict__ counts,
                                                                    int64_t k, int d, float* __restrict__ cent,
                                                                    float* __restrict__ half_sq, int n_copies) {
  // one thread per centroid ELEMENT (a thread per row or a 16-lane group per row leaves a k = 300 update on 19
  // workgroups, each walking the copies serially: 18-22 us); d divides 256 (32 / 64 / 128 / 256), so a row never
  // straddles workgroups.  The copies are read with independent loads first and cleared afterwards.
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t idx = (int64_t)blockIdx.x * 256 + tid;
  const int64_t c = idx / d;
  const int col = (int)(idx % d);
  const bool valid = c < k;
  const int64_t kd = k * (int64_t)d;
  float cnt = 0.f, sum = 0.f, v = 0.f;
  if (valid) {
    for (int g = 0; g < n_copies; ++g) cnt += counts[(int64_t)g * k + c];      // copies in a fixed order
    const float* p = sums + c * d + col;
    int g = 0;
    for (; g + 4 <= n_copies; g += 4) {
      const float a0 = p[(int64_t)g * kd], a1 = p[(int64_t)(g + 1) * kd], a2 = p[(int64_t)(g + 2) * kd],
                  a3 = p[(int64_t)(g + 3) * kd];
      sum += a0;
      sum += a1;
      sum += a2;
      sum += a3;
    }
    for (; g < n_copies; ++g) sum += p[(int64_t)g * kd];
    v = cent[c * d + col];
    if (cnt > 0.f) {
      v = sum / cnt;
      cent[c * d + col] = v;
    }
    for (g = 0; g < n_copies; ++g) sums[(int64_t)g * kd + c * d + col] = 0.f;
  }
  float ss = v * v;
  if (d <= 64) {
    for (int off = d >> 1; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);     // butterfly inside the row's d lanes
  } else {
    ss = gcr_wave_sum(ss);
    if (lane == 0) red[tid >> 6] = ss;
    __syncthreads();
    const int w0 = (tid >> 6) / (d >> 6) * (d >> 6);                                 // first wave of this row
    ss = 0.f;
    for (int w = 0; w < (d >> 6); ++w) ss += red[w0 + w];
  }
  if (valid && col == 0) {
    half_sq[c] = 0.5f * ss;
    counts[c] = cnt;                                       // the total, for the split step (which clears it)
    for (int g = 1; g < n_copies; ++g) counts[(int64_t)g * k + c] = 0.f;
  }
}

// centroid = fixed-point sum / (scale * count) from the INCREMENTAL sums of gcr_kmeans_search_image_incr_f32 (nothing is
// cleared: the sums persist across the iterations); an empty cluster keeps its centroid; half_sq = 0.5 |c|^2; the float
// copy of the counts is what the split step reads (and clears).  One thread per centroid element, d divides 256.
__global__ __launch_bounds__(256) void kmeans_finalize_q_kernel(const long long* __restrict__ sums_q,
                                                                const int32_t* __restrict__ counts_i, const float* __restrict__ qscale,
                                                                int64_t k, int d, float* __restrict__ cent, float* __restrict__ half_sq,
                                                                float* __restrict__ counts_f, u32x4* __restrict__ image,
                                                                float* __restrict__ bias, int n_copies) {
  __shared__ float red[4];
  __shared__ float rowbuf[256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t idx = (int64_t)blockIdx.x * 256 + tid;
  const int64_t c = idx / d;
  const int col = (int)(idx % d);
  const bool valid = c < k;
  float v = 0.f;
  int cnt = 0;
  if (valid) {
    long long sum = 0;
    for (int g = 0; g < n_copies; ++g) {                     // integer sums: exact in any order
      cnt += counts_i[(int64_t)g * k + c];
      sum += sums_q[((int64_t)g * k + c) * d + col];
    }
    v = cent[c * d + col];
    if (cnt > 0) {
      v = (float)((double)sum * (double)qscale[1] / (double)cnt);
      cent[c * d + col] = v;
    }
  }
  float ss = v * v;
  if (d <= 64) {
    for (int off = d >> 1; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  } else {
    ss = gcr_wave_sum(ss);
    if (lane == 0) red[tid >> 6] = ss;
    __syncthreads();
    const int w0 = (tid >> 6) / (d >> 6) * (d >> 6);
    ss = 0.f;
    for (int w = 0; w < (d >> 6); ++w) ss += red[w0 + w];
  }
  if (valid && col == 0) {
    half_sq[c] = 0.5f * ss;
    counts_f[c] = (float)cnt;
    if (bias != nullptr) bias[c] = -0.5f * ss;
  }
  if (image != nullptr) {                                  // this row's fragments of the search kernel's operand image
    rowbuf[tid] = v;
    __syncthreads();
    if (valid && (col & 7) == 0) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = rowbuf[tid + e];
      image_store_fragment(image, d, c, col >> 3, f);
    }
  }
}

constexpr int kSplitThreads = 256;    // (1024 at first: beside the backward of the NCL step a sixteen-wave workgroup waited up to 0.6 ms for a CU)

template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, T* sh, Op op) {
  // 1024 threads = 16 waves: wave butterfly, then wave 0 folds the 16 wave results; every thread gets the result
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
  __syncthreads();                                   // sh may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  T r = sh[0];
#pragma unroll
  for (int w = 1; w < kSplitThreads / 64; ++w) r = op(r, sh[w]);
  return r;
}

// faiss Clustering.cpp `split_clusters`: every empty cluster ci (ascending) takes over a copy of the centroid of a cluster
// cj found by walking cj = 0, 1, ..., k-1, 0, ... and accepting cj with probability (size_cj - 1) / (n - k) (so never an
// empty or one-point cluster); the two copies are pushed apart by the symmetric perturbation (1 +- 1/1024) alternating
// over the dimensions; the sizes are split in half.  faiss draws from a generator seeded 1234 inside the call; here trial
// q of the e-th empty cluster of iteration `iter` is the Philox word x of counter (q, e, iter, 'KMSP') under key `seed`
// (oracle_np.kmeans_split_clusters restates it).  A walk of 64 k trials without a hit (probability < e^-60 unless nearly
// every cluster has <= 1 point) falls back to the largest cluster (smallest id among ties) when it has >= 2 points, and
// leaves the centroid alone otherwise.  One block; the common case (no empty cluster) is one pass over the counts.
// The counts are CLEARED on the way out.
__global__ __launch_bounds__(kSplitThreads) void kmeans_split_kernel(float* __restrict__ counts, int64_t k, int d,
                                                                     float* __restrict__ cent, float* __restrict__ half_sq,
                                                                     int64_t n_points, uint64_t seed, uint32_t iter,
                                                                     int32_t* __restrict__ n_split_out,
                                                                     u32x4* __restrict__ image = nullptr,
                                                                     float* __restrict__ bias = nullptr) {
  __shared__ long long sh_ll[kSplitThreads / 64];
  __shared__ float sh_f[kSplitThreads / 64];
  const int tid = threadIdx.x;
  auto min_ll = [](long long a, long long b) { return a < b ? a : b; };
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const double denom = (double)(n_points - k);
  long long last = -1;
  int n_split = 0;
  for (uint32_t e = 0;; ++e) {
    // next empty cluster after `last` (the set of empty clusters is fixed: a split gives ci >= 1 point and leaves cj >= 1)
    long long ci = k;
    for (long long c = last + 1 + tid; c < k; c += kSplitThreads)
      if (counts[c] == 0.f) { ci = c; break; }
    ci = block_reduce<long long>(ci, sh_ll, min_ll);
    if (ci >= k) break;
    last = ci;
    long long hit = -1;
    if (denom > 0.0) {
      const long long max_trials = 64 * (long long)k;
      for (long long q0 = 0; q0 < max_trials && hit < 0; q0 += kSplitThreads) {
        const long long q = q0 + tid;
        long long mine = max_trials;
        if (q < max_trials) {
          const long long cj = q % k;
          const float p = (float)(((double)counts[cj] - 1.0) / denom);
          const U4 r = philox4x32_10(U4{(uint32_t)q, e, iter, kStreamSplit}, k0 + (uint32_t)((uint64_t)q >> 32), k1);
          const float u = (float)(r.x >> 8) * (1.0f / 16777216.0f);
          if (u < p) mine = q;
        }
        const long long first = block_reduce<long long>(mine, sh_ll, min_ll);
        if (first < max_trials) hit = first % k;
      }
    }
    if (hit < 0) {                      // fallback: the largest cluster, if it can be split at all
      float best = -1.f;
      long long arg = k;
      for (long long c = tid; c < k; c += kSplitThreads)
        if (counts[c] > best) { best = counts[c]; arg = c; }
      const float bmax = block_reduce<float>(best, sh_f, [](float a, float b) { return a > b ? a : b; });
      const long long barg = block_reduce<long long>(best == bmax ? arg : (long long)k, sh_ll, min_ll);
      if (bmax >= 2.f) hit = barg;
    }
    if (hit >= 0) {
      const long long cj = hit;
      float ss_i = 0.f, ss_j = 0.f;
      for (int col = tid; col < d; col += kSplitThreads) {
        const float c0 = cent[cj * d + col];
        const float up = c0 * (1.0f + 1.0f / 1024.0f), dn = c0 * (1.0f - 1.0f / 1024.0f);
        const float vi = (col & 1) == 0 ? up : dn, vj = (col & 1) == 0 ? dn : up;
        cent[ci * d + col] = vi;
        cent[cj * d + col] = vj;
        ss_i += vi * vi;
        ss_j += vj * vj;
      }
      auto add_f = [](float a, float b) { return a + b; };
      ss_i = block_reduce<float>(ss_i, sh_f, add_f);
      ss_j = block_reduce<float>(ss_j, sh_f, add_f);
      if (image != nullptr) {                                // the two rewritten rows in the search kernel's operand image
        __syncthreads();                                     // (this block's own stores to cent above)
        if (tid < 2 * (d / 8)) {
          const long long row = tid < d / 8 ? ci : cj;
          const int col8 = tid % (d / 8);
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = cent[row * d + 8 * col8 + e];
          image_store_fragment(image, d, row, col8, f);
        }
        if (tid == 0) {
          bias[ci] = -0.5f * ss_i;
          bias[cj] = -0.5f * ss_j;
        }
      }
      if (tid == 0) {
        half_sq[ci] = 0.5f * ss_i;
        half_sq[cj] = 0.5f * ss_j;
        const float ni = floorf(counts[cj] * 0.5f);       // hassign[ci] = hassign[cj] / 2 on integral floats
        counts[ci] = ni;
        counts[cj] -= ni;
      }
      ++n_split;
    }
    __syncthreads();                    // counts / centroids of this split visible to the next walk
  }
  __syncthreads();
  for (long long c = tid; c < k; c += kSplitThreads) counts[c] = 0.f;
  if (tid == 0 && n_split_out != nullptr && n_split > 0) atomicAdd(n_split_out, n_split);
}

}  // namespace

extern "C" int32_t gcr_kmeans_update_f32(const float* x, int64_t n, int32_t d, const int64_t* assign, int64_t k,
                                         float* centroids, float* half_sqnorm, float* sums, float* counts,
                                         void* stream) {
  GCR_CHECK_ARG(n >= 0 && k >= 1 && d >= 1);
  GCR_CHECK_ARG(centroids && half_sqnorm);
  hipStream_t s = (hipStream_t)stream;
  if (n > 0) {
    GCR_CHECK_ARG(x && assign && sums && counts);
    hipError_t err = hipMemsetAsync(sums, 0, sizeof(float) * (size_t)(k * d), s);
    if (err == hipSuccess) err = hipMemsetAsync(counts, 0, sizeof(float) * (size_t)k, s);
    if (err != hipSuccess) return gcr_hip_status(err);
    const int64_t want = (n + 3) / 4;
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, s, x, n,
                       d, assign, k, sums, counts, 1);
  }
  const int64_t wantk = (k + 15) / 16;
  hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((unsigned)(wantk > 4096 ? 4096 : wantk)), dim3(256), 0, s, sums,
                     n > 0 ? counts : nullptr, k, d, centroids, half_sqnorm);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_update_sorted_f32(const float* x, int64_t n, int32_t d, const uint32_t* keys_sorted,
                                                const int32_t* perm, int64_t k, float* centroids, float* half_sqnorm,
                                                float* sums, float* counts, void* stream) {
  GCR_CHECK_ARG(n >= 1 && n < (1ll << 31) && k >= 1 && d >= 1 && d <= 256);
  GCR_CHECK_ARG(x && keys_sorted && perm && centroids && half_sqnorm && sums && counts);
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipMemsetAsync(sums, 0, sizeof(float) * (size_t)(k * d), s);
  if (err == hipSuccess) err = hipMemsetAsync(counts, 0, sizeof(float) * (size_t)k, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  const int64_t want = ((n + 63) / 64 + 3) / 4;
  hipLaunchKernelGGL(kmeans_accumulate_sorted_kernel, dim3((unsigned)(want > 65536 ? 65536 : want)), dim3(256), 0, s, x, n,
                     d, keys_sorted, perm, k, sums, counts);
  const int64_t wantk = (k + 15) / 16;
  hipLaunchKernelGGL(kmeans_finalize_kernel, dim3((unsigned)(wantk > 4096 ? 4096 : wantk)), dim3(256), 0, s, sums, counts,
                     k, d, centroids, half_sqnorm);
  return GCR_LAUNCH_STATUS();
}


extern "C" int32_t gcr_kmeans_lloyd_update_f32(const float* x, int64_t n, int32_t d, const int64_t* assign,
                                               const uint32_t* keys_sorted, const int32_t* perm, int64_t k,
                                               float* centroids, float* half_sqnorm, float* sums, float* counts,
                                               int32_t n_copies, uint64_t seed, int32_t iter, int32_t* n_split,
                                               void* stream) {
  GCR_CHECK_ARG(n >= 1 && n < (1ll << 31) && k >= 1 && k < (1ll << 31) && d >= 1 && d <= 256 && iter >= 0);
  GCR_CHECK_ARG(n_copies >= 1 && n_copies <= 64);
  GCR_CHECK_ARG(x && centroids && half_sqnorm && sums && counts);
  GCR_CHECK_ARG((keys_sorted != nullptr) == (perm != nullptr));
  hipStream_t s = (hipStream_t)stream;
  if (keys_sorted == nullptr && assign == nullptr) {
    // sums / counts already hold this iteration's accumulation (gcr_kmeans_assign_accumulate_f32)
  } else if (keys_sorted != nullptr) {
    const int64_t want = ((n + 63) / 64 + 3) / 4;
    hipLaunchKernelGGL(kmeans_accumulate_sorted_kernel, dim3((unsigned)(want > 65536 ? 65536 : want)), dim3(256), 0, s, x,
                       n, d, keys_sorted, perm, k, sums, counts);
  } else {
    const int64_t want = (n + 3) / 4;
    hipLaunchKernelGGL(kmeans_accumulate_kernel, dim3((unsigned)(want > 16384 ? 16384 : want)), dim3(256), 0, s, x, n, d,
                       assign, k, sums, counts, (int)n_copies);
  }
  GCR_CHECK_ARG(256 % d == 0 && k * (int64_t)d < (1ll << 39));       // rows must not straddle workgroups (d in 32..256)
  hipLaunchKernelGGL(kmeans_finalize_clear_kernel, dim3((unsigned)((k * (int64_t)d + 255) / 256)), dim3(256), 0, s, sums,
                     counts, k, d, centroids, half_sqnorm, keys_sorted != nullptr ? 1 : (int)n_copies);
  hipLaunchKernelGGL(kmeans_split_kernel, dim3(1), dim3(kSplitThreads), 0, s, counts, k, d, centroids, half_sqnorm, n, seed,
                     (uint32_t)iter, n_split);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_lloyd_update_q_f32(const int64_t* sums_q, const int32_t* counts, const float* qscale, int64_t k,
                                                 int32_t d, float* centroids, float* half_sqnorm, float* counts_scratch,
                                                 int64_t n_points, uint64_t seed, int32_t iter, int32_t* n_split, void* image,
                                                 int32_t n_copies, void* stream) {
  GCR_CHECK_ARG(k >= 1 && k < (1ll << 31) && d >= 1 && d <= 256 && 256 % d == 0 && iter >= 0 && n_points >= 1);
  GCR_CHECK_ARG(n_copies >= 1 && n_copies <= 64);
  GCR_CHECK_ARG(sums_q && counts && qscale && centroids && half_sqnorm && counts_scratch);
  GCR_CHECK_ARG(image == nullptr || d == 32 || d == 64 || d == 128);
  hipStream_t s = (hipStream_t)stream;
  u32x4* img = reinterpret_cast<u32x4*>(image);
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  float* bias = image != nullptr ? reinterpret_cast<float*>(img + tiles * 3 * (d / 16) * 64) : nullptr;
  hipLaunchKernelGGL(kmeans_finalize_q_kernel, dim3((unsigned)((k * (int64_t)d + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const long long*>(sums_q), counts, qscale, k, d, centroids, half_sqnorm, counts_scratch, img,
                     bias, (int)n_copies);
  hipLaunchKernelGGL(kmeans_split_kernel, dim3(1), dim3(kSplitThreads), 0, s, counts_scratch, k, d, centroids, half_sqnorm,
                     n_points, seed, (uint32_t)iter, n_split, img, bias);
  return GCR_LAUNCH_STATUS();
}
