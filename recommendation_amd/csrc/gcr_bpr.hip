// Fused BPR pairwise loss over gathered embedding rows (gfx950).
//
// Replaces the gather x3 + dot x2 + log-sigmoid + mean chain of ncl.py:314-317,116-120,
// lightgcn.py:95-108 (n_neg negatives averaged), gcl.py:216-221, sept.py:34-38, mhcn.py:35-39,
// and accumulates the squared norms every regulariser variant needs (lightgcn.py:118,
// gcl.py:222, ncl.py:122-123, sept.py:241) in the same pass.
//
// Forward: a 16-lane group per sample (4 samples per wavefront), float4 per lane when d % 4 == 0
// (d = 64: exactly one 16-B load per lane and row); dots reduced with 4 in-row DPP/LDS-crossbar
// steps; per-block partial sums are written to a workspace and added in block order by a second
// tiny kernel, so the scalar loss is bitwise reproducible.
// Backward: one wavefront per sample, lane l owns columns l, l+64, ...: each row gradient is one
// 256-B no-return global_atomic_add_f32 wave instruction (the shape that runs at the chip-wide
// atomic rate, MI355X_MICROARCH.md §Global float atomics); duplicates in the index vectors
// accumulate like torch's index_put(accumulate=True) backward.
// Large batches (lightgcn.py trains on ALL E edges per step: 10M triples = 30M row atomics = 7.8 ms at
// cfg2) take the SORTED backward instead: the samples are ordered by user / positive item / negative
// item (rocPRIM radix sort of 32-bit keys; the orders of the fixed (u, i) arrays can be reused by the
// caller across steps), a wave walks 64 consecutive entries of one order, sums every run of equal keys
// in registers and issues ONE row atomic per run and chunk: ~10x fewer atomics on the user side, ~60x
// on the item sides.
#include <rocprim/device/device_radix_sort.hpp>

#include "gcr_common.h"

namespace {

constexpr int kFwdThreads = 256;
constexpr int kGroups = kFwdThreads / 16;

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// d(loss)/dx and loss for x = pos - neg
__device__ __forceinline__ void bpr_point(int variant, float x, float& loss, float& dl) {
  const float s = 1.0f / (1.0f + __expf(-x));
  if (variant == GCR_BPR_NCL) {            // -log(1e-5 + sigmoid(x))   ncl.py:119 (literal 10e-6)
    loss = -__logf(10e-6f + s);
    dl = -(s * (1.0f - s)) / (10e-6f + s);
  } else if (variant == GCR_BPR_LOGSIGMOID) {  // -logsigmoid(x)          gcl.py:221, sept.py:37
    loss = (x > 0.f) ? log1pf(__expf(-x)) : (-x + log1pf(__expf(x)));
    dl = -(1.0f - s);
  } else {                                  // -log(sigmoid(x))          lightgcn.py:108
    loss = -__logf(s);
    dl = -(1.0f - s);
  }
}

template <bool VEC4>
__global__ __launch_bounds__(kFwdThreads) void bpr_fwd_kernel(const float* __restrict__ user_tab,
                                                              const float* __restrict__ item_tab, int d,
                                                              const int64_t* __restrict__ u_idx,
                                                              const int64_t* __restrict__ i_idx,
                                                              const int64_t* __restrict__ j_idx, int64_t batch,
                                                              int n_neg, int variant, int64_t n_users, int64_t n_items,
                                                              float* __restrict__ dloss_dx,
                                                              float* __restrict__ block_partials) {
  __shared__ float red[kGroups][5];
  const int l16 = threadIdx.x & 15;
  const int grp = threadIdx.x >> 4;
  float t_loss = 0.f, t_su = 0.f, t_sp = 0.f, t_sn = 0.f, t_err = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * kGroups + grp; b < batch; b += (int64_t)gridDim.x * kGroups) {
    const int64_t u = u_idx[b], i = i_idx[b];
    bool ok = u >= 0 && u < n_users && i >= 0 && i < n_items;
    for (int k = 0; k < n_neg; ++k) {
      const int64_t j = j_idx[b * n_neg + k];
      ok = ok && j >= 0 && j < n_items;
    }
    if (!ok) {  // never dereference a bad id; reported through sums[4]
      t_err += 1.f;
      if (l16 == 0) dloss_dx[b] = __builtin_nanf("");   // marks the sample for the sorted backward
      continue;
    }
    const float* ur = user_tab + u * d;
    const float* pr = item_tab + i * d;
    float pos = 0.f, neg = 0.f, su = 0.f, sp = 0.f, sn = 0.f;
    if (VEC4) {
      for (int c = l16 * 4; c < d; c += 64) {
        const float4 uu = *reinterpret_cast<const float4*>(ur + c);
        const float4 pp = *reinterpret_cast<const float4*>(pr + c);
        pos += uu.x * pp.x + uu.y * pp.y + uu.z * pp.z + uu.w * pp.w;
        su += uu.x * uu.x + uu.y * uu.y + uu.z * uu.z + uu.w * uu.w;
        sp += pp.x * pp.x + pp.y * pp.y + pp.z * pp.z + pp.w * pp.w;
        for (int k = 0; k < n_neg; ++k) {
          const float4 nn = *reinterpret_cast<const float4*>(item_tab + j_idx[b * n_neg + k] * d + c);
          neg += uu.x * nn.x + uu.y * nn.y + uu.z * nn.z + uu.w * nn.w;
          sn += nn.x * nn.x + nn.y * nn.y + nn.z * nn.z + nn.w * nn.w;
        }
      }
    } else {
      for (int c = l16; c < d; c += 16) {
        const float uu = ur[c], pp = pr[c];
        pos += uu * pp;
        su += uu * uu;
        sp += pp * pp;
        for (int k = 0; k < n_neg; ++k) {
          const float nn = item_tab[j_idx[b * n_neg + k] * d + c];
          neg += uu * nn;
          sn += nn * nn;
        }
      }
    }
    pos = group16_sum(pos);
    neg = group16_sum(neg);
    su = group16_sum(su);
    sp = group16_sum(sp);
    sn = group16_sum(sn);
    float loss, dl;
    bpr_point(variant, pos - neg / (float)n_neg, loss, dl);
    if (l16 == 0) dloss_dx[b] = dl;
    t_loss += loss;
    t_su += su;
    t_sp += sp;
    t_sn += sn;
  }
  if (l16 == 0) {
    red[grp][0] = t_loss;
    red[grp][1] = t_su;
    red[grp][2] = t_sp;
    red[grp][3] = t_sn;
    red[grp][4] = t_err;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    float s = 0.f;
    for (int g = 0; g < kGroups; ++g) s += red[g][threadIdx.x];  // fixed order
    block_partials[(int64_t)blockIdx.x * 5 + threadIdx.x] = s;
  }
}

// sums[0..4] = {sum loss, sum |u|^2, sum |p|^2, sum |n|^2, #samples skipped for bad ids}
__global__ void bpr_reduce_kernel(const float* __restrict__ block_partials, int n_blocks, float* __restrict__ sums) {
  const int k = threadIdx.x >> 6;  // 5 waves, one per scalar
  const int lane = threadIdx.x & 63;
  double s = 0.0;
  for (int b = lane; b < n_blocks; b += 64) s += (double)block_partials[(int64_t)b * 5 + k];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) sums[k] = (float)s;
}

template <int NV>
__global__ __launch_bounds__(256) void bpr_bwd_kernel(const float* __restrict__ user_tab,
                                                      const float* __restrict__ item_tab, int d,
                                                      const int64_t* __restrict__ u_idx,
                                                      const int64_t* __restrict__ i_idx,
                                                      const int64_t* __restrict__ j_idx, int64_t batch, int n_neg,
                                                      int64_t n_users, int64_t n_items,
                                                      const float* __restrict__ dloss_dx,
                                                      const float* __restrict__ grad_sums,
                                                      float* __restrict__ grad_user,
                                                      float* __restrict__ grad_item) {
  const int lane = threadIdx.x & 63;
  // upstream gradient of the four forward sums; d(sum |x|^2)/dx = 2x
  const float g_loss = grad_sums[0], c_u = 2.f * grad_sums[1], c_p = 2.f * grad_sums[2], c_n = 2.f * grad_sums[3];
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t b = wave0; b < batch; b += (int64_t)gridDim.x * 4) {
    const int64_t u = u_idx[b], i = i_idx[b];
    bool ok = u >= 0 && u < n_users && i >= 0 && i < n_items;
    for (int k = 0; k < n_neg; ++k) {
      const int64_t j = j_idx[b * n_neg + k];
      ok = ok && j >= 0 && j < n_items;
    }
    if (!ok) continue;
    const float g = g_loss * dloss_dx[b];
    const float gn = -g / (float)n_neg;
    float uu[NV], du[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = lane + 64 * v;
      uu[v] = c < d ? user_tab[u * d + c] : 0.f;
      const float pp = c < d ? item_tab[i * d + c] : 0.f;
      du[v] = g * pp + c_u * uu[v];
      if (c < d) atomicAdd(grad_item + i * d + c, g * uu[v] + c_p * pp);
    }
    for (int k = 0; k < n_neg; ++k) {
      const int64_t j = j_idx[b * n_neg + k];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + 64 * v;
        if (c < d) {
          const float nn = item_tab[j * d + c];
          du[v] += gn * nn;
          atomicAdd(grad_item + j * d + c, gn * uu[v] + c_n * nn);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int c = lane + 64 * v;
      if (c < d) atomicAdd(grad_user + u * d + c, du[v]);
    }
  }
}

// ---- sorted backward ---------------------------------------------------------------------------
constexpr uint32_t kBadKey = 0xFFFFFFFFu;

__global__ void sort_keys_kernel(const int64_t* __restrict__ idx, int64_t n, int64_t n_keys, uint32_t* __restrict__ key,
                                 int32_t* __restrict__ val) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = idx[k];
    key[k] = (v >= 0 && v < n_keys) ? (uint32_t)v : (uint32_t)n_keys;     // invalid ids: one key past the range
    val[k] = (int32_t)k;
  }
}

size_t sort_u32_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 32, (hipStream_t)0);
  return bytes;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// SIDE 0: rows of grad_user keyed by u; 1: rows of grad_item keyed by the positive item; 2: keyed by the
// negative item (entries are the batch * n_neg negative slots).  Same arithmetic as bpr_bwd_kernel.
// POS = false (SIDE 0 only): the positive item's row is left out — the caller adds the positive-pair parts itself (for a
// batch that IS the training graph's edge list they are an SpMM with per-edge coefficients: functional.bpr_edge_sums).
template <int SIDE, int NV, bool ONE_NEG, bool POS = true>
__global__ __launch_bounds__(256) void bpr_bwd_sorted_kernel(
    const float* __restrict__ user_tab, const float* __restrict__ item_tab, int d, const int64_t* __restrict__ u_idx,
    const int64_t* __restrict__ i_idx, const int64_t* __restrict__ j_idx, int64_t batch, int n_neg, int64_t n_users,
    int64_t n_items, const float* __restrict__ dloss_dx, const float* __restrict__ grad_sums,
    const uint32_t* __restrict__ keys, const int32_t* __restrict__ perm, int64_t n_entries, float* __restrict__ grad_out) {
  const int lane = threadIdx.x & 63;
  const float g_loss = grad_sums[0];
  const float c_self = 2.f * grad_sums[1 + SIDE];
  const float* self_tab = SIDE == 0 ? user_tab : item_tab;
  const int64_t n_chunks = (n_entries + 63) / 64;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks; chunk += (int64_t)gridDim.x * 4) {
    const int64_t c0 = chunk * 64;
    const int cnt = (int)(n_entries - c0 < 64 ? n_entries - c0 : 64);
    // lane e prepares entry c0 + e: key, coefficient and the row(s) it pulls in
    uint32_t my_key = kBadKey;
    float my_coef = 0.f;
    int64_t my_u = 0, my_i = 0, my_b = 0, my_j = 0;
    if (lane < cnt) {
      my_key = keys[c0 + lane];
      if (my_key >= (uint32_t)(SIDE == 0 ? n_users : n_items)) my_key = kBadKey;   // gcr_sort_index's out-of-range key
      const int64_t slot = perm[c0 + lane];
      const int64_t b = SIDE == 2 ? slot / n_neg : slot;
      // the forward left NaN in dloss_dx for a sample with any id out of range: it contributes nothing
      const float dl = dloss_dx[b];
      const bool ok = my_key != kBadKey && dl == dl;
      const float g = g_loss * dl;
      my_coef = SIDE == 2 ? -g / (float)n_neg : g;
      if (SIDE == 0) {
        if (POS) my_i = ok ? i_idx[b] : 0;
        if (ONE_NEG) my_j = ok ? j_idx[b] : 0;
      } else {
        my_u = ok ? u_idx[b] : 0;
      }
      my_b = b;
      if (!ok) my_key = kBadKey;
    }
    uint32_t cur = kBadKey;
    float acc[NV];
    float run_n = 0.f;                      // entries of the current run: the regulariser's 2 g x per entry
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    auto flush = [&]() {
      if (cur != kBadKey) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = lane + 64 * v;
          if (c < d) atomicAdd(grad_out + (int64_t)cur * d + c, acc[v] + c_self * run_n * self_tab[(int64_t)cur * d + c]);
        }
      }
    };
    auto lane64 = [&](int64_t x, int e) {
      return ((int64_t)__builtin_amdgcn_readlane((int)(x >> 32), e) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)x, e);
    };
    // kGather entries at a time: all their row gathers are issued before the first is consumed
    constexpr int kGather = NV == 1 ? 8 : 4;
    for (int e0 = 0; e0 < cnt; e0 += kGather) {
      uint32_t key[kGather];
      float coef[kGather], row[kGather][NV];
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        const int e = e0 + q < cnt ? e0 + q : cnt - 1;
        key[q] = e0 + q < cnt ? (uint32_t)__builtin_amdgcn_readlane((int)my_key, e) : kBadKey;
        coef[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_coef), e));
        const bool live = key[q] != kBadKey;
        const int64_t src = live ? lane64(SIDE == 0 ? my_i : my_u, e) : 0;
        const float* tab = SIDE == 0 ? item_tab : user_tab;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = lane + 64 * v;
          row[q][v] = (c < d && (POS || SIDE != 0)) ? tab[src * d + c] : 0.f;
        }
        if (SIDE == 0 && ONE_NEG) {
          const int64_t j = live ? lane64(my_j, e) : 0;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const int c = lane + 64 * v;
            if (c < d) row[q][v] -= item_tab[j * d + c];
          }
        } else if (SIDE == 0) {
          const int64_t b = lane64(my_b, e);
          const float inv = 1.0f / (float)n_neg;
          for (int k = 0; k < n_neg; ++k) {
            const int64_t j = live ? j_idx[b * n_neg + k] : 0;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              const int c = lane + 64 * v;
              if (c < d) row[q][v] -= inv * item_tab[j * d + c];
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        if (key[q] == kBadKey) continue;
        if (key[q] != cur) {
          flush();
          cur = key[q];
          run_n = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[v] = 0.f;
        }
        run_n += 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += coef[q] * row[q][v];
      }
    }
    flush();
  }
}

// Per-non-zero values of the coefficient operator of functional.bpr_edge_sums: val[e] = g * dL/dx of the pair that
// non-zero e stands for — the pair itself in the user-major half [0, E), its mirror image in the item-major half
// [E, 2E); a NaN (a sample that the forward dropped: an id out of range) counts 0 and is tallied per positive item.
__global__ __launch_bounds__(256) void bpr_edge_values_kernel(const float* __restrict__ dloss_dx,
                                                              const int64_t* __restrict__ mirror,
                                                              const int32_t* __restrict__ col, int64_t n_users, int64_t e_pairs,
                                                              const float* __restrict__ grad_sums, float* __restrict__ val,
                                                              float* __restrict__ dropped_per_item) {
  const float g = grad_sums[0];
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < 2 * e_pairs; e += (int64_t)gridDim.x * 256) {
    const int64_t pair = e < e_pairs ? e : mirror[e];
    const float dl = dloss_dx[pair];
    const bool live = dl == dl;
    val[e] = live ? g * dl : 0.f;
    if (!live && e < e_pairs) atomicAdd(dropped_per_item + (col[e] - n_users), 1.0f);
  }
}

// The negatives' USER side of the same full batch as one more SpMM: the training edges are in user-major order, so the
// negatives drawn for them form a CSR block [U x I] on the graph's own user row pointer (times n_neg) whose columns are
// this step's negatives and whose values are - g dL/dx / n_neg:  dU[u] -= sum_{e in row u} g dl_e mean_k I[j_ek].  This
// kernel writes that block's columns (sanitised: a dropped sample's slot points at row 0 with value 0) and values, and
// tallies the dropped samples per user (the |U[u]|^2 term counts live samples only).
__global__ __launch_bounds__(256) void bpr_neg_block_kernel(const float* __restrict__ dloss_dx, const int64_t* __restrict__ j_idx,
                                                            const int64_t* __restrict__ u_idx, int64_t batch, int n_neg,
                                                            int64_t n_items, const float* __restrict__ grad_sums,
                                                            int32_t* __restrict__ col, float* __restrict__ val,
                                                            float* __restrict__ dropped_per_user,
                                                            uint32_t* __restrict__ sort_key, uint64_t* __restrict__ sort_payload) {
  const float g = -grad_sums[0] / (float)n_neg;
  const int64_t slots = batch * n_neg;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < slots; s += (int64_t)gridDim.x * 256) {
    const int64_t e = s / n_neg;
    const float dl = dloss_dx[e];
    const int64_t j = j_idx[s];
    const bool live = dl == dl && j >= 0 && j < n_items;   // (a bad id anywhere in the sample left NaN in dloss_dx)
    const float v = live ? g * dl : 0.f;
    if (col != nullptr) {
      col[s] = live ? (int32_t)j : 0;
      val[s] = v;
    }
    const int64_t u = u_idx[e];
    if (dropped_per_user != nullptr && !(dl == dl) && s == e * n_neg) atomicAdd(dropped_per_user + u, 1.0f);
    if (sort_key) {                                         // the same slot for the ITEM side: key j, payload (u, value)
      sort_key[s] = live ? (uint32_t)j : (uint32_t)n_items;
      sort_payload[s] = live ? ((uint64_t)(uint32_t)u << 32) | (uint64_t)__builtin_bit_cast(uint32_t, v) : 0ull;
    }
  }
}

// The negatives' ITEM rows from the slots ordered by negative item WITH their payload (gcr_sort_pairs_u64 of the key /
// payload arrays bpr_neg_block_kernel writes): a wave walks 64 consecutive sorted entries — key, user and coefficient
// arrive by three coalesced loads instead of perm -> sample -> (u_idx, dloss_dx) —, keeps 16 user rows in flight and adds
// a run of equal keys to its item row with one row atomic (+ the run's share of the |I[j]|^2 term).
template <int NV>
__global__ __launch_bounds__(256) void bpr_neg_items_sorted_kernel(const float* __restrict__ user_tab,
                                                                   const float* __restrict__ item_tab, int d,
                                                                   const uint32_t* __restrict__ keys,
                                                                   const uint64_t* __restrict__ payload, int64_t n_entries,
                                                                   int64_t n_items, const float* __restrict__ grad_sums,
                                                                   float* __restrict__ grad_item) {
  const int lane = threadIdx.x & 63;
  const float c_self = 2.f * grad_sums[3];
  const int64_t n_chunks = (n_entries + 63) / 64;
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); chunk < n_chunks; chunk += (int64_t)gridDim.x * 4) {
    const int64_t c0 = chunk * 64;
    const int cnt = (int)(n_entries - c0 < 64 ? n_entries - c0 : 64);
    uint32_t my_key = kBadKey, my_u = 0;
    float my_coef = 0.f;
    if (lane < cnt) {
      my_key = keys[c0 + lane];
      if (my_key >= (uint32_t)n_items) my_key = kBadKey;          // dropped slots carry the key n_items and sort last
      const uint64_t p = payload[c0 + lane];
      my_u = (uint32_t)(p >> 32);
      my_coef = __builtin_bit_cast(float, (uint32_t)p);
    }
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)my_key) == kBadKey) continue;   // sorted: nothing live from here on
    uint32_t cur = kBadKey;
    float acc[NV];
    float run_n = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    auto flush = [&]() {
      if (cur != kBadKey) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = lane + 64 * v;
          if (c < d) atomicAdd(grad_item + (int64_t)cur * d + c, acc[v] + c_self * run_n * item_tab[(int64_t)cur * d + c]);
        }
      }
    };
    constexpr int kGather = NV == 1 ? 16 : 8;
    for (int e0 = 0; e0 < cnt; e0 += kGather) {
      uint32_t key[kGather];
      float coef[kGather], row[kGather][NV];
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        const int e = e0 + q < cnt ? e0 + q : cnt - 1;
        key[q] = e0 + q < cnt ? (uint32_t)__builtin_amdgcn_readlane((int)my_key, e) : kBadKey;
        coef[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_coef), e));
        const int64_t src = key[q] != kBadKey ? (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)my_u, e) : 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = lane + 64 * v;
          row[q][v] = c < d ? user_tab[src * d + c] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        if (key[q] == kBadKey) continue;
        if (key[q] != cur) {
          flush();
          cur = key[q];
          run_n = 0.f;
#pragma unroll
          for (int v = 0; v < NV; ++v) acc[v] = 0.f;
        }
        run_n += 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += coef[q] * row[q][v];
      }
    }
    flush();
  }
}

size_t sort_u32_u64_temp_bytes(int64_t n) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  uint64_t* v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0, 32, (hipStream_t)0);
  return bytes;
}

int fwd_blocks(int64_t batch) {
  const int64_t want = (batch + kGroups - 1) / kGroups;
  return (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
}

}  // namespace

extern "C" int64_t gcr_bpr_workspace_floats(int64_t batch) { return (int64_t)fwd_blocks(batch) * 5; }

extern "C" int32_t gcr_bpr_fwd_f32(const float* user_tab, const float* item_tab, int32_t d, const int64_t* u_idx,
                                   const int64_t* i_idx, const int64_t* j_idx, int64_t batch, int32_t n_neg,
                                   int32_t variant, int64_t n_users, int64_t n_items, float* dloss_dx, float* sums,
                                   float* workspace, void* stream) {
  GCR_CHECK_ARG(batch >= 0 && n_neg >= 1 && d >= 1 && d <= 256 && n_users >= 0 && n_items >= 0);
  GCR_CHECK_ARG(variant >= GCR_BPR_NCL && variant <= GCR_BPR_LOG_SIGMOID);
  GCR_CHECK_ARG(sums != nullptr && workspace != nullptr);
  GCR_CHECK_ARG(batch == 0 || (user_tab && item_tab && u_idx && i_idx && j_idx && dloss_dx));
  hipStream_t s = (hipStream_t)stream;
  const int blocks = fwd_blocks(batch);
  if (d % 4 == 0)
    hipLaunchKernelGGL((bpr_fwd_kernel<true>), dim3(blocks), dim3(kFwdThreads), 0, s, user_tab, item_tab, d, u_idx,
                       i_idx, j_idx, batch, n_neg, variant, n_users, n_items, dloss_dx, workspace);
  else
    hipLaunchKernelGGL((bpr_fwd_kernel<false>), dim3(blocks), dim3(kFwdThreads), 0, s, user_tab, item_tab, d, u_idx,
                       i_idx, j_idx, batch, n_neg, variant, n_users, n_items, dloss_dx, workspace);
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(bpr_reduce_kernel, dim3(1), dim3(320), 0, s, workspace, blocks, sums);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_bpr_bwd_f32(const float* user_tab, const float* item_tab, int32_t d, const int64_t* u_idx,
                                   const int64_t* i_idx, const int64_t* j_idx, int64_t batch, int32_t n_neg,
                                   int64_t n_users, int64_t n_items, const float* dloss_dx, const float* grad_sums,
                                   float* grad_user, float* grad_item, void* stream) {
  GCR_CHECK_ARG(batch >= 0 && n_neg >= 1 && d >= 1 && d <= 256 && n_users >= 0 && n_items >= 0);
  if (batch == 0) return GCR_OK;
  GCR_CHECK_ARG(user_tab && item_tab && u_idx && i_idx && j_idx && dloss_dx && grad_sums && grad_user && grad_item);
  hipStream_t s = (hipStream_t)stream;
  const int64_t want = (batch + 3) / 4;
  const int blocks = (int)(want > 8192 ? 8192 : want);
#define GCR_BWD(NV)                                                                                              \
  hipLaunchKernelGGL((bpr_bwd_kernel<NV>), dim3(blocks), dim3(256), 0, s, user_tab, item_tab, d, u_idx, i_idx, \
                     j_idx, batch, n_neg, n_users, n_items, dloss_dx, grad_sums, grad_user, grad_item)
  if (d <= 64) GCR_BWD(1);
  else if (d <= 128) GCR_BWD(2);
  else if (d <= 192) GCR_BWD(3);
  else GCR_BWD(4);
#undef GCR_BWD
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_sort_index_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return (int64_t)(align256(sizeof(uint32_t) * (size_t)n) + align256(sizeof(int32_t) * (size_t)n) + sort_u32_temp_bytes(n));
}

extern "C" int32_t gcr_sort_index(const int64_t* idx, int64_t n, int64_t n_keys, uint32_t* keys_sorted, int32_t* perm,
                                  void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0 && n < (1ll << 31) && n_keys >= 0 && n_keys < 0xFFFFFFFFll);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(idx && keys_sorted && perm && workspace);
  hipStream_t s = (hipStream_t)stream;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  uint32_t* key_in = reinterpret_cast<uint32_t*>(w);
  int32_t* val_in = reinterpret_cast<int32_t*>(w + align256(sizeof(uint32_t) * (size_t)n));
  void* tmp = w + align256(sizeof(uint32_t) * (size_t)n) + align256(sizeof(int32_t) * (size_t)n);
  size_t tmp_bytes = sort_u32_temp_bytes(n);
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(sort_keys_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, s, idx, n, n_keys,
                     key_in, val_in);
  // only the bits that n_keys (the key of the invalid ids, which therefore sort last) needs
  int bits = 1;
  while (bits < 32 && ((uint64_t)n_keys >> bits) != 0) ++bits;
  hipError_t err = rocprim::radix_sort_pairs(tmp, tmp_bytes, key_in, keys_sorted, val_in, perm, (size_t)n, 0, bits, s);
  if (err != hipSuccess) return gcr_hip_status(err);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_bpr_bwd_sorted_f32(const float* user_tab, const float* item_tab, int32_t d, const int64_t* u_idx,
                                          const int64_t* i_idx, const int64_t* j_idx, int64_t batch, int32_t n_neg,
                                          int64_t n_users, int64_t n_items, const float* dloss_dx,
                                          const float* grad_sums, const uint32_t* keys_u, const int32_t* perm_u,
                                          const uint32_t* keys_i, const int32_t* perm_i, const uint32_t* keys_j,
                                          const int32_t* perm_j, float* grad_user, float* grad_item, void* stream) {
  GCR_CHECK_ARG(batch >= 0 && n_neg >= 1 && d >= 1 && d <= 256 && n_users >= 0 && n_items >= 0);
  GCR_CHECK_ARG(batch * n_neg < (1ll << 31));
  if (batch == 0) return GCR_OK;
  GCR_CHECK_ARG(user_tab && item_tab && u_idx && j_idx && dloss_dx && grad_sums && grad_user && grad_item);
  GCR_CHECK_ARG((keys_j != nullptr) == (perm_j != nullptr) && (keys_i != nullptr) == (perm_i != nullptr) &&
                (keys_u != nullptr) == (perm_u != nullptr));
  const bool pos = keys_i != nullptr;                     // false: the caller adds the positive-pair parts (gcr.h)
  const bool user_side = keys_u != nullptr;               // false (only without the positive parts): the caller adds the
  GCR_CHECK_ARG(!pos || (i_idx != nullptr && user_side)); // users' rows too (gcr_bpr_neg_block_f32 + one SpMM)
  hipStream_t s = (hipStream_t)stream;
  auto blocks_for = [](int64_t n_entries) {
    const int64_t want = ((n_entries + 63) / 64 + 3) / 4;
    return (unsigned)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
  };
#define GCR_SIDE(SIDE, NV, KEYS, PERM, NE, OUT)                                                                       \
  if (n_neg == 1)                                                                                                     \
    hipLaunchKernelGGL((bpr_bwd_sorted_kernel<SIDE, NV, true>), dim3(blocks_for(NE)), dim3(256), 0, s, user_tab,       \
                       item_tab, d, u_idx, i_idx, j_idx, batch, n_neg, n_users, n_items, dloss_dx, grad_sums, KEYS,    \
                       PERM, NE, OUT);                                                                                 \
  else                                                                                                                \
    hipLaunchKernelGGL((bpr_bwd_sorted_kernel<SIDE, NV, false>), dim3(blocks_for(NE)), dim3(256), 0, s, user_tab,      \
                       item_tab, d, u_idx, i_idx, j_idx, batch, n_neg, n_users, n_items, dloss_dx, grad_sums, KEYS,    \
                       PERM, NE, OUT)
#define GCR_NEG0(NV)                                                                                                  \
  if (n_neg == 1)                                                                                                     \
    hipLaunchKernelGGL((bpr_bwd_sorted_kernel<0, NV, true, false>), dim3(blocks_for(batch)), dim3(256), 0, s, user_tab, \
                       item_tab, d, u_idx, i_idx, j_idx, batch, n_neg, n_users, n_items, dloss_dx, grad_sums, keys_u, \
                       perm_u, batch, grad_user);                                                                     \
  else                                                                                                                \
    hipLaunchKernelGGL((bpr_bwd_sorted_kernel<0, NV, false, false>), dim3(blocks_for(batch)), dim3(256), 0, s, user_tab, \
                       item_tab, d, u_idx, i_idx, j_idx, batch, n_neg, n_users, n_items, dloss_dx, grad_sums, keys_u, \
                       perm_u, batch, grad_user)
#define GCR_ALL(NV)                                           \
  if (pos) {                                                  \
    GCR_SIDE(0, NV, keys_u, perm_u, batch, grad_user);        \
    GCR_SIDE(1, NV, keys_i, perm_i, batch, grad_item);        \
  } else if (user_side) {                                     \
    GCR_NEG0(NV);                                             \
  }                                                           \
  if (keys_j != nullptr) { GCR_SIDE(2, NV, keys_j, perm_j, batch * n_neg, grad_item); }
  if (d <= 64) { GCR_ALL(1); }
  else if (d <= 128) { GCR_ALL(2); }
  else if (d <= 192) { GCR_ALL(3); }
  else { GCR_ALL(4); }
#undef GCR_ALL
#undef GCR_NEG0
#undef GCR_SIDE
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_bpr_edge_values_f32(const float* dloss_dx, const int64_t* mirror, const int32_t* col, int64_t n_users,
                                           int64_t n_pairs, const float* grad_sums, float* val, float* dropped_per_item,
                                           void* stream) {
  GCR_CHECK_ARG(n_pairs >= 0 && n_users >= 0 && n_pairs < (1ll << 40));
  if (n_pairs == 0) return GCR_OK;
  GCR_CHECK_ARG(dloss_dx && mirror && col && grad_sums && val && dropped_per_item);
  const int64_t want = (2 * n_pairs + 255) / 256;
  hipLaunchKernelGGL(bpr_edge_values_kernel, dim3((unsigned)(want > 65536 ? 65536 : want)), dim3(256), 0, (hipStream_t)stream,
                     dloss_dx, mirror, col, n_users, n_pairs, grad_sums, val, dropped_per_item);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_bpr_neg_block_f32(const float* dloss_dx, const int64_t* j_idx, const int64_t* u_idx, int64_t batch,
                                         int32_t n_neg, int64_t n_items, const float* grad_sums, int32_t* col, float* val,
                                         float* dropped_per_user, uint32_t* sort_key, uint64_t* sort_payload, void* stream) {
  GCR_CHECK_ARG(batch >= 0 && n_neg >= 1 && n_items >= 1 && batch * n_neg < (1ll << 40));
  if (batch == 0) return GCR_OK;
  GCR_CHECK_ARG(dloss_dx && j_idx && u_idx && grad_sums);
  GCR_CHECK_ARG((col != nullptr) == (val != nullptr) && (col != nullptr) == (dropped_per_user != nullptr));
  GCR_CHECK_ARG((sort_key != nullptr) == (sort_payload != nullptr) && (!sort_key || n_items < 0xFFFFFFFFll));
  GCR_CHECK_ARG(col != nullptr || sort_key != nullptr);
  const int64_t want = (batch * n_neg + 255) / 256;
  hipLaunchKernelGGL(bpr_neg_block_kernel, dim3((unsigned)(want > 65536 ? 65536 : want)), dim3(256), 0, (hipStream_t)stream,
                     dloss_dx, j_idx, u_idx, batch, (int)n_neg, n_items, grad_sums, col, val, dropped_per_user, sort_key,
                     sort_payload);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_sort_pairs_u64_workspace_bytes(int64_t n) {
  return n <= 0 ? 0 : (int64_t)sort_u32_u64_temp_bytes(n);
}

extern "C" int32_t gcr_sort_pairs_u64(const uint32_t* keys, const uint64_t* payload, int64_t n, int64_t n_keys,
                                      uint32_t* keys_sorted, uint64_t* payload_sorted, void* workspace, void* stream) {
  GCR_CHECK_ARG(n >= 0 && n < (1ll << 31) && n_keys >= 0 && n_keys < 0xFFFFFFFFll);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(keys && payload && keys_sorted && payload_sorted && workspace);
  size_t tmp_bytes = sort_u32_u64_temp_bytes(n);
  int bits = 1;                                            // keys live in [0, n_keys]
  while (bits < 32 && ((uint64_t)n_keys >> bits) != 0) ++bits;
  hipError_t err = rocprim::radix_sort_pairs(workspace, tmp_bytes, keys, keys_sorted, payload, payload_sorted, (size_t)n, 0,
                                             bits, (hipStream_t)stream);
  if (err != hipSuccess) return gcr_hip_status(err);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_bpr_neg_items_sorted_f32(const float* user_tab, const float* item_tab, int32_t d,
                                                const uint32_t* keys_sorted, const uint64_t* payload_sorted,
                                                int64_t n_entries, int64_t n_users, int64_t n_items,
                                                const float* grad_sums, float* grad_item, void* stream) {
  GCR_CHECK_ARG(n_entries >= 0 && d >= 1 && d <= 256 && n_users >= 0 && n_users < (1ll << 32) && n_items >= 0 &&
                n_items < 0xFFFFFFFFll);
  if (n_entries == 0) return GCR_OK;
  GCR_CHECK_ARG(user_tab && item_tab && keys_sorted && payload_sorted && grad_sums && grad_item);
  const int64_t want = ((n_entries + 63) / 64 + 3) / 4;
  const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 65536 ? 65536 : want));
#define GCR_NIS(NV)                                                                                                  \
  hipLaunchKernelGGL((bpr_neg_items_sorted_kernel<NV>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, user_tab,   \
                     item_tab, d, keys_sorted, payload_sorted, n_entries, n_items, grad_sums, grad_item)
  if (d <= 64) GCR_NIS(1);
  else if (d <= 128) GCR_NIS(2);
  else if (d <= 192) GCR_NIS(3);
  else GCR_NIS(4);
#undef GCR_NIS
  return GCR_LAUNCH_STATUS();
}
