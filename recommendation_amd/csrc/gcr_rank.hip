// Full-ranking evaluation: scores of a block of users against ALL items on the fp32
// MFMA, training positives masked, exact top-N per user.
//
// Replaces  torch.matmul(user_emb, item_emb.t()) + per-user python masking + argsort / topk
//   lightgcn.py:48-57, gcl.py:87-96 (`scores_user[known] = -inf; argsort(-scores)[:k]`),
//   ncl.py:253-264,390-394 (`candidates[rated] = -1e8; torch.topk(candidates, max_N)`).
//
// score_rows: the tile is oriented with ITEMS on the lanes (stationary, 64 per wave in registers)
//   and gathered USER rows streamed through LDS, so that one accumulator register of the 32x32
//   tile is 32 consecutive items of one user: the store is a coalesced 128-B row segment.
// topk_masked: one 256-thread block per user row: scatter -inf over the user's training items,
//   3-pass radix select (11 + 11 + 10 bits) of the k-th largest key over the L2-resident row, gather
//   the winners, bitonic sort by (score desc, item id asc).  Ties at the threshold are resolved
//   towards the smaller item id (deterministic).
#include "gcr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 32;

template <int D>
struct RShape {
  static constexpr int KH = D / 2;
  static constexpr int STRIDE = D + 4;
  static constexpr int NT = D <= 128 ? 2 : 1;
  static constexpr int NLD = (kTile * D / 4) / 256;
  static constexpr int ITEMS_PER_BLOCK = 4 * 32 * NT;
};

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int kTileJ = 32;
#include "gcr_b3.h"   // split-operand bf16 MFMA helpers (inside this anonymous namespace)

template <int D>
__global__ __launch_bounds__(256, 2) void score_rows_kernel(const float* __restrict__ user_emb,
                                                            const int64_t* __restrict__ user_ids, int64_t n_query,
                                                            int64_t n_users, const float* __restrict__ item_emb,
                                                            int64_t n_items, int64_t q_tiles_per_split, int nsplit,
                                                            float* __restrict__ scores) {
  using S = RShape<D>;
  __shared__ __align__(16) float lds[2][kTile * S::STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t iblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t j0 = (iblk * 4 + wave) * (32 * S::NT);

  float bfrag[S::NT][S::KH];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const int64_t j = j0 + 32 * t + i32;
    const bool valid = j < n_items;
    const float* p = item_emb + (valid ? j : 0) * D + h * S::KH;
#pragma unroll
    for (int q = 0; q < S::KH / 4; ++q) {
      const float4 v = valid ? *reinterpret_cast<const float4*>(p + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
      bfrag[t][4 * q + 0] = v.x;
      bfrag[t][4 * q + 1] = v.y;
      bfrag[t][4 * q + 2] = v.z;
      bfrag[t][4 * q + 3] = v.w;
    }
  }
  const int64_t total_tiles = (n_query + kTile - 1) / kTile;
  const int64_t tile0 = (int64_t)split * q_tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + q_tiles_per_split);
  float4 regs[S::NLD];
  auto stage_load = [&](int64_t q0) {
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t q = q0 + row;
      int64_t uid = q < n_query ? (user_ids != nullptr ? user_ids[q] : q) : 0;
      const bool ok = q < n_query && uid >= 0 && uid < n_users;
      uid = ok ? uid : 0;
      float4 v = *reinterpret_cast<const float4*>(user_emb + uid * D + 4 * c4);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      regs[u] = v;
    }
  };
  auto stage_store = [&](float* tile) {
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      *reinterpret_cast<float4*>(tile + row * S::STRIDE + 4 * c4) = regs[u];
    }
  };
  if (tile0 < tile1) {
    stage_load(tile0 * kTile);
    stage_store(lds[0]);
  }
  __syncthreads();
  for (int64_t tt = tile0; tt < tile1; ++tt) {
    const int cur = (int)((tt - tile0) & 1);
    const int64_t nxt = tt + 1 < tile1 ? tt + 1 : tt;
    stage_load(nxt * kTile);
    f32x16 acc[S::NT];
#pragma unroll
    for (int t = 0; t < S::NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* base = lds[cur] + i32 * S::STRIDE + h * S::KH;
#pragma unroll
    for (int q = 0; q < S::KH / 4; ++q) {
      const float4 av = *reinterpret_cast<const float4*>(base + 4 * q);
      const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < S::NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
    }
    // register r = query row acc_row(r, h) of this tile, lane = item: 128-B coalesced row segments
    const int64_t q0 = tt * kTile;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t q = q0 + acc_row(r, h);
      if (q < n_query) {
#pragma unroll
        for (int t = 0; t < S::NT; ++t) {
          const int64_t j = j0 + 32 * t + i32;
          if (j < n_items) scores[q * n_items + j] = acc[t][r];
        }
      }
    }
    stage_store(lds[cur ^ 1]);
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t order_key(float f) {
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // larger float -> larger key; -inf smallest
}

// ------------------------------------------------------------------------------------------------
// Fused ranking: the U x I score matrix never reaches HBM.
//   pass 0 (MODE 0)  scores of every query user against the first kSample items, written dense to a small
//                    [Q, kSample] buffer; rank_threshold_kernel masks the training positives in it and takes the
//                    k-th largest: a lower bound thr[q] of the row's k-th largest eligible score;
//   pass 1 (MODE 1)  the same score tiles over ALL items; a score >= thr[q] is appended to the user's candidate
//                    list — about k * I / kSample entries per user instead of I scores;
//   rank_finish      gathers a user's candidates, drops the training positives among them (binary search in the
//                    user's sorted row), bitonic-sorts the rest in LDS, writes the exact top-k
//                    (ties -> smaller item id).  A list that overflowed (or came up short) sets status[q] = 1 and
//                    the host re-ranks that user through gcr_score_rows_f32 + gcr_topk_masked_f32.
// Score tile: the split-operand bf16 MFMA (gcr_b3.h, f32-accurate), users stationary on the lanes (64 per wave),
// items streamed through LDS.  A lane owns its user for the block's whole item range, so the candidate counter is
// a lane-private register and every (user, item split, lane half) triple appends to its own region: no atomics.  Both passes
// build the same tiles with the same MFMA sequence, so a score is bitwise the same in both.
// ------------------------------------------------------------------------------------------------
constexpr int kFuseSample = 4096;
constexpr int kFuseCap = 4096;          // candidate entries per user over all item splits

__device__ __forceinline__ bool is_train_item(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ items,
                                              int64_t uid, int32_t j) {
  int64_t lo = rowptr[uid], hi = rowptr[uid + 1];
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    const int32_t v = items[mid];
    if (v == j) return true;
    if (v < j) lo = mid + 1; else hi = mid;
  }
  return false;
}

template <int D, int MODE>
__global__ __launch_bounds__(256, 2) void rank_fused_b3_kernel(
    const float* __restrict__ user_emb, const int64_t* __restrict__ user_ids, int64_t n_query, int64_t n_users,
    const float* __restrict__ item_emb, int64_t n_items, int64_t j_end, int nsplit, int64_t tiles_per_split,
    float* __restrict__ sample, int sample_stride, const float* __restrict__ thr,
    const int64_t* __restrict__ user_rowptr, const int32_t* __restrict__ user_items,
    unsigned long long* __restrict__ cand, int32_t* __restrict__ counts, int cap_split) {
  using S = ShapeB3<D>;
  __shared__ __align__(16) unsigned char lds[2][3 * S::PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t ublk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t q0 = (ublk * 4 + wave) * (32 * S::NT);

  // stationary users (gathered by id): three bf16 planes of features [h * KH, (h + 1) * KH)
  u32x4 bq[S::NT][3][S::KC];
  int64_t uid[S::NT];
  float thr_l[S::NT];
  int cnt[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const int64_t q = q0 + 32 * t + i32;
    int64_t u = q < n_query ? (user_ids != nullptr ? user_ids[q] : q) : -1;
    const bool ok = u >= 0 && u < n_users;
    uid[t] = ok ? u : -1;
    load_stationary_b3<D>(user_emb, nullptr, ok ? u + 1 : 0, ok ? u : 0, h, 1.0f, bq[t]);
    thr_l[t] = (MODE == 1 && ok) ? thr[q] : INFINITY;       // an invalid query never produces a candidate
    cnt[t] = 0;
  }

  const int64_t total_tiles = (j_end + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  float4 regs[S::NLD];
  auto stage_load_items = [&](int64_t j0) {
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t j = j0 + row;
      const int64_t jj = j < n_items ? j : n_items - 1;
      float4 v = *reinterpret_cast<const float4*>(item_emb + jj * D + 4 * c4);
      if (j >= n_items) v = make_float4(0.f, 0.f, 0.f, 0.f);
      regs[u] = v;
    }
  };
  if (tile0 < tile1) {
    stage_load_items(tile0 * kTileJ);
    stage_store_b3<D>(lds[0], tid, regs);
  }
  __syncthreads();
  for (int64_t tt = tile0; tt < tile1; ++tt) {
    const int cur = (int)((tt - tile0) & 1);
    const int64_t nxt = tt + 1 < tile1 ? tt + 1 : tt;
    stage_load_items(nxt * kTileJ);
    f32x16 acc[S::NT];
    score_tile_b3<D, S::NT>(lds[cur], i32, h, bq, acc);
    const int64_t j0 = tt * kTileJ;
    if (MODE == 0) {
      // dense sample: registers 4g .. 4g+3 are items j0 + 8g + 4h .. +3 of the lane's user: one 16-B store
#pragma unroll
      for (int t = 0; t < S::NT; ++t) {
        const int64_t q = q0 + 32 * t + i32;
        if (q < n_query) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int64_t j = j0 + 8 * g + 4 * h;
            if (j + 3 < (int64_t)sample_stride)
              *reinterpret_cast<float4*>(sample + q * sample_stride + j) =
                  make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
          }
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < S::NT; ++t) {
        float mx = fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3]));
#pragma unroll
        for (int g = 1; g < 4; ++g)
          mx = fmaxf(mx, fmaxf(fmaxf(acc[t][4 * g], acc[t][4 * g + 1]), fmaxf(acc[t][4 * g + 2], acc[t][4 * g + 3])));
        if (__any(mx >= thr_l[t])) {                   // rare: ~k / kSample of the scores pass
          if (mx >= thr_l[t]) {
            const int64_t q = q0 + 32 * t + i32;
            unsigned long long* region = cand + (((int64_t)q * nsplit + split) * 2 + h) * cap_split;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int64_t j = j0 + acc_row(r, h);
              if (acc[t][r] >= thr_l[t] && j < n_items) {
                // training positives are weeded out by rank_finish_kernel (a binary search in global memory here
                // would stall the wave that feeds the matrix pipe)
                if (cnt[t] < cap_split)
                  region[cnt[t]] = ((unsigned long long)order_key(acc[t][r]) << 32) | (0xFFFFFFFFu - (uint32_t)j);
                ++cnt[t];                              // keeps counting past the capacity: overflow is detected
              }
            }
          }
        }
      }
    }
    stage_store_b3<D>(lds[cur ^ 1], tid, regs);
    __syncthreads();
  }
  if (MODE == 1) {
#pragma unroll
    for (int t = 0; t < S::NT; ++t) {
      const int64_t q = q0 + 32 * t + i32;
      if (q < n_query) counts[(q * nsplit + split) * 2 + h] = cnt[t];     // one sub-region per lane half
    }
  }
}

constexpr int kTopThreads = 256;
constexpr int kMaxK = 256;
constexpr int kEqCap = 1024;

// finds, scanning bins from the top, the bin where the cumulative count reaches `want`
__device__ int pick_bin(const uint32_t* hist, int nbins, uint32_t want, uint32_t* above_out, uint32_t* sh) {
  // 256 threads: each sums a contiguous slice from the top, then thread 0 walks the slice sums
  const int tid = threadIdx.x;
  const int per = (nbins + kTopThreads - 1) / kTopThreads;
  uint32_t s = 0;
  for (int b = 0; b < per; ++b) {
    const int bin = nbins - 1 - (tid * per + b);
    if (bin >= 0) s += hist[bin];
  }
  sh[tid] = s;
  __syncthreads();
  if (tid == 0) {
    uint32_t cum = 0;
    int slice = 0;
    for (; slice < kTopThreads; ++slice) {
      if (cum + sh[slice] >= want) break;
      cum += sh[slice];
    }
    int bin = nbins - 1 - slice * per;
    for (int b = 0; b < per; ++b, --bin) {
      const uint32_t c = bin >= 0 ? hist[bin] : 0;
      if (cum + c >= want) break;
      cum += c;
    }
    if (bin < 0) bin = 0;
    sh[kTopThreads] = (uint32_t)bin;
    sh[kTopThreads + 1] = cum;
  }
  __syncthreads();
  *above_out = sh[kTopThreads + 1];
  const int bin = (int)sh[kTopThreads];
  __syncthreads();
  return bin;
}

// Exact top-k of one row by a 3-pass radix select (11 + 11 + 10 bits) over the whole row + gather + bitonic
// sort; ties at the threshold go to the smaller item id.  Used for short rows and as the fallback of the
// candidate filter below.
__device__ void topk_row_full(const float* __restrict__ row, int64_t n_items, int k, int64_t q,
                              int64_t* __restrict__ top_items, float* __restrict__ top_scores, uint32_t* hist,
                              uint32_t* sh, unsigned long long* cand, uint32_t* eq_idx, uint32_t& n_gt, uint32_t& n_eq) {
  const int tid = threadIdx.x;
  const int kk = (int)(k < n_items ? k : n_items);
  // --- pass A: top 11 bits
  for (int b = tid; b < 2048; b += kTopThreads) hist[b] = 0;
  __syncthreads();
  for (int64_t j = tid; j < n_items; j += kTopThreads) atomicAdd(&hist[order_key(row[j]) >> 21], 1u);
  __syncthreads();
  uint32_t above = 0;
  const uint32_t b1 = (uint32_t)pick_bin(hist, 2048, (uint32_t)kk, &above, sh);
  uint32_t want = (uint32_t)kk - above;
  // --- pass B: next 11 bits inside bin b1
  for (int b = tid; b < 2048; b += kTopThreads) hist[b] = 0;
  __syncthreads();
  for (int64_t j = tid; j < n_items; j += kTopThreads) {
    const uint32_t key = order_key(row[j]);
    if ((key >> 21) == b1) atomicAdd(&hist[(key >> 10) & 0x7FFu], 1u);
  }
  __syncthreads();
  const uint32_t b2 = (uint32_t)pick_bin(hist, 2048, want, &above, sh);
  want -= above;
  // --- pass C: last 10 bits
  for (int b = tid; b < 1024; b += kTopThreads) hist[b] = 0;
  __syncthreads();
  const uint32_t prefix = (b1 << 11) | b2;
  for (int64_t j = tid; j < n_items; j += kTopThreads) {
    const uint32_t key = order_key(row[j]);
    if ((key >> 10) == prefix) atomicAdd(&hist[key & 0x3FFu], 1u);
  }
  __syncthreads();
  const uint32_t b3 = (uint32_t)pick_bin(hist, 1024, want, &above, sh);
  const uint32_t need_eq = want - above;          // how many elements equal to the threshold to take
  const uint32_t thr = (prefix << 10) | b3;
  // --- gather: everything above the threshold, and the `need_eq` smallest item ids equal to it
  if (tid == 0) {
    n_gt = 0;
    n_eq = 0;
  }
  __syncthreads();
  for (int64_t j = tid; j < n_items; j += kTopThreads) {
    const uint32_t key = order_key(row[j]);
    if (key > thr) {
      const uint32_t p = atomicAdd(&n_gt, 1u);
      if (p < (uint32_t)kMaxK) cand[p] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - (uint32_t)j);
    } else if (key == thr) {
      const uint32_t p = atomicAdd(&n_eq, 1u);
      if (p < (uint32_t)kEqCap) eq_idx[p] = (uint32_t)j;
    }
  }
  __syncthreads();
  const uint32_t gt = n_gt;
  if (n_eq <= (uint32_t)kEqCap) {
    // rank of each tied id among the ties (O(n_eq^2 / threads), n_eq is tiny for real scores)
    for (uint32_t a = tid; a < n_eq; a += kTopThreads) {
      const uint32_t me = eq_idx[a];
      uint32_t rank = 0;
      for (uint32_t b = 0; b < n_eq; ++b) rank += eq_idx[b] < me;
      if (rank < need_eq) cand[gt + rank] = ((unsigned long long)thr << 32) | (0xFFFFFFFFu - me);
    }
  } else if (tid == 0) {
    // pathological (huge tie, e.g. an all-equal row): first `need_eq` ids in index order
    uint32_t taken = 0;
    for (int64_t j = 0; j < n_items && taken < need_eq; ++j)
      if (order_key(row[j]) == thr) cand[gt + taken++] = ((unsigned long long)thr << 32) | (0xFFFFFFFFu - (uint32_t)j);
  }
  __syncthreads();
  // --- bitonic sort (descending) of the kk candidates, padded to a power of two with 0
  int n2 = 1;
  while (n2 < kk) n2 <<= 1;
  for (int i = kk + tid; i < n2; i += kTopThreads) cand[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= n2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < n2; i += kTopThreads) {
        const int partner = i ^ stride;
        if (partner > i) {
          const bool desc = (i & size) == 0;
          const unsigned long long a = cand[i], b = cand[partner];
          if ((a < b) == desc) {
            cand[i] = b;
            cand[partner] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += kTopThreads) {
    if (i < kk) {
      const uint32_t item = 0xFFFFFFFFu - (uint32_t)(cand[i] & 0xFFFFFFFFull);
      top_items[q * k + i] = item;
      top_scores[q * k + i] = row[item];
    } else {
      top_items[q * k + i] = -1;
      top_scores[q * k + i] = -INFINITY;
    }
  }
}


__device__ __forceinline__ float key_to_float(uint32_t key) {
  return __uint_as_float((key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key);
}

// thr[q] = k-th largest score of the user's first kFuseSample items after masking the training positives among
// them (a lower bound of the row's k-th largest eligible score); -inf when the sample has fewer than k eligible
// items (everything passes, the list overflows and the user takes the exact fallback).
__global__ __launch_bounds__(kTopThreads) void rank_threshold_kernel(const float* __restrict__ sample, int64_t n_query,
                                                                     const int64_t* __restrict__ user_ids, int64_t n_users,
                                                                     const int64_t* __restrict__ user_rowptr,
                                                                     const int32_t* __restrict__ user_items, int k,
                                                                     float* __restrict__ thr) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[kTopThreads + 2];
  __shared__ uint32_t skey[kFuseSample];
  const int tid = threadIdx.x;
  for (int64_t q = blockIdx.x; q < n_query; q += gridDim.x) {
    const float* row = sample + q * kFuseSample;
    for (int j = tid; j < kFuseSample; j += kTopThreads) skey[j] = order_key(row[j]);
    __syncthreads();
    const int64_t uid = user_ids != nullptr ? user_ids[q] : q;
    if (user_rowptr != nullptr && uid >= 0 && uid < n_users) {
      for (int64_t e = user_rowptr[uid] + tid; e < user_rowptr[uid + 1]; e += kTopThreads) {
        const int32_t it = user_items[e];
        if (it >= 0 && it < kFuseSample) skey[it] = 0u;          // below every real score, -inf included
      }
    }
    __syncthreads();
    uint32_t prefix = 0, want = (uint32_t)k, above = 0;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      const int nb = 1 << widths[pass];
      for (int b2 = tid; b2 < nb; b2 += kTopThreads) hist[b2] = 0;
      __syncthreads();
      for (int j = tid; j < kFuseSample; j += kTopThreads) {
        const uint32_t key = skey[j];
        const bool in_prefix = pass == 0 || (key >> (shifts[pass] + widths[pass])) == prefix;
        if (in_prefix) atomicAdd(&hist[(key >> shifts[pass]) & (uint32_t)(nb - 1)], 1u);
      }
      __syncthreads();
      const uint32_t bin = (uint32_t)pick_bin(hist, nb, want, &above, sh);
      want -= above;
      prefix = (prefix << widths[pass]) | bin;
    }
    if (tid == 0) thr[q] = prefix > order_key(-INFINITY) ? key_to_float(prefix) : -INFINITY;
    __syncthreads();
  }
}

// exact top-k of one user from its candidate regions (radix select of the k-th key in LDS, then a bitonic sort of
// the k winners only); status[q] = 1 asks for the fallback
__global__ __launch_bounds__(kTopThreads) void rank_finish_kernel(const unsigned long long* __restrict__ cand_g,
                                                                  const int32_t* __restrict__ counts, int n_regions,
                                                                  int cap_split, int64_t n_query, int k,
                                                                  const int64_t* __restrict__ user_ids, int64_t n_users,
                                                                  const int64_t* __restrict__ user_rowptr,
                                                                  const int32_t* __restrict__ user_items,
                                                                  int64_t* __restrict__ top_items,
                                                                  float* __restrict__ top_scores, int32_t* __restrict__ status) {
  __shared__ unsigned long long cand[kFuseCap];
  __shared__ unsigned long long win[kMaxK];
  __shared__ unsigned long long eq[kEqCap];
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[kTopThreads + 2];
  __shared__ uint32_t n_gt, n_eq;
  __shared__ int s_total, s_bad, s_masked;
  const int tid = threadIdx.x;
  for (int64_t q = blockIdx.x; q < n_query; q += gridDim.x) {
    if (tid == 0) {
      int total = 0, bad = 0;
      for (int r = 0; r < n_regions; ++r) {
        const int c = counts[q * n_regions + r];
        bad |= c > cap_split;
        total += c < cap_split ? c : cap_split;
      }
      s_total = total;
      s_bad = bad || total > kFuseCap || total < k;
      s_masked = 0;
    }
    __syncthreads();
    const int total = s_total;
    if (s_bad) {
      if (tid == 0) status[q] = 1;
      __syncthreads();
      continue;
    }
    if (tid == 0) status[q] = 0;
    // gather: region r's entries land at the prefix sum of the counts before it
    int base = 0;
    for (int r = 0; r < n_regions; ++r) {
      const int c = counts[q * n_regions + r];
      const unsigned long long* src = cand_g + ((int64_t)q * n_regions + r) * cap_split;
      for (int i = tid; i < c; i += kTopThreads) cand[base + i] = src[i];
      base += c;
    }
    __syncthreads();
    // training positives among the candidates sink to the end (key 0 is below every real score)
    const int64_t uid = user_ids != nullptr ? user_ids[q] : q;
    if (user_rowptr != nullptr && uid >= 0 && uid < n_users) {
      const int64_t t0 = user_rowptr[uid], t1 = user_rowptr[uid + 1];
      const int n_train = (int)(t1 - t0);
      if (n_train <= kEqCap * 2) {
        // the user's sorted training row staged in LDS (the tie buffer is free until the select): the binary
        // searches of all candidates then never leave the CU
        int32_t* trow = reinterpret_cast<int32_t*>(eq);
        for (int i = tid; i < n_train; i += kTopThreads) trow[i] = user_items[t0 + i];
        __syncthreads();
        for (int i = tid; i < total; i += kTopThreads) {
          const int32_t j = (int32_t)(0xFFFFFFFFu - (uint32_t)(cand[i] & 0xFFFFFFFFull));
          int lo = 0, hi = n_train;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (trow[mid] < j) lo = mid + 1; else hi = mid;
          }
          if (lo < n_train && trow[lo] == j) {
            cand[i] = 0ull;
            atomicAdd(&s_masked, 1);
          }
        }
      } else {
        for (int i = tid; i < total; i += kTopThreads) {
          const int32_t j = (int32_t)(0xFFFFFFFFu - (uint32_t)(cand[i] & 0xFFFFFFFFull));
          if (is_train_item(user_rowptr, user_items, uid, j)) {
            cand[i] = 0ull;
            atomicAdd(&s_masked, 1);
          }
        }
      }
    }
    __syncthreads();
    if (total - s_masked < k) {                      // cannot happen when the threshold came from >= k eligible items
      if (tid == 0) status[q] = 1;
      __syncthreads();
      continue;
    }
    // k largest of the `total` distinct 64-bit entries: 3-pass radix select on the 32-bit score key (the masked
    // entries carry key 0), the winners gathered, then a bitonic sort of k entries only
    uint32_t prefix = 0, want = (uint32_t)k, above = 0;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      const int nb = 1 << widths[pass];
      for (int b2 = tid; b2 < nb; b2 += kTopThreads) hist[b2] = 0;
      __syncthreads();
      for (int i = tid; i < total; i += kTopThreads) {
        const uint32_t key = (uint32_t)(cand[i] >> 32);
        const bool in_prefix = pass == 0 || (key >> (shifts[pass] + widths[pass])) == prefix;
        if (in_prefix) atomicAdd(&hist[(key >> shifts[pass]) & (uint32_t)(nb - 1)], 1u);
      }
      __syncthreads();
      const uint32_t bin = (uint32_t)pick_bin(hist, nb, want, &above, sh);
      want -= above;
      prefix = (prefix << widths[pass]) | bin;
    }
    const uint32_t thr_key = prefix, need_eq = want;     // take every key > thr_key and `need_eq` of the keys == thr_key
    if (tid == 0) {
      n_gt = 0;
      n_eq = 0;
    }
    __syncthreads();
    for (int i = tid; i < total; i += kTopThreads) {
      const unsigned long long e = cand[i];
      const uint32_t key = (uint32_t)(e >> 32);
      if (key > thr_key) {
        win[atomicAdd(&n_gt, 1u)] = e;                   // < k of them by construction
      } else if (key == thr_key) {
        const uint32_t p = atomicAdd(&n_eq, 1u);
        if (p < (uint32_t)kEqCap) eq[p] = e;
      }
    }
    __syncthreads();
    const uint32_t gt = n_gt, ne = n_eq;
    if (ne > (uint32_t)kEqCap) {                       // a huge tie group at the threshold: exact fallback
      if (tid == 0) status[q] = 1;
      __syncthreads();
      continue;
    }
    for (uint32_t a2 = tid; a2 < ne; a2 += kTopThreads) {     // ties: larger low word = smaller item id first
      const unsigned long long me = eq[a2];
      uint32_t rank = 0;
      for (uint32_t b2 = 0; b2 < ne; ++b2) rank += eq[b2] > me;
      if (rank < need_eq) win[gt + rank] = me;
    }
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    __syncthreads();
    for (int i = k + tid; i < n2; i += kTopThreads) win[i] = 0ull;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < n2; i += kTopThreads) {
          const int partner = i ^ stride;
          if (partner > i) {
            const bool desc = (i & size) == 0;
            const unsigned long long x = win[i], y = win[partner];
            if ((x < y) == desc) {
              win[i] = y;
              win[partner] = x;
            }
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < k; i += kTopThreads) {
      top_items[q * k + i] = (int64_t)(0xFFFFFFFFu - (uint32_t)(win[i] & 0xFFFFFFFFull));
      top_scores[q * k + i] = key_to_float((uint32_t)(win[i] >> 32));
    }
    __syncthreads();
  }
}

constexpr int kSample = 4096;      // prefix of the row that supplies the candidate threshold
constexpr int kCandCap = 4096;     // candidates kept in LDS

// One 256-thread block per user row.  Long rows (the 100K-item catalogue) are not radix-selected in
// full: the k-th largest key of the first 4096 scores is a lower bound of the row's k-th largest, so one
// streaming pass keeps only the scores >= that bound (about k * n / 4096 of them) in LDS, and the exact
// top-k (ties -> smaller item id) is a bitonic sort of those.  More than 4096 candidates (a row whose
// prefix is unrepresentatively low) falls back to the full select: the result is exact either way.
__global__ __launch_bounds__(kTopThreads) void topk_masked_kernel(float* __restrict__ scores, int64_t n_query,
                                                                  int64_t n_items, const int64_t* __restrict__ user_ids,
                                                                  int64_t n_users,
                                                                  const int64_t* __restrict__ user_rowptr,
                                                                  const int32_t* __restrict__ user_items, int k,
                                                                  int64_t* __restrict__ top_items,
                                                                  float* __restrict__ top_scores) {
  __shared__ uint32_t hist[2048];
  __shared__ uint32_t sh[kTopThreads + 2];
  __shared__ unsigned long long cand[kCandCap];   // (key << 32) | (0xFFFFFFFF - item); first kSample words double as the sample
  __shared__ uint32_t eq_idx[kEqCap];
  __shared__ uint32_t n_gt, n_eq, n_cand;
  const int tid = threadIdx.x;
  for (int64_t q = blockIdx.x; q < n_query; q += gridDim.x) {
    float* row = scores + q * n_items;
    const int64_t uid = user_ids != nullptr ? user_ids[q] : q;
    if (user_rowptr != nullptr && uid >= 0 && uid < n_users) {
      for (int64_t e = user_rowptr[uid] + tid; e < user_rowptr[uid + 1]; e += kTopThreads) {
        const int32_t it = user_items[e];
        if (it >= 0 && it < n_items) row[it] = -INFINITY;
      }
    }
    __syncthreads();
    const int kk = (int)(k < n_items ? k : n_items);
    bool done = false;
    if (n_items >= 4 * kSample) {
      // --- threshold: kk-th largest key of the prefix (3-pass radix select over the LDS copy)
      uint32_t* skey = reinterpret_cast<uint32_t*>(cand);
      for (int j = tid; j < kSample; j += kTopThreads) skey[j] = order_key(row[j]);
      uint32_t prefix = 0, want = (uint32_t)kk, above = 0;
      const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
      for (int pass = 0; pass < 3; ++pass) {
        const int nb = 1 << widths[pass];
        for (int b2 = tid; b2 < nb; b2 += kTopThreads) hist[b2] = 0;
        __syncthreads();
        for (int j = tid; j < kSample; j += kTopThreads) {
          const uint32_t key = skey[j];
          const bool in_prefix = pass == 0 || (key >> (shifts[pass] + widths[pass])) == prefix;
          if (in_prefix) atomicAdd(&hist[(key >> shifts[pass]) & (uint32_t)(nb - 1)], 1u);
        }
        __syncthreads();
        const uint32_t bin = (uint32_t)pick_bin(hist, nb, want, &above, sh);
        want -= above;
        prefix = (prefix << widths[pass]) | bin;
      }
      const uint32_t thr0 = prefix;                 // exact kk-th largest key of the prefix
      if (tid == 0) n_cand = 0;
      __syncthreads();
      // --- one pass over the row: keep everything >= thr0 (16-B loads, four in flight per thread)
      auto keep = [&](float v, int64_t j) {
        const uint32_t key = order_key(v);
        if (key >= thr0) {
          const uint32_t p = atomicAdd(&n_cand, 1u);
          if (p < (uint32_t)kCandCap) cand[p] = ((unsigned long long)key << 32) | (0xFFFFFFFFu - (uint32_t)j);
        }
      };
      if ((n_items & 3) == 0) {
        const float4* row4 = reinterpret_cast<const float4*>(row);
        const int64_t n4 = n_items >> 2;
        for (int64_t j0 = 0; j0 < n4; j0 += 4 * kTopThreads) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t j4 = j0 + u * kTopThreads + tid;
            v[u] = j4 < n4 ? row4[j4] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t j = 4 * (j0 + u * kTopThreads + tid);
            if (j < n_items) {
              keep(v[u].x, j);
              keep(v[u].y, j + 1);
              keep(v[u].z, j + 2);
              keep(v[u].w, j + 3);
            }
          }
        }
      } else {
        for (int64_t j = tid; j < n_items; j += kTopThreads) keep(row[j], j);
      }
      __syncthreads();
      const uint32_t nc = n_cand;
      if (nc <= (uint32_t)kCandCap) {
        int n2 = 1;
        while (n2 < (int)nc) n2 <<= 1;
        for (int i = (int)nc + tid; i < n2; i += kTopThreads) cand[i] = 0ull;
        __syncthreads();
        for (int size = 2; size <= n2; size <<= 1) {
          for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < n2; i += kTopThreads) {
              const int partner = i ^ stride;
              if (partner > i) {
                const bool desc = (i & size) == 0;
                const unsigned long long x = cand[i], y = cand[partner];
                if ((x < y) == desc) {
                  cand[i] = y;
                  cand[partner] = x;
                }
              }
            }
            __syncthreads();
          }
        }
        for (int i = tid; i < k; i += kTopThreads) {
          if (i < kk) {
            const uint32_t item = 0xFFFFFFFFu - (uint32_t)(cand[i] & 0xFFFFFFFFull);
            top_items[q * k + i] = item;
            top_scores[q * k + i] = row[item];
          } else {
            top_items[q * k + i] = -1;
            top_scores[q * k + i] = -INFINITY;
          }
        }
        done = true;
      }
      __syncthreads();
    }
    if (!done) topk_row_full(row, n_items, k, q, top_items, top_scores, hist, sh, cand, eq_idx, n_gt, n_eq);
    __syncthreads();
  }
}

bool rank_dim_supported(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

template <int D>
int32_t launch_score(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                     const float* item_emb, int64_t n_items, float* scores, hipStream_t s) {
  const int64_t iblocks = (n_items + RShape<D>::ITEMS_PER_BLOCK - 1) / RShape<D>::ITEMS_PER_BLOCK;
  const int64_t total_tiles = (n_query + kTile - 1) / kTile;
  int64_t nsplit = iblocks * 2 <= 512 ? 512 / iblocks : (total_tiles + 63) / 64;
  if (nsplit > total_tiles) nsplit = total_tiles;
  if (nsplit < 1) nsplit = 1;
  const int64_t tps = (total_tiles + nsplit - 1) / nsplit;
  nsplit = (total_tiles + tps - 1) / tps;
  hipLaunchKernelGGL((score_rows_kernel<D>), dim3((unsigned)(iblocks * nsplit)), dim3(256), 0, s, user_emb, user_ids,
                     n_query, n_users, item_emb, n_items, tps, (int)nsplit, scores);
  return GCR_LAUNCH_STATUS();
}


struct FusePlan {
  int nsplit;
  int64_t tiles_per_split;
  int64_t ublocks;
  int cap_split;
};

FusePlan plan_fuse(int64_t n_query, int64_t n_items, int users_per_block) {
  FusePlan p;
  p.ublocks = (n_query + users_per_block - 1) / users_per_block;
  const int64_t total_tiles = (n_items + kTileJ - 1) / kTileJ;
  // >= 3 rounds of 512 resident blocks, but never fewer than 64 tiles per block and at most 8 splits
  // (the candidate regions are sized per split)
  int64_t nsplit = (3 * 512 + p.ublocks - 1) / p.ublocks;
  if (nsplit > 8) nsplit = 8;
  if (nsplit > total_tiles / 64) nsplit = total_tiles / 64 > 0 ? total_tiles / 64 : 1;
  p.tiles_per_split = (total_tiles + nsplit - 1) / nsplit;
  p.nsplit = (int)((total_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
  p.cap_split = kFuseCap / (2 * p.nsplit);
  return p;
}

template <int D>
int32_t launch_fused(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                     const float* item_emb, int64_t n_items, const int64_t* user_rowptr, const int32_t* user_items,
                     int k, int64_t* top_items, float* top_scores, int32_t* status, void* workspace, hipStream_t s) {
  constexpr int UPB = ShapeB3<D>::ANCHORS_PER_BLOCK;
  const FusePlan p = plan_fuse(n_query, n_items, UPB);
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  float* sample = reinterpret_cast<float*>(w);
  w += (size_t)n_query * kFuseSample * sizeof(float);
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(w);
  w += (size_t)n_query * kFuseCap * sizeof(unsigned long long);
  float* thr = reinterpret_cast<float*>(w);
  w += (size_t)n_query * sizeof(float);
  int32_t* counts = reinterpret_cast<int32_t*>(w);
  // pass 0: dense sample scores (one split: 128 tiles per block)
  hipLaunchKernelGGL((rank_fused_b3_kernel<D, 0>), dim3((unsigned)p.ublocks), dim3(256), 0, s, user_emb, user_ids, n_query,
                     n_users, item_emb, n_items, (int64_t)kFuseSample, 1, (int64_t)(kFuseSample / kTileJ), sample,
                     kFuseSample, (const float*)nullptr, user_rowptr, user_items, cand, counts, 0);
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  const int64_t tb = n_query < 65536 ? n_query : 65536;
  hipLaunchKernelGGL(rank_threshold_kernel, dim3((unsigned)tb), dim3(kTopThreads), 0, s, sample, n_query, user_ids,
                     n_users, user_rowptr, user_items, k, thr);
  st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  // pass 1: all items, candidates only
  hipLaunchKernelGGL((rank_fused_b3_kernel<D, 1>), dim3((unsigned)(p.ublocks * p.nsplit)), dim3(256), 0, s, user_emb,
                     user_ids, n_query, n_users, item_emb, n_items, n_items, p.nsplit, p.tiles_per_split, sample,
                     kFuseSample, thr, user_rowptr, user_items, cand, counts, p.cap_split);
  st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(rank_finish_kernel, dim3((unsigned)tb), dim3(kTopThreads), 0, s, cand, counts, 2 * p.nsplit,
                     p.cap_split, n_query, k, user_ids, n_users, user_rowptr, user_items, top_items, top_scores, status);
  return GCR_LAUNCH_STATUS();
}


// Per query user and cut-off n: #hits in the first n ranked items, DCG = sum over hit positions p of 1 / log2(p + 2),
// IDCG = sum_{p < min(|test|, n)} 1 / log2(p + 2) — the per-user terms of ncl.py:133-163 (Metric.hits / NDCG); the
// means over users are the caller's.  One thread per user; test items sorted per user (binary search).
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int64_t* __restrict__ top_items, int64_t n_query, int k,
                                                           const int64_t* __restrict__ test_rowptr,
                                                           const int32_t* __restrict__ test_items,
                                                           const int32_t* __restrict__ cutoffs, int n_cut,
                                                           int32_t* __restrict__ hits, double* __restrict__ dcg,
                                                           double* __restrict__ idcg) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_query) return;
  const int64_t lo0 = test_rowptr[q], hi0 = test_rowptr[q + 1];
  const int64_t n_test = hi0 - lo0;
  int h = 0;
  double d = 0.0, ideal = 0.0;
  int c = 0;
  for (int p = 0; p <= k && c < n_cut; ++p) {
    while (c < n_cut && cutoffs[c] == p) {            // cut-offs ascending: emit the running sums at n = p
      hits[q * n_cut + c] = h;
      dcg[q * n_cut + c] = d;
      idcg[q * n_cut + c] = ideal;
      ++c;
    }
    if (p == k) break;
    const double gain = 1.0 / log2((double)(p + 2));
    if (p < n_test) ideal += gain;
    const int64_t it = top_items[q * k + p];
    if (it >= 0) {
      int64_t lo = lo0, hi = hi0;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (test_items[mid] < it) lo = mid + 1; else hi = mid;
      }
      if (lo < hi0 && test_items[lo] == it) {
        // the reference counts hits on SETS (a repeated item counts once, ncl.py:136) but adds a DCG gain at every
        // position holding a test item (ncl.py:157); hits are rare, so the look-back scan is cheap
        bool seen = false;
        for (int pp = 0; pp < p && !seen; ++pp) seen = top_items[q * k + pp] == it;
        h += seen ? 0 : 1;
        d += gain;
      }
    }
  }
  for (; c < n_cut; ++c) {                            // cut-offs beyond k see the whole list
    hits[q * n_cut + c] = h;
    dcg[q * n_cut + c] = d;
    double x = ideal;
    for (int p = k; p < cutoffs[c] && p < n_test; ++p) x += 1.0 / log2((double)(p + 2));
    idcg[q * n_cut + c] = x;
  }
}

}  // namespace

extern "C" int32_t gcr_rank_metrics(const int64_t* top_items, int64_t n_query, int32_t k, const int64_t* test_rowptr,
                                    const int32_t* test_items_sorted, const int32_t* cutoffs, int32_t n_cut,
                                    int32_t* hits, double* dcg, double* idcg, void* stream) {
  GCR_CHECK_ARG(n_query >= 0 && k >= 1 && n_cut >= 1 && n_cut <= 64);
  if (n_query == 0) return GCR_OK;
  GCR_CHECK_ARG(top_items && test_rowptr && cutoffs && hits && dcg && idcg);
  hipLaunchKernelGGL(rank_metrics_kernel, dim3((unsigned)((n_query + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     top_items, n_query, k, test_rowptr, test_items_sorted, cutoffs, n_cut, hits, dcg, idcg);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_rank_fused_supported(int64_t n_items, int32_t d, int32_t k) {
  return (d == 32 || d == 64 || d == 128) && n_items >= 4 * kFuseSample && k >= 1 && k <= kMaxK ? 1 : 0;
}

extern "C" int64_t gcr_rank_fused_workspace_bytes(int64_t n_query) {
  if (n_query <= 0) return 0;
  return n_query * ((int64_t)kFuseSample * 4 + (int64_t)kFuseCap * 8 + 4 + 16 * 4);
}

extern "C" int32_t gcr_rank_fused_f32(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                                      const float* item_emb, int64_t n_items, int32_t d, const int64_t* user_rowptr,
                                      const int32_t* user_items_sorted, int32_t k, int64_t* top_items, float* top_scores,
                                      int32_t* status, void* workspace, void* stream) {
  GCR_CHECK_ARG(n_query >= 0 && n_users >= 1 && n_items >= 1 && n_items < (1ll << 31));
  if (!gcr_rank_fused_supported(n_items, d, k)) return GCR_EUNSUPPORTED;
  if (n_query == 0) return GCR_OK;
  GCR_CHECK_ARG(user_emb && item_emb && top_items && top_scores && status && workspace);
  GCR_CHECK_ARG((user_rowptr == nullptr) == (user_items_sorted == nullptr));
  GCR_CHECK_ARG(n_query < (1ll << 24));
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 32: return launch_fused<32>(user_emb, user_ids, n_query, n_users, item_emb, n_items, user_rowptr, user_items_sorted,
                                     k, top_items, top_scores, status, workspace, s);
    case 64: return launch_fused<64>(user_emb, user_ids, n_query, n_users, item_emb, n_items, user_rowptr, user_items_sorted,
                                     k, top_items, top_scores, status, workspace, s);
    default: return launch_fused<128>(user_emb, user_ids, n_query, n_users, item_emb, n_items, user_rowptr,
                                      user_items_sorted, k, top_items, top_scores, status, workspace, s);
  }
}

extern "C" int32_t gcr_score_rows_f32(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                                      const float* item_emb, int64_t n_items, int32_t d, float* scores, void* stream) {
  GCR_CHECK_ARG(n_query >= 0 && n_users >= 1 && n_items >= 1);
  if (!rank_dim_supported(d)) return GCR_EUNSUPPORTED;
  if (n_query == 0) return GCR_OK;
  GCR_CHECK_ARG(user_emb && item_emb && scores);
  GCR_CHECK_ARG(n_query * n_items < (1ll << 40));
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 32: return launch_score<32>(user_emb, user_ids, n_query, n_users, item_emb, n_items, scores, s);
    case 64: return launch_score<64>(user_emb, user_ids, n_query, n_users, item_emb, n_items, scores, s);
    case 128: return launch_score<128>(user_emb, user_ids, n_query, n_users, item_emb, n_items, scores, s);
    default: return launch_score<256>(user_emb, user_ids, n_query, n_users, item_emb, n_items, scores, s);
  }
}

extern "C" int32_t gcr_topk_masked_f32(float* scores, int64_t n_query, int64_t n_items, const int64_t* user_ids,
                                       int64_t n_users, const int64_t* user_rowptr, const int32_t* user_items_sorted,
                                       int32_t k, int64_t* top_items, float* top_scores, void* stream) {
  GCR_CHECK_ARG(n_query >= 0 && n_items >= 1 && n_items < (1ll << 32) - 1 && k >= 1 && k <= kMaxK);
  if (n_query == 0) return GCR_OK;
  GCR_CHECK_ARG(scores && top_items && top_scores);
  GCR_CHECK_ARG((user_rowptr == nullptr) == (user_items_sorted == nullptr));
  const int64_t blocks = n_query < 65536 ? n_query : 65536;
  hipLaunchKernelGGL(topk_masked_kernel, dim3((unsigned)blocks), dim3(kTopThreads), 0, (hipStream_t)stream, scores,
                     n_query, n_items, user_ids, n_users, user_rowptr, user_items_sorted, k, top_items, top_scores);
  return GCR_LAUNCH_STATUS();
}
