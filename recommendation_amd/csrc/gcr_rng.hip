// Counter-based RNG kernels (gfx950): uniform negative sampler with rejection against the
// user's sorted training row, and Bernoulli edge-dropout bitmaps.
//
// Replaces the Python-level samplers of ncl.py:91-114 (rejection, <= 100 retries),
// gcl.py:111-125 / ssl4rec.py:33-50 (rejection until success, n_negs per positive),
// lightgcn.py:91-94 (torch.randint, no rejection) and the masks of gcl.py:22-25 (EdgeRemoving,
// `rand >= pe`) / buir.py:300-309 (Bernoulli + rescale).  The reference RNG streams are unseeded
// python / numpy / torch generators, so bit-exactness is defined against the CPU restatement
// with the same Philox-4x32-10 counters (oracle/oracle_np.py, oracle/oracle.c); what is shared
// with the reference is the contract: uniform over items, never a training positive.
#include "gcr_common.h"
#include "gcr_philox.h"

namespace {

constexpr uint32_t kStreamNeg = 0x4E454753u;   // 'NEGS'
constexpr uint32_t kStreamEdge = 0x45444745u;  // 'EDGE'

__device__ __forceinline__ uint32_t pick(const U4& r, int w) { return w == 0 ? r.x : (w == 1 ? r.y : (w == 2 ? r.z : r.w)); }

__global__ __launch_bounds__(256) void neg_sample_kernel(const int64_t* __restrict__ user_rowptr,
                                                         const int32_t* __restrict__ user_items_sorted,
                                                         const int64_t* __restrict__ u_idx, int64_t n_slots,
                                                         int n_negs, int64_t n_users, uint32_t num_items, uint64_t seed,
                                                         uint64_t offset, int max_trials, int64_t* __restrict__ out) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots;
       s += (int64_t)gridDim.x * blockDim.x) {
    const int64_t user = u_idx[s / n_negs];
    int64_t res = -1;
    if (user >= 0 && user < n_users) {
      const int64_t lo0 = user_rowptr[user], hi0 = user_rowptr[user + 1];
      const uint64_t slot = offset + (uint64_t)s;
      const int draws = max_trials > 0 ? max_trials : 1;
      U4 r{0, 0, 0, 0};
      for (int t = 0; t < draws; ++t) {
        if ((t & 3) == 0) r = philox4x32_10(U4{(uint32_t)slot, (uint32_t)(slot >> 32), (uint32_t)(t >> 2), kStreamNeg}, k0, k1);
        const int32_t cand = (int32_t)__umulhi(pick(r, t & 3), num_items);
        if (max_trials == 0) {
          res = cand;
          break;
        }
        int64_t lo = lo0, hi = hi0;
        while (lo < hi) {  // lower_bound in the user's sorted positives
          const int64_t mid = (lo + hi) >> 1;
          if (user_items_sorted[mid] < cand) lo = mid + 1;
          else hi = mid;
        }
        if (!(lo < hi0 && user_items_sorted[lo] == cand)) {
          res = cand;
          break;
        }
      }
    }
    out[s] = res;
  }
}

// one thread per 32-bit word of the bitmap; edge e uses word (id & 3) of philox(block = id >> 2)
__global__ __launch_bounds__(256) void edge_mask_kernel(int64_t nnz, float pe, uint64_t seed,
                                                        const int64_t* __restrict__ edge_id,
                                                        uint32_t* __restrict__ bits, int64_t n_words) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (int64_t)gridDim.x * blockDim.x) {
    uint32_t word = 0;
    if (edge_id == nullptr) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint64_t blk = (uint64_t)w * 8 + q;
        const U4 r = philox4x32_10(U4{(uint32_t)blk, (uint32_t)(blk >> 32), 0u, kStreamEdge}, k0, k1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float u = (float)(pick(r, t) >> 8) * 5.9604644775390625e-8f;  // 24-bit grid like torch.rand
          word |= (uint32_t)(u >= pe) << (q * 4 + t);
        }
      }
    } else {
      for (int bit = 0; bit < 32; ++bit) {
        const int64_t e = w * 32 + bit;
        if (e >= nnz) break;
        const uint64_t id = (uint64_t)edge_id[e];
        const uint64_t blk = id >> 2;
        const U4 r = philox4x32_10(U4{(uint32_t)blk, (uint32_t)(blk >> 32), 0u, kStreamEdge}, k0, k1);
        const float u = (float)(pick(r, (int)(id & 3)) >> 8) * 5.9604644775390625e-8f;
        word |= (uint32_t)(u >= pe) << bit;
      }
    }
    const int64_t rem = nnz - w * 32;
    if (rem < 32) word &= rem <= 0 ? 0u : ((1u << rem) - 1u);
    bits[w] = word;
  }
}

int grid_for(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int32_t gcr_neg_sample(const int64_t* user_rowptr, const int32_t* user_items_sorted, const int64_t* u_idx,
                                  int64_t batch, int32_t n_negs, int64_t n_users, int64_t num_items, uint64_t seed,
                                  uint64_t offset, int32_t max_trials, int64_t* out, void* stream) {
  GCR_CHECK_ARG(batch >= 0 && n_negs >= 1 && max_trials >= 0 && n_users >= 0);
  GCR_CHECK_ARG(num_items >= 1 && num_items < (1ll << 31));
  if (batch == 0) return GCR_OK;
  GCR_CHECK_ARG(user_rowptr && u_idx && out && (user_items_sorted || max_trials == 0));
  const int64_t n_slots = batch * n_negs;
  hipLaunchKernelGGL(neg_sample_kernel, dim3(grid_for(n_slots, 256)), dim3(256), 0, (hipStream_t)stream, user_rowptr,
                     user_items_sorted, u_idx, n_slots, n_negs, n_users, (uint32_t)num_items, seed, offset, max_trials,
                     out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_edge_mask_bits(int64_t nnz, float pe, uint64_t seed, const int64_t* edge_id, uint32_t* bits,
                                      void* stream) {
  GCR_CHECK_ARG(nnz >= 0 && pe >= 0.f && pe <= 1.f);
  if (nnz == 0) return GCR_OK;
  GCR_CHECK_ARG(bits != nullptr);
  const int64_t n_words = (nnz + 31) / 32;
  hipLaunchKernelGGL(edge_mask_kernel, dim3(grid_for(n_words, 256)), dim3(256), 0, (hipStream_t)stream, nnz, pe, seed,
                     edge_id, bits, n_words);
  return GCR_LAUNCH_STATUS();
}
