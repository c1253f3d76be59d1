// InfoNCE / prototype-contrast logits with fused temperature scale and row logsumexp on the
// fp32 MFMA of gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD, ~155 TF dense).
//
// Replaces the materialised score matrices of
//   InfoNCE            ncl.py:125-130, ssl4rec.py:19-23   (view1 @ view2.T / t -> log_softmax diag)
//   info_nce_loss      gcl.py:28-35                       (sim, sim.T cross-entropies)
//   ssl_layer_loss     ncl.py:358-367                     (B anchors x ALL layer-0 rows, exp-sum)
//   ProtoNCE_loss      ncl.py:369-375, batch_softmax_loss ssl4rec.py:25-30
// The M x N logits never leave the register file.
//
// Mapping.  The score tile is computed TRANSPOSED: MFMA "A" = 32 streamed table rows j, MFMA "B" =
// 32 stationary anchors i, so in the 32x32 accumulator a lane owns ONE anchor (column i = lane&31)
// and its 16 registers are 16 different table rows: the online max / exp2-sum over j is in-lane
// work, no cross-lane traffic until one final lane<->lane+32 merge.  The K (= d) dimension is
// split between the two lane halves (half h covers features [h*d/2, (h+1)*d/2)), so each lane's
// operand slice is contiguous in memory: 16-byte global loads for the stationary anchors, and
// conflict-free ds_read_b128 (row stride padded by 16 B) for the streamed tile.
// A wave keeps NT x 32 anchors (pre-multiplied by 1/||a|| * 1/tau * log2(e)) in registers and walks
// a column range of the table; table tiles of 32 rows are staged global -> registers -> LDS
// (normalised on the way) one tile ahead of the MFMAs (double buffer, one barrier per tile).
// Grid = anchor blocks x column splits; split partials (max, sum) are merged by a second kernel.
#include <stdlib.h>

#include <type_traits>

#include "gcr_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileJ = 32;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -1.0e30f;

template <int D>
struct Shape {
  static constexpr int KH = D / 2;                       // k-steps per 32x32 tile (2 k per MFMA)
  static constexpr int STRIDE = D + 4;                   // floats; +16 B pad -> conflict-free b128
  static constexpr int NT = D <= 128 ? 2 : 1;            // anchor tiles of 32 per wave
  static constexpr int NLD = (kTileJ * D / 4) / 256;     // float4 staged per thread and tile
  static constexpr int ANCHORS_PER_BLOCK = 4 * 32 * NT;
};

// C/D layout of the 32x32 accumulator: register r of lane (col = lane&31, h = lane>>5) is row
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Branch-free staging load (keeps the tile loop one scheduling region): rows past the table are
// clamped to the last row and multiplied by 0.
template <int D>
__device__ __forceinline__ void stage_load(const float* __restrict__ b, const float* __restrict__ b_scale,
                                           int64_t n_rows, int64_t j0, int tid, float4 (&regs)[Shape<D>::NLD],
                                           float mult = 1.0f) {
#pragma unroll
  for (int u = 0; u < Shape<D>::NLD; ++u) {
    const int idx = tid + 256 * u;
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    const int64_t j = j0 + row;
    const int64_t jj = j < n_rows ? j : n_rows - 1;
    float4 v = *reinterpret_cast<const float4*>(b + jj * D + 4 * c4);
    float s = b_scale != nullptr ? b_scale[jj] * mult : mult;
    s = j < n_rows ? s : 0.f;
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    regs[u] = v;
  }
}

template <int D>
__device__ __forceinline__ void stage_store(float* __restrict__ tile, int tid, const float4 (&regs)[Shape<D>::NLD]) {
#pragma unroll
  for (int u = 0; u < Shape<D>::NLD; ++u) {
    const int idx = tid + 256 * u;
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    *reinterpret_cast<float4*>(tile + row * Shape<D>::STRIDE + 4 * c4) = regs[u];
  }
}

// stationary operand: lane (i = lane&31, h) holds features [h*KH, (h+1)*KH) of its row, pre-scaled
template <int D>
__device__ __forceinline__ void load_stationary(const float* __restrict__ a, const float* __restrict__ a_scale,
                                                int64_t m_rows, int64_t row, int h, float mult,
                                                float (&frag)[Shape<D>::KH]) {
  const bool valid = row < m_rows;
  const float s = valid ? (a_scale != nullptr ? a_scale[row] : 1.0f) * mult : 0.f;
  const float* p = a + (valid ? row : 0) * D + h * Shape<D>::KH;
#pragma unroll
  for (int q = 0; q < Shape<D>::KH / 4; ++q) {
    float4 v = valid ? *reinterpret_cast<const float4*>(p + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    frag[4 * q + 0] = v.x * s;
    frag[4 * q + 1] = v.y * s;
    frag[4 * q + 2] = v.z * s;
    frag[4 * q + 3] = v.w * s;
  }
}

// S^T tile: acc[t][r] = log2-domain logit of (table row acc_row(r,h), anchor tile t / lane&31)
template <int D>
__device__ __forceinline__ void score_tile(const float* __restrict__ tile, int i32, int h,
                                           const float (&bfrag)[Shape<D>::NT][Shape<D>::KH],
                                           f32x16 (&acc)[Shape<D>::NT]) {
#pragma unroll
  for (int t = 0; t < Shape<D>::NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* base = tile + i32 * Shape<D>::STRIDE + h * Shape<D>::KH;
#pragma unroll
  for (int q = 0; q < Shape<D>::KH / 4; ++q) {
    const float4 av = *reinterpret_cast<const float4*>(base + 4 * q);
    const float ae[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < Shape<D>::NT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
  }
}

// online-softmax update with one finished score tile; `rem` = table rows left from this lane's
// first accumulator row (n_rows - j0 - 4h): rows at or past it are masked out (ragged last tile).
// EXD: additionally the row whose in-tile offset (acc_row without the 4h) equals xr[t] is left out of
// anchor tile t's sums (the excluded diagonal pair; xr[t] < 0: none in this tile).
template <int NT, bool MASK, bool EXD = false>
__device__ __forceinline__ void lse_update(const f32x16 (&acc)[NT], int rem, float (&m_run)[NT], float (&l_run)[NT],
                                           const int* xr = nullptr) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = (r & 3) + 8 * (r >> 2);
      const bool keep = (!MASK || rc < rem) && !(EXD && rc == xr[t]);
      v[r] = keep ? acc[t][r] : -INFINITY;
    }
    float tmax = v[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, v[r]);
    const float m_new = fmaxf(m_run[t], tmax);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += __builtin_amdgcn_exp2f(v[r] - m_new);
    l_run[t] = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m_new) + sum;
    m_run[t] = m_new;
  }
}

// log2-domain pieces of the BCE-with-logits element (lightgcn.py:109-113): for the log2-domain score s2 = s log2 e
//   softplus(s) = ln2 * (max(s2, 0) + log2(1 + u)),  sigmoid(s) = s2 >= 0 ? 1 / (1 + u) : u / (1 + u),  u = 2^-|s2|.
// log2(1 + u) is compensated for the rounding of v = 1 + u (e = u - (v - 1) exactly; log2(v + e) = log2 v + (e / v) log2 e),
// so a strongly negative score keeps its ~2^s2 instead of 0.  The correction is taken as e log2 e: leaving the 1 / v out
// changes it by e u / v log2 e <= 2^-24 u log2 e, i.e. 2^-23 of log2(1 + u) at most — and the row-sum launch, which needs
// no sigmoid, then needs no reciprocal either (one quarter-rate instruction and a multiply per score less).
// A masked score (-inf) gives softplus = 0 and sigmoid = 0.
__device__ __forceinline__ void bce_terms(float s2, float& softplus2, float& sig) {
  const float u = __builtin_amdgcn_exp2f(-fabsf(s2));
  const float v = 1.0f + u;
  const float e = u - (v - 1.0f);
  softplus2 = fmaxf(s2, 0.f) + fmaf(e, kLog2e, __builtin_amdgcn_logf(v));
  const float rc = __builtin_amdgcn_rcpf(v);
  sig = s2 >= 0.f ? rc : u * rc;
}

// sigmoid alone (the backward's probability): 1 / (1 + 2^-s2) holds for either sign — 2^-s2 = inf gives 0, a masked score
// (-inf) gives 0, and 1 + 2^-s2 rounds at 2^-24 of itself whatever its size — three instructions instead of six.
__device__ __forceinline__ float bce_sigmoid(float s2) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-s2));
}

// BCE forward on the f32 engine: l_run += sum over the tile's live rows of softplus2
template <int NT, bool MASK>
__device__ __forceinline__ void bce_update(const f32x16 (&acc)[NT], int rem, float (&l_run)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = (r & 3) + 8 * (r >> 2);
      float sp, sg;
      bce_terms((!MASK || rc < rem) ? acc[t][r] : -INFINITY, sp, sg);
      sum += sp;
    }
    l_run[t] += sum;
  }
}

// in-tile offset of table row `i` for lane half h of the tile starting at j0 (matches (r&3)+8(r>>2) of
// the register that holds it), or -1 when the row is not one of this lane's 16
__device__ __forceinline__ int diag_offset(int64_t i, int64_t j0, int h) {
  const int64_t d = i - j0 - 4 * h;
  return (d >= 0 && d < 28) ? (int)d : -1;
}

__device__ __forceinline__ int rows_left(int64_t n_rows, int64_t j0, int h) {
  const int64_t rem = n_rows - j0 - 4 * h;
  return (int)(rem > 64 ? 64 : (rem < 0 ? 0 : rem));
}

// d <= 64 runs 3 blocks per CU: two symmetric waves on one SIMD fall into lock-step (their MFMA
// phases and exp2 phases coincide, scripts/exp_infonce.hip), a third wave breaks it (109 -> 126 TF
// on the bare tile loop); needs VGPRs <= 168.  (Folding a logit bound into the MFMA C operand to
// drop the running max was tried: not faster, and it costs ~1e-5 of the lse - pos difference.)
//
// COLSUM (gcl.py:34, the `sim.T` cross-entropy): the same pass also accumulates, for every table row
// j, sum_i exp2(t_ij - col_bound2) over the anchors: 32 extra exp2 per tile, a 5-step DPP row
// reduction per accumulator register (lane 31 / 63 end up with the two half-wave sums) and one
// float atomic per (wave, table row).  It needs a known logit bound (unit-norm rows: |t| <= scale2),
// and replaces a whole second pass with the roles swapped.
__device__ __forceinline__ float half_wave_sum_to_last_lane(float v) {
  // inclusive DPP scan inside each row of 16, then row 0 -> row 1 and row 2 -> row 3 broadcast-add:
  // lane 31 = sum(lanes 0..31), lane 63 = sum(lanes 32..63)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));
  return v;
}

template <int D, bool COLSUM, bool EXD = false, bool BCE = false>
__global__ __launch_bounds__(256, (D <= 64 ? 3 : 2)) void infonce_fwd_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ a_scale, int64_t m_rows,
                                                             const float* __restrict__ b,
                                                             const float* __restrict__ b_scale, int64_t n_rows,
                                                             float scale2, int nsplit, int64_t tiles_per_split,
                                                             float2* __restrict__ part, float* __restrict__ col_sum,
                                                             float col_bound2) {
  using S = Shape<D>;
  __shared__ __align__(16) float lds[2][kTileJ * S::STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t i0 = (mblk * 4 + wave) * (32 * S::NT);

  float bfrag[S::NT][S::KH];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) load_stationary<D>(a, a_scale, m_rows, i0 + 32 * t + i32, h, scale2, bfrag[t]);

  float m_run[S::NT], l_run[S::NT], a_valid[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    m_run[t] = kNegBig;
    l_run[t] = 0.f;
    a_valid[t] = (i0 + 32 * t + i32 < m_rows) ? 1.0f : 0.f;   // padding anchors add nothing to a column
  }

  const int64_t total_tiles = (n_rows + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  float4 regs[S::NLD];
  if (tile0 < tile1) {
    stage_load<D>(b, b_scale, n_rows, tile0 * kTileJ, tid, regs);
    stage_store<D>(lds[0], tid, regs);
  }
  __syncthreads();
  for (int64_t tt = tile0; tt < tile1; ++tt) {
    const int cur = (int)((tt - tile0) & 1);
    // always stage a tile (the last iteration re-stages its own): keeps the loop body branch-free
    const int64_t nxt = tt + 1 < tile1 ? tt + 1 : tt;
    stage_load<D>(b, b_scale, n_rows, nxt * kTileJ, tid, regs);
    f32x16 acc[S::NT];
    score_tile<D>(lds[cur], i32, h, bfrag, acc);
    if (BCE) {                                      // sum of softplus instead of the online logsumexp
      if ((tt + 1) * kTileJ > n_rows)
        bce_update<S::NT, true>(acc, rows_left(n_rows, tt * kTileJ, h), l_run);
      else
        bce_update<S::NT, false>(acc, 64, l_run);
    } else if (EXD) {                               // diagonal pair (i, j = i) left out of the sums
      int xr[S::NT];
#pragma unroll
      for (int t = 0; t < S::NT; ++t) xr[t] = diag_offset(i0 + 32 * t + i32, tt * kTileJ, h);
      lse_update<S::NT, true, true>(acc, rows_left(n_rows, tt * kTileJ, h), m_run, l_run, xr);
    } else if ((tt + 1) * kTileJ > n_rows)   // only the table's ragged last tile pays for the row mask
      lse_update<S::NT, true>(acc, rows_left(n_rows, tt * kTileJ, h), m_run, l_run);
    else
      lse_update<S::NT, false>(acc, 64, m_run, l_run);
    if (COLSUM) {
      const int64_t j0 = tt * kTileJ;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = 0.f;
#pragma unroll
        for (int t = 0; t < S::NT; ++t) e += a_valid[t] * __builtin_amdgcn_exp2f(acc[t][r] - col_bound2);
        e = half_wave_sum_to_last_lane(e);
        const int64_t j = j0 + acc_row(r, h);
        if (i32 == 31 && j < n_rows) atomicAdd(col_sum + j, e);
      }
    }
    stage_store<D>(lds[cur ^ 1], tid, regs);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const float m_o = __shfl_xor(m_run[t], 32, 64), l_o = __shfl_xor(l_run[t], 32, 64);
    const float m = fmaxf(m_run[t], m_o);
    const float l = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m) + l_o * __builtin_amdgcn_exp2f(m_o - m);
    const int64_t row = i0 + 32 * t + i32;
    if (h == 0 && row < m_rows) part[(int64_t)split * m_rows + row] = BCE ? make_float2(0.f, l_run[t] + l_o) : make_float2(m, l);
  }
}

#include "gcr_b3.h"  // (included inside the anonymous namespace: split-operand helpers shared with gcr_rank.hip)

// deferred rescale of the flash forwards: the reference point of a row's running sums only moves when a tile's
// maximum exceeds it by more than 2^kDefer
constexpr float kDefer = 8.0f;

// ------------------------------------------------------------------------------------------
// The two operand formats of the split-operand kernels (forward, and the pipelined two-product loop).
//   EngB3  three bf16 planes, six product terms (gcr_b3.h): any finite f32 operand, f32 accuracy.
//   EngH2  two f16 planes, THREE product terms: x = hi + lo with hi = f16(x), lo = f16(x - hi) (RNE,
//          v_cvt_pk_f16_f32) leaves <= 2^-22 |x|; hi*hi + hi*lo + lo*hi drops lo*lo (2^-22): half the matrix-core work of
//          EngB3 at an error of a few f32 roundings per product.  f16 has a 5-bit exponent, so the format is only used for
//          operands whose range is known — rows of at most unit norm (the caller's promise GCR_INFONCE_UNIT_ROWS; every
//          contrast loss of the reference normalises) and probabilities — each pre-scaled by a power of two that keeps its
//          largest value inside the f16 range and its typical values' residuals in the normal range:
//            stationary rows (<= inv_tau log2 e, inv_tau <= kH2MaxInvTau)   x 2^-2     (|.| <= 7.3)
//            streamed rows   (<= 1)                                          x 2^2
//            scores                                                          = the accumulator (product of the pre-scales 1)
//            P, MODE 1 (<= 2^kDefer = 2 relative to the lagging reference)   x 2^14
//            P, MODE 0 (w e^{s - lse} <= 2 max|w|)                           x 2^14 / 2^ceil(log2 max|w|) (max|w| from a
//                                                                            one-block pre-pass, h2_wscale_kernel)
//          Values below the f16 normal range (2^-14 after scaling: a row element under 2^-22, a probability 2^-28 under
//          the reference) go sub-normal, which v_cvt_pk_f16_f32 produces and v_mfma_f32_32x32x16_f16 honours
//          (scripts/exp/f16_denorm_probe.hip, run on the box): their absolute error stays <= 2^-25 of the scaled unit.
// ------------------------------------------------------------------------------------------
// operand-split form of EngH2 inside the two-product loop (EngH2::split): 0 = convert, two f16 -> f32 converts, two
// subtracts, convert (six instructions per pair); 1 = the residual out of one v_fma_mix_f32 per value (four); 2 = the lo
// plane straight out of v_fma_mixlo_f16 / v_fma_mixhi_f16 (three).  Measured on one box at 2048 x 1M x 64:
//   MODE 1 (flash forward)        form 0: 1.81 ms   form 1: 1.70   form 2: same as form 1 within noise
//   MODE 0 (table-side backward)  round 2 (175 vector instructions per tile): form 0: 1.78 ms, 1: 1.88, 2: 1.90;
//                                 round 3 (110 per tile after the fold pre-pass): form 0: 1.39 ms, 1: 1.32, 2: 1.33
//                                 (100K x 100K, 8 waves: 7.00 / 6.49 / 6.53; d = 32: 0.814 / 0.773 / 0.787)
// Form 2 never beats form 1 although it is one instruction shorter per pair: the f16-destination mixed forms issue at
// about twice the cost of a plain vector instruction.
template <int MODE>
constexpr int kSplitForm = 1;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
struct EngB3 {
  static constexpr int NPL = 3, NTERM = 6;
  static constexpr int kMinBlocks = 2;
  static constexpr float kSX = 1.0f, kSY = 1.0f;         // operand pre-scales
  static constexpr float kSInv = 1.0f;                   // accumulator -> log2-domain score
  static constexpr float kPExp = 0.0f;                   // log2 of the scale of P
  static constexpr float kDeferE = kDefer;
  static __host__ __device__ constexpr int ta(int t) { constexpr int v[6] = {2, 0, 1, 1, 0, 0}; return v[t]; }   // streamed-side
  static __host__ __device__ constexpr int tb(int t) { constexpr int v[6] = {0, 2, 1, 0, 1, 0}; return v[t]; }   // plane / other
  template <int MIX = 1>
  static __device__ __forceinline__ void split(float a, float b, unsigned (&p)[3]) { split3(a, b, p[0], p[1], p[2]); }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) { return mfma_bf16(a, b, c); }
};

struct EngH2 {
  static constexpr int NPL = 2, NTERM = 3;
  static constexpr int kMinBlocks = 2;          // 3 would cap the kernel at 168 VGPRs: 11-52 spilled dwords, 4-12 % slower
  // Operand pre-scales whose PRODUCT is 1 (round 4; rounds 2-3 used 2^4 and 2^8 with an accumulator scale of 2^-12): the
  // accumulator then IS the log2-domain score and the `acc * 2^-12 - m` fma in front of every exp2 disappears — one vector
  // instruction per probability less in loops whose time is their count of vector instructions.  Emulated in
  // numpy over 20 000 row pairs per case (unit rows, near-duplicates, rows with a few large and many tiny elements; d = 32 /
  // 64; 1/tau = 2 / 10 / 20): worst |score error| 0.91e-6 / 1.90e-6 / 1.04e-6 at 1/tau = 10, against 0.89e-6 / 1.84e-6 /
  // 1.01e-6 with the old pre-scales — what matters for the f16 residual planes is that typical elements stay above 2^-3
  // (stationary rows carry the 1/tau log2 e factor: ~0.45-0.9 after x 2^-2; streamed unit rows ~0.5 after x 2^2), below it
  // a residual goes sub-normal with ABSOLUTE error <= 2^-25, which the dot product tolerates (tests/test_h2_format_cpu.py).
  static constexpr float kSX = 0.25f, kSY = 4.0f;
  static constexpr float kSInv = 1.0f;
  static constexpr float kPExp = 14.0f;
  static constexpr float kDeferE = 1.0f;
  static __host__ __device__ constexpr int ta(int t) { constexpr int v[3] = {1, 0, 0}; return v[t]; }
  static __host__ __device__ constexpr int tb(int t) { constexpr int v[3] = {0, 1, 0}; return v[t]; }
  // The lo plane = f16(x - hi) straight out of ONE mixed-precision FMA per value: v_fma_mixlo_f16 / v_fma_mixhi_f16 read
  // the f16 half of `hi` in place, subtract it from the f32 value in f32 (exact: the difference of a value and its 11-bit
  // rounding has <= 13 significant bits) and round the result to f16 into the low / high half of the destination (RNE,
  // sub-normals kept, like v_cvt_pk_f16_f32): THREE instructions per pair of values, bit for bit the planes of the plain
  // form (convert, two f16 -> f32 converts, two subtracts, convert: six; `MIX = 0`, kept for reference and A/B).
  // Round 2's v_fma_mix_f32 form (residual in f32, then a second convert: four) is `MIX = 1`.
  template <int MIX = 1>
  static __device__ __forceinline__ void split(float a, float b, unsigned (&p)[2]) {
    const f16x2 hi = __builtin_convertvector((f32x2){a, b}, f16x2);                 // v_cvt_pk_f16_f32, RNE
    p[0] = __builtin_bit_cast(unsigned, hi);
    if (MIX == 2) {
      unsigned lo;
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(p[0]), "v"(a));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(p[0]), "v"(b));
      p[1] = lo;
      return;
    }
    float ra, rb;
    if (MIX == 1) {
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(p[0]), "v"(a));
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(p[0]), "v"(b));
    } else {
      ra = a - (float)hi[0];
      rb = b - (float)hi[1];
    }
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){ra, rb}, f16x2));
  }
  static __device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// stationary operand planes of engine E: frag[p][c] = 8 consecutive features [h*KH + 8c, +8) of plane p
template <class E, int D>
__device__ __forceinline__ void load_stationary_e(const float* __restrict__ a, const float* __restrict__ a_scale,
                                                  int64_t m_rows, int64_t row, int h, float mult,
                                                  u32x4 (&frag)[E::NPL][ShapeB3<D>::KC]) {
  using S = ShapeB3<D>;
  const bool valid = row < m_rows;
  const float s = valid ? (a_scale != nullptr ? a_scale[row] : 1.0f) * mult : 0.f;
  const float* p = a + (valid ? row : 0) * D + h * S::KH;
#pragma unroll
  for (int c = 0; c < S::KC; ++c) {
    const float4 v0 = valid ? *reinterpret_cast<const float4*>(p + 8 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 v1 = valid ? *reinterpret_cast<const float4*>(p + 8 * c + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned q[4][E::NPL];
    E::split(v0.x * s, v0.y * s, q[0]);
    E::split(v0.z * s, v0.w * s, q[1]);
    E::split(v1.x * s, v1.y * s, q[2]);
    E::split(v1.z * s, v1.w * s, q[3]);
#pragma unroll
    for (int pl = 0; pl < E::NPL; ++pl) frag[pl][c] = (u32x4){q[0][pl], q[1][pl], q[2][pl], q[3][pl]};
  }
}

// one staged float4 of the streamed tile -> NPL row-major planes in LDS
template <class E, int D>
__device__ __forceinline__ void stage_store_e_one(unsigned char* __restrict__ tile, int tid, const float4& v, int u) {
  using S = ShapeB3<D>;
  const int idx = tid + 256 * u;
  const int row = idx / (D / 4), c4 = idx % (D / 4);
  unsigned qa[E::NPL], qb[E::NPL];
  E::split(v.x, v.y, qa);
  E::split(v.z, v.w, qb);
  unsigned char* p = tile + row * S::ROWB + c4 * 8;
#pragma unroll
  for (int pl = 0; pl < E::NPL; ++pl) *reinterpret_cast<uint2*>(p + pl * S::PLANE) = make_uint2(qa[pl], qb[pl]);
}

template <class E, int D>
__device__ __forceinline__ void stage_store_e(unsigned char* __restrict__ tile, int tid,
                                              const float4 (&regs)[ShapeB3<D>::NLD]) {
#pragma unroll
  for (int u = 0; u < ShapeB3<D>::NLD; ++u) stage_store_e_one<E, D>(tile, tid, regs[u], u);
}

// S^T tile (streamed rows x stationary rows), NTERM products per k-chunk, smallest terms first
template <class E, int D, int NT>
__device__ __forceinline__ void score_tile_e(const unsigned char* __restrict__ tile, int i32, int h,
                                             const u32x4 (&bq)[NT][E::NPL][ShapeB3<D>::KC], f32x16 (&acc)[NT]) {
  using S = ShapeB3<D>;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const unsigned char* base = tile + i32 * S::ROWB + h * (S::KH * 2);
#pragma unroll
  for (int c = 0; c < S::KC; ++c) {
    u32x4 ap[E::NPL];
#pragma unroll
    for (int pl = 0; pl < E::NPL; ++pl) ap[pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE + 16 * c);
#pragma unroll
    for (int term = 0; term < E::NTERM; ++term)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = E::mfma(ap[E::ta(term)], bq[t][E::tb(term)][c], acc[t]);
  }
}


// staging with a workgroup of THREADS lanes (the 512-thread form of infonce_fwd_e_kernel: one float4 per lane at d = 64)
// one tile's share of a lane between its global load and its LDS store
template <int N>
struct StagedRows {
  float4 v[N];
  float s[N];
  bool live[N];
};

// The loads only: the row AND its scale go to registers untouched, the multiply happens where the tile is staged
// (stage_store_t_one), most of a tile loop iteration later.  (Multiplying here — and loading the scale inside an
// `if (b_scale)` — made every iteration open with `global_load ...; s_waitcnt vmcnt(0)`: a whole memory round trip in
// front of the first MFMA of every tile, for all waves of the workgroup at once.)  `b_scale == nullptr` loads a dummy
// word of `b` through a zero stride instead of branching (the value is replaced by 1 where it is used; both pointers are
// global, so the load stays a global_load — a generic-address load would tie vmcnt to lgkmcnt).

template <int D, int THREADS>
__device__ __forceinline__ void stage_load_t(const float* __restrict__ b, const float* __restrict__ b_scale, int64_t n_rows,
                                             int64_t j0, int tid, StagedRows<(kTileJ * D / 4) / THREADS>& g) {
  const float* __restrict__ sp = b_scale != nullptr ? b_scale + j0 : b;          // wave-uniform bases + small 32-bit lane offsets
  const float* __restrict__ tb = b + j0 * D;
  const unsigned stride = b_scale != nullptr ? 1u : 0u;
  const int rem = (int)min((int64_t)kTileJ, n_rows - j0);  // rows of this tile that exist (>= 1)
#pragma unroll
  for (int u = 0; u < (kTileJ * D / 4) / THREADS; ++u) {
    const int idx = tid + THREADS * u;
    const int row = idx / (D / 4), c4 = idx % (D / 4);
    const unsigned rr = (unsigned)min(row, rem - 1) & (kTileJ - 1);   // rows behind the end repeat the last one ...
    g.v[u] = *reinterpret_cast<const float4*>(tb + (rr * D + 4u * c4));
    g.s[u] = sp[rr * stride];
    g.live[u] = row < rem;                                   // ... with scale 0, applied with the multiply
  }
}

template <class E, int D, int THREADS>
__device__ __forceinline__ void stage_store_t_one(unsigned char* __restrict__ tile, int tid, const float4& v, float sc,
                                                  bool live, bool has_scale, int u) {
  using S = ShapeB3<D>;
  const int idx = tid + THREADS * u;
  const int row = idx / (D / 4), c4 = idx % (D / 4);
  unsigned qa[E::NPL], qb[E::NPL];
  sc = live ? (has_scale ? sc * E::kSY : E::kSY) : 0.f;
  E::split(v.x * sc, v.y * sc, qa);
  E::split(v.z * sc, v.w * sc, qb);
  unsigned char* p = tile + row * S::ROWB + c4 * 8;
#pragma unroll
  for (int pl = 0; pl < E::NPL; ++pl) *reinterpret_cast<uint2*>(p + pl * S::PLANE) = make_uint2(qa[pl], qb[pl]);
}

// NW = 8: a 512-thread workgroup of 512 anchors shares each staged table tile — half the staging (operand split + LDS
// stores) per MFMA of the 256-thread form; chosen for long splits at d = 64 (launch_fwd).
template <class E, int D, bool COLSUM, bool EXD = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void infonce_fwd_e_kernel(const float* __restrict__ a,
                                                                const float* __restrict__ a_scale, int64_t m_rows,
                                                                const float* __restrict__ b,
                                                                const float* __restrict__ b_scale, int64_t n_rows,
                                                                float scale2, int nsplit, int64_t tiles_per_split,
                                                                float2* __restrict__ part, float* __restrict__ col_sum,
                                                                float col_bound2) {
  using S = ShapeB3<D>;
  constexpr int NPL = E::NPL, NTERM = E::NTERM;
  constexpr int THREADS = 64 * NW, NLD = (kTileJ * D / 4) / THREADS;
  static_assert(NLD >= 1, "more lanes than float4s in a tile");
  __shared__ __align__(16) unsigned char lds[2][NPL * S::PLANE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t i0 = (mblk * NW + wave) * (32 * S::NT);

  u32x4 bq[S::NT][NPL][S::KC];
#pragma unroll
  for (int t = 0; t < S::NT; ++t)
    load_stationary_e<E, D>(a, a_scale, m_rows, i0 + 32 * t + i32, h, scale2 * E::kSX, bq[t]);

  // Two-plane format = rows of at most unit norm (the caller's promise): every log2-domain logit is <= scale2, so the
  // running sums take that bound as a FIXED reference point (terms >= 2^(-2 scale2) >= 2^-58 at 1/tau <= 20): no row
  // maximum, no rescale — 14 of ~100 vector instructions per 32 x 32 tile less
  constexpr bool FIXREF = E::NPL == 2;
  // ... and that fixed point is ZERO: a term is 2^s with |s| <= scale2 <= 29, the sum over up to 2^40 rows stays below 2^70
  // — no reference needed for range, none for precision (f32 is scale-invariant) — so with E::kSInv == 1 a probability is
  // exp2 of the accumulator register itself
  const float m_fix = 0.f;
  float m_run[S::NT], l_run[S::NT], a_valid[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    m_run[t] = FIXREF ? m_fix : kNegBig;
    l_run[t] = 0.f;
    a_valid[t] = (i0 + 32 * t + i32 < m_rows) ? 1.0f : 0.f;
  }

  const int64_t total_tiles = (n_rows + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  StagedRows<NLD> ga, gb;                                  // two tiles in flight: loaded in one step, staged in the next
  auto colsum = [&](const f32x16 (&acc)[S::NT], int64_t tt) {
    const int64_t j0 = tt * kTileJ;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = 0.f;
#pragma unroll
      for (int t = 0; t < S::NT; ++t) e += a_valid[t] * __builtin_amdgcn_exp2f(acc[t][r] - col_bound2);   // acc already scaled
      e = half_wave_sum_to_last_lane(e);
      const int64_t j = j0 + acc_row(r, h);
      if (i32 == 31 && j < n_rows) atomicAdd(col_sum + j, e);
    }
  };
  // the table's last tile is the only one that can be ragged
  auto epilogue_any = [&](f32x16 (&acc)[S::NT], int64_t tt) {
    if (E::kSInv != 1.0f) {                               // accumulator -> log2-domain scores
#pragma unroll
      for (int t = 0; t < S::NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= E::kSInv;
    }
    if (EXD) {
      int xr[S::NT];
#pragma unroll
      for (int t = 0; t < S::NT; ++t) xr[t] = diag_offset(i0 + 32 * t + i32, tt * kTileJ, h);
      lse_update<S::NT, true, true>(acc, rows_left(n_rows, tt * kTileJ, h), m_run, l_run, xr);
    } else if ((tt + 1) * kTileJ > n_rows)
      lse_update<S::NT, true>(acc, rows_left(n_rows, tt * kTileJ, h), m_run, l_run);
    else
      lse_update<S::NT, false>(acc, 64, m_run, l_run);
    if (COLSUM) colsum(acc, tt);
  };
  if (tile0 >= tile1) {
    // empty split (plan_fwd never makes one): fall through to the (kNegBig, 0) partial
  } else {
    // Software pipeline, interleaved by hand: the MFMAs of tile tt+1 alternate in program order with
    // the softmax VALU work of tile tt (branch-free: every tile before the split's last one is full)
    // and with the operand split of tile tt+2, one share of VALU "units" per MFMA slot, pinned by
    // sched_barrier.  A wave issues in order, so this is what lets ONE wave keep the matrix pipe busy
    // under its own exp2 / max / add stream (two waves of a SIMD otherwise fall into lock-step and
    // serialise: measured 48 % MFMA utilisation before, scripts/perf_infonce_engine_ab.py).
    constexpr int NS = NTERM * S::KC * S::NT;                     // MFMA slots per step
    constexpr int NU = 22 * S::NT + NLD + (COLSUM ? 16 : 0);      // VALU micro-units per step
    const int64_t last = tile1 - 1;
    f32x16 acc_a[S::NT], acc_b[S::NT];
    auto store_all = [&](unsigned char* tile, const StagedRows<NLD>& g) {
#pragma unroll
      for (int u = 0; u < NLD; ++u) stage_store_t_one<E, D, THREADS>(tile, tid, g.v[u], g.s[u], g.live[u], b_scale != nullptr, u);
    };
    stage_load_t<D, THREADS>(b, b_scale, n_rows, tile0 * kTileJ, tid, ga);
    stage_load_t<D, THREADS>(b, b_scale, n_rows, min(tile0 + 1, last) * kTileJ, tid, gb);
    store_all(lds[0], ga);
    stage_load_t<D, THREADS>(b, b_scale, n_rows, min(tile0 + 2, last) * kTileJ, tid, ga);
    __syncthreads();
    score_tile_e<E, D, S::NT>(lds[0], i32, h, bq, acc_a);
    store_all(lds[1], gb);
    __syncthreads();
    // step tt: scores of tile tt + 1 (in lds[nb]) || softmax of tile tt || staging of tile tt + 2 (loaded one step
    // earlier, in `st`) into the other buffer || the loads of tile tt + 3 into `ld`: a load has a whole step to land — it
    // used to be issued at the top of the step that staged it, with its scale multiplied in on the spot, i.e. behind a
    // `s_waitcnt vmcnt(0)` that every wave of the workgroup sat out in front of the step's first MFMA
    auto step = [&](f32x16 (&cur)[S::NT], f32x16 (&nxt)[S::NT], int64_t tt, int nb, StagedRows<NLD>& ld,
                    const StagedRows<NLD>& st) {
      stage_load_t<D, THREADS>(b, b_scale, n_rows, min(tt + 3, last) * kTileJ, tid, ld);
      const unsigned char* base = lds[nb] + i32 * S::ROWB + h * (S::KH * 2);
      unsigned char* out = lds[nb ^ 1];
      float tmax[S::NT], m_new[S::NT], sum[S::NT];
      int xr[S::NT];
      if (EXD) {
#pragma unroll
        for (int t = 0; t < S::NT; ++t) xr[t] = diag_offset(i0 + 32 * t + i32, tt * kTileJ, h);
      }
      // micro-unit m of the step's VALU work: m < 22*NT -> softmax unit m / NT of anchor tile m % NT
      // (0-3 partial max, 4 new max, 5-20 exp2-add of one register, 21 fold into the running sum);
      // then the operand split + LDS store of one staged float4; then (COLSUM) one register's column sums
      auto micro = [&](int m) {
        const int u = m / S::NT, t = m % S::NT;
        if (m >= 22 * S::NT + NLD) {
          if (COLSUM) {
            const int r = m - 22 * S::NT - NLD;
            float e = 0.f;
#pragma unroll
            for (int q = 0; q < S::NT; ++q) e += a_valid[q] * __builtin_amdgcn_exp2f(fmaf(cur[q][r], E::kSInv, -col_bound2));
            e = half_wave_sum_to_last_lane(e);
            const int64_t j = tt * kTileJ + acc_row(r, h);
            if (i32 == 31 && j < n_rows) atomicAdd(col_sum + j, e);
          }
        } else if (m >= 22 * S::NT) {
          stage_store_t_one<E, D, THREADS>(out, tid, st.v[m - 22 * S::NT], st.s[m - 22 * S::NT], st.live[m - 22 * S::NT],
                                           b_scale != nullptr, m - 22 * S::NT);
        } else if (u < 4) {
          if (EXD) {                                    // excluded diagonal pair: -inf before it is seen by max / exp2
#pragma unroll
            for (int e = 0; e < 4; ++e) cur[t][4 * u + e] = (xr[t] == e + 8 * u) ? -INFINITY : cur[t][4 * u + e];
          }
          if (!FIXREF) {
            // chained so that each pair of scores costs one v_max3_f32
            const float x0 = u == 0 ? cur[t][0] : tmax[t];
            tmax[t] = fmaxf(fmaxf(fmaxf(fmaxf(x0, cur[t][4 * u]), cur[t][4 * u + 1]), cur[t][4 * u + 2]), cur[t][4 * u + 3]);
          }
        } else if (u == 4) {
          m_new[t] = FIXREF ? m_fix : fmaxf(m_run[t], tmax[t] * E::kSInv);
          sum[t] = 0.f;
        } else if (u < 21) {
          if (FIXREF && E::kSInv == 1.0f) sum[t] += __builtin_amdgcn_exp2f(cur[t][u - 5]);
          else sum[t] += __builtin_amdgcn_exp2f(fmaf(cur[t][u - 5], E::kSInv, -m_new[t]));
        } else if (FIXREF) {
          l_run[t] += sum[t];
        } else {
          l_run[t] = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m_new[t]) + sum[t];
          m_run[t] = m_new[t];
        }
      };
      u32x4 ap[2][NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) ap[0][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE);
#pragma unroll
      for (int c = 0; c < S::KC; ++c) {
        if (c + 1 < S::KC) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            ap[(c + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE + 16 * (c + 1));
        }
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
#pragma unroll
          for (int t = 0; t < S::NT; ++t) {
            const int slot = (c * NTERM + term) * S::NT + t;
            f32x16 cin = nxt[t];
            if (c == 0 && term == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) cin[r] = 0.f;
            }
            nxt[t] = E::mfma(ap[c & 1][E::ta(term)], bq[t][E::tb(term)][c], cin);
#pragma unroll
            for (int u = slot * NU / NS; u < (slot + 1) * NU / NS; ++u) micro(u);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      // pin the running statistics here: without it the compiler sinks this step's softmax work past
      // the barrier into the next step, in front of its MFMAs
#pragma unroll
      for (int t = 0; t < S::NT; ++t) asm volatile("" : "+v"(m_run[t]), "+v"(l_run[t]));
      __syncthreads();
    };
    int64_t tt = tile0;
    for (; tt + 2 <= last; tt += 2) {
      step(acc_a, acc_b, tt, 1, gb, ga);
      step(acc_b, acc_a, tt + 1, 0, ga, gb);
    }
    if (tt < last) {
      step(acc_a, acc_b, tt, 1, gb, ga);
      epilogue_any(acc_b, last);
    } else {
      epilogue_any(acc_a, last);
    }
  }

#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const float m_o = __shfl_xor(m_run[t], 32, 64), l_o = __shfl_xor(l_run[t], 32, 64);
    const float m = fmaxf(m_run[t], m_o);
    const float l = l_run[t] * __builtin_amdgcn_exp2f(m_run[t] - m) + l_o * __builtin_amdgcn_exp2f(m_o - m);
    const int64_t row = i0 + 32 * t + i32;
    if (h == 0 && row < m_rows) part[(int64_t)split * m_rows + row] = make_float2(m, l);
  }
}

// Engine selection.  Default: the split-operand bf16 engine for d <= 128, where forward AND backward have
// it; d = 256 stays on the f32 MFMA (its three operand planes would not fit the registers).  The flag
// GCR_INFONCE_ENGINE_F32 of the _ex entry points forces the f32 MFMA.  The choice is an ARGUMENT, never
// read from the environment: the host wrapper resolves it once per forward and hands the same flag to
// the backward (which must recompute the forward's logits with the forward's engine, or sum_j P_ij = 1
// only holds to ~1e-6 and near-cancelling gradients lose digits).
bool use_b3(int d, bool force_f32 = false) { return d <= 128 && !force_f32; }
// the two-plane f16 format of the pipelined two-product loop (EngH2 below): rows of at most unit norm (the caller's
// promise), d <= 64, 1/tau within the pre-scale's head-room
// (1/tau <= 20 covers every call site of the reference — tau in 0.07 .. 0.5 — and keeps the format's logit error,
// 3 * 2^-22 * inv_tau * log2 e in the worst case, an order below the 1e-5 of the parity tests; at 1/tau = 40 .. 60 the
// two formats differ by up to 1.3e-5 in the gradients, scripts/stress_infonce_formats.py)
constexpr float kH2MaxInvTau = 20.0f;
bool use_h2(int d, float inv_tau, bool unit_rows, bool force_f32) {
  return use_b3(d, force_f32) && unit_rows && d <= 64 && inv_tau > 0.f && inv_tau <= kH2MaxInvTau;
}
// ... and the two-product loop (flash forward, backward) also at d = 128: two planes of 128 features fit the pipelined
// loop's registers (12-99 spilled dwords at 256), three do not (infonce_fwdo_b3_kernel / infonce_bwd_b3_kernel)
bool use_h2_loop(int d, float inv_tau, bool unit_rows, bool force_f32) {
  return use_b3(d, force_f32) && unit_rows && d <= 128 && inv_tau > 0.f && inv_tau <= kH2MaxInvTau;
}
constexpr int64_t kBwdHeader = 256;   // d <= 64: the backward's workspace starts with the two floats of h2_wscale_kernel
// ... followed by h2_fold_kernel's image of the streamed rows (two per-row arrays and the scaled rows themselves, padded
// to whole tiles), rounded to 256 B; the per-split partial gradients start behind them
constexpr int64_t fold_rows(int64_t ny) { return (ny + kTileJ - 1) / kTileJ * kTileJ; }
constexpr int64_t bwd_header_bytes(int64_t ny, int d) {
  return d <= 128 ? kBwdHeader + ((fold_rows(ny) * (2 + d) * (int64_t)sizeof(float) + 255) / 256) * 256 : 0;
}

// natural-log LSE of the scaled logits from the per-split (max2, sum2) partials
__global__ void infonce_merge_kernel(const float2* __restrict__ part, int nsplit, int64_t m_rows,
                                     float* __restrict__ lse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m_rows) return;
  float m = kNegBig;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part[(int64_t)s * m_rows + i].x);
  float l = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float2 p = part[(int64_t)s * m_rows + i];
    l += p.y * __builtin_amdgcn_exp2f(p.x - m);
  }
  lse[i] = (m + __log2f(l)) * kLn2;
}

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// out[r] = 1 / max(||x_r||_2, eps)   (F.normalize's denominator, ncl.py:127, gcl.py:29-30)
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, int64_t n, int d, float eps,
                                                           float* __restrict__ out) {
  const int l16 = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += (int64_t)gridDim.x * 16) {
    float ss = 0.f;
    if ((d & 3) == 0) {
      for (int c = l16 * 4; c < d; c += 64) {
        const float4 v = *reinterpret_cast<const float4*>(x + r * d + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
    } else {
      for (int c = l16; c < d; c += 16) ss += x[r * d + c] * x[r * d + c];
    }
    ss = group16_sum(ss);
    if (l16 == 0) out[r] = 1.0f / fmaxf(sqrtf(ss), eps);
  }
}

// out[i] = scale * a_scale[i] * b_scale[p] * <a_i, b_p>,  p = pos[i] (or i): the positive logit
__global__ __launch_bounds__(256) void pos_logit_kernel(const float* __restrict__ a, const float* __restrict__ a_scale,
                                                        const float* __restrict__ b, const float* __restrict__ b_scale,
                                                        const int64_t* __restrict__ pos, int64_t m, int64_t n, int d,
                                                        float scale, float* __restrict__ out) {
  const int l16 = threadIdx.x & 15;
  for (int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < m; i += (int64_t)gridDim.x * 16) {
    const int64_t p = pos != nullptr ? pos[i] : i;
    float dot = 0.f;
    const bool ok = p >= 0 && p < n;
    if (ok) {
      if ((d & 3) == 0) {
        for (int c = l16 * 4; c < d; c += 64) {
          const float4 u = *reinterpret_cast<const float4*>(a + i * d + c);
          const float4 v = *reinterpret_cast<const float4*>(b + p * d + c);
          dot += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
        }
      } else {
        for (int c = l16; c < d; c += 16) dot += a[i * d + c] * b[p * d + c];
      }
    }
    dot = group16_sum(dot);
    if (l16 == 0) {
      const float sa = a_scale != nullptr ? a_scale[i] : 1.0f;
      const float sb = (ok && b_scale != nullptr) ? b_scale[p] : 1.0f;
      out[i] = ok ? scale * sa * sb * dot : __builtin_nanf("");
    }
  }
}


struct FwdPlan {
  int nsplit;
  int64_t tiles_per_split;
  int64_t m_blocks;
};

// Column split.  `resident` = blocks the chip holds at once (256 CUs x blocks/CU of the kernel).
//  * few row blocks: ONE wave of blocks, nsplit = floor(resident / m_blocks): a grid slightly larger
//    than the resident set (784 vs 768) costs a whole extra round (2.15 -> 3.0 ms measured);
//  * many row blocks: several rounds are unavoidable, so make them short: <= 320 tiles per block and
//    >= 3 rounds (100K x 100K: 95 TF at 782 blocks, 119 TF at 3128; scripts/perf_infonce_ab.py);
//  * never fewer than 16 tiles per block (anchor prologue + partial merge dominate below that).
FwdPlan plan_fwd(int64_t m, int64_t n, int anchors_per_block, int64_t resident) {
  FwdPlan p;
  p.m_blocks = (m + anchors_per_block - 1) / anchors_per_block;
  const int64_t total_tiles = (n + kTileJ - 1) / kTileJ;
  int64_t nsplit;
  if (p.m_blocks * 2 <= resident) {
    nsplit = resident / p.m_blocks;
  } else {
    nsplit = (total_tiles + 319) / 320;
    if (p.m_blocks * nsplit < 3 * resident) nsplit = (3 * resident + p.m_blocks - 1) / p.m_blocks;
  }
  const int64_t max_split = total_tiles / 16 > 0 ? total_tiles / 16 : 1;
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit > total_tiles) nsplit = total_tiles;
  if (nsplit < 1) nsplit = 1;
  p.tiles_per_split = (total_tiles + nsplit - 1) / nsplit;
  p.nsplit = (int)((total_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
  if (p.nsplit < 1) p.nsplit = 1;
  return p;
}

// ------------------------------------------------------------------------------------------
// Backward (flash-style recompute).  For stationary rows x_i and streamed rows y_j:
//   P_ij = w_x[i] * exp(s_ij - lse_x[i]) + w_y[j] * exp(s_ij - lse_y[j]),  s = logits (1/tau scaled)
//   G[i, :] = inv_tau * sum_j P_ij * yhat_j                  (gradient w.r.t. the scaled row xhat_i)
// The score tile is recomputed exactly as in the forward; because its accumulator already has the
// stationary row on the lane and the streamed rows in the registers, register r IS the MFMA
// B-operand of k-step r of the second product G^T[c][i] += yhat[j_r][c] * P[j_r][i] (no LDS round
// trip for P); the A operand yhat[j_r(h)][c] is a conflict-free ds_read_b32 from the same tile.
// w_x/lse_x are the "lane side" statistics (row-softmax of the stationary rows), w_y/lse_y the
// "register side" ones (softmax over the stationary index for every streamed row): calling the
// kernel twice with the roles swapped yields both input gradients of the symmetric loss.
// ------------------------------------------------------------------------------------------
template <int D>
struct BwdShape {
  static constexpr int NT = D <= 64 ? 2 : 1;
  static constexpr int CT = D / 32 > 4 ? 4 : D / 32;     // column tiles of 32 handled per launch
  static constexpr int PASSES = (D / 32) / CT;
  static constexpr int ROWS_PER_BLOCK = 4 * 32 * NT;
};

template <int D, bool EXD = false, bool BCE = false>
__global__ __launch_bounds__(256, 2) void infonce_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ x_scale, int64_t mx, const float* __restrict__ y,
    const float* __restrict__ y_scale, int64_t ny, float scale2, float out_scale, const float* __restrict__ lse_x,
    const float* __restrict__ w_x, const float* __restrict__ lse_y, const float* __restrict__ w_y, int ct0, int nsplit,
    int64_t tiles_per_split, float* __restrict__ gpart) {
  using S = Shape<D>;
  using B = BwdShape<D>;
  __shared__ __align__(16) float lds[2][kTileJ * S::STRIDE];
  __shared__ __align__(16) float st_lse[2][kTileJ];
  __shared__ __align__(16) float st_w[2][kTileJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t i0 = (mblk * 4 + wave) * (32 * B::NT);

  float bfrag[B::NT][S::KH];
  float lse2l[B::NT], wl[B::NT];
#pragma unroll
  for (int t = 0; t < B::NT; ++t) {
    const int64_t row = i0 + 32 * t + i32;
    load_stationary<D>(x, x_scale, mx, row, h, scale2, bfrag[t]);
    const bool on = row < mx && w_x != nullptr;
    wl[t] = on ? w_x[row] : 0.f;
    lse2l[t] = (on && !BCE) ? lse_x[row] * kLog2e : 1.0e30f;  // disabled term: exp2(-huge) = 0, never 0 * inf
  }
  f32x16 gacc[B::NT][B::CT];
#pragma unroll
  for (int t = 0; t < B::NT; ++t)
#pragma unroll
    for (int c = 0; c < B::CT; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[t][c][r] = 0.f;

  const int64_t total_tiles = (ny + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  float4 regs[S::NLD];
  float s_lse = 0.f, s_w = 0.f;
  auto load_stats = [&](int64_t j0) {
    if (tid < kTileJ) {
      const int64_t j = j0 + tid;
      const bool on = j < ny && w_y != nullptr;
      s_w = on ? w_y[j] : 0.f;
      s_lse = (on && !BCE) ? lse_y[j] * kLog2e : 1.0e30f;
    }
  };
  auto store_stats = [&](int buf) {
    if (tid < kTileJ) {
      st_lse[buf][tid] = s_lse;
      st_w[buf][tid] = s_w;
    }
  };
  if (tile0 < tile1) {
    stage_load<D>(y, y_scale, ny, tile0 * kTileJ, tid, regs);
    load_stats(tile0 * kTileJ);
    stage_store<D>(lds[0], tid, regs);
    store_stats(0);
  }
  __syncthreads();
  for (int64_t tt = tile0; tt < tile1; ++tt) {
    const int cur = (int)((tt - tile0) & 1);
    const bool more = tt + 1 < tile1;
    if (more) {
      stage_load<D>(y, y_scale, ny, (tt + 1) * kTileJ, tid, regs);
      load_stats((tt + 1) * kTileJ);
    }
    f32x16 acc[B::NT];
    {
#pragma unroll
      for (int t = 0; t < B::NT; ++t)
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
          for (int t = 0; t < B::NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae[e], bfrag[t][4 * q + e], acc[t], 0, 0, 0);
      }
    }
    const int64_t j0 = tt * kTileJ;
    const bool ragged = j0 + kTileJ > ny;
    // P in place of the scores
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 lr = *reinterpret_cast<const float4*>(&st_lse[cur][8 * g + 4 * h]);
      const float4 wr = *reinterpret_cast<const float4*>(&st_w[cur][8 * g + 4 * h]);
      const float lre[4] = {lr.x, lr.y, lr.z, lr.w};
      const float wre[4] = {wr.x, wr.y, wr.z, wr.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const bool dead_j = ragged && (j0 + acc_row(r, h) >= ny);
#pragma unroll
        for (int t = 0; t < B::NT; ++t) {
          // EXD: the diagonal pair (stationary row i, streamed row j = i) carries no probability
          const bool dead = dead_j || (EXD && diag_offset(i0 + 32 * t + i32, j0, h) == e + 8 * g);
          const float sc = dead ? -INFINITY : acc[t][r];
          if (BCE) {                                // P = (w_x[i] + w_y[j]) sigmoid(s_ij)
            acc[t][r] = (wl[t] + wre[e]) * bce_sigmoid(sc);
          } else
            acc[t][r] = wl[t] * __builtin_amdgcn_exp2f(sc - lse2l[t]) + wre[e] * __builtin_amdgcn_exp2f(sc - lre[e]);
        }
      }
    }
    // G^T[c][i] += yhat[j_r][c] * P[j_r][i]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* yrow = lds[cur] + acc_row(r, h) * S::STRIDE + 32 * ct0 + i32;
#pragma unroll
      for (int c = 0; c < B::CT; ++c) {
        const float yv = yrow[32 * c];
#pragma unroll
        for (int t = 0; t < B::NT; ++t)
          gacc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, acc[t][r], gacc[t][c], 0, 0, 0);
      }
    }
    if (more) {
      stage_store<D>(lds[cur ^ 1], tid, regs);
      store_stats(cur ^ 1);
    }
    __syncthreads();
  }

  // gacc[t][c] register rr of lane (i, h) is column 32*(ct0+c) + acc_row(rr, h) of row i
  float* gout = gpart + (int64_t)split * mx * D;
#pragma unroll
  for (int t = 0; t < B::NT; ++t) {
    const int64_t row = i0 + 32 * t + i32;
    if (row < mx) {
#pragma unroll
      for (int c = 0; c < B::CT; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(gacc[t][c][4 * g + 0] * out_scale, gacc[t][c][4 * g + 1] * out_scale,
                                 gacc[t][c][4 * g + 2] * out_scale, gacc[t][c][4 * g + 3] * out_scale);
          *reinterpret_cast<float4*>(gout + row * D + 32 * (ct0 + c) + 8 * g + 4 * h) = v;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Backward on the split-operand engine (d <= 64).  Same structure as infonce_bwd_kernel; the score
// tile is recomputed with exactly the forward's sequence of bf16 MFMAs (bitwise the forward's
// logits when the roles are the forward's, so sum_j P_ij = 1 to rounding), P is split into three
// bf16 planes in registers (an accumulator register pair -> one packed dword of each plane: the
// accumulator layout is already the B-operand layout of the 32x32x16 MFMA, k = the 8 + 8 streamed
// rows a lane holds per 16-row chunk), and the second product reads its A operand
// yhat[rows][feature] from a TRANSPOSED copy of the three planes in LDS ([feature][row] bf16,
// 72-B rows: conflict-free ds_read_b64), staged together with the row-major planes.
// ------------------------------------------------------------------------------------------
template <int D>
struct BwdB3 {
  static constexpr int CT = D / 32;                    // feature tiles of 32, all in one launch
  static constexpr int RT = 72;                        // bytes per feature of a transposed plane
  static constexpr int TPLANE = D * RT;
  static constexpr int ROWS_PER_BLOCK = 128;           // one tile of 32 stationary rows per wave
  static constexpr int TILE_BYTES = 3 * ShapeB3<D>::PLANE + 3 * TPLANE;
};

template <int D>
__device__ __forceinline__ void stage_store_b3t_one(unsigned char* __restrict__ tile, int tid, const float4& v, int u) {
  using S = ShapeB3<D>;
  using B = BwdB3<D>;
  const int idx = tid + 256 * u;
  const int row = idx / (D / 4), c4 = idx % (D / 4);
  unsigned a1, a2, a3, b1, b2, b3;
  split3(v.x, v.y, a1, a2, a3);
  split3(v.z, v.w, b1, b2, b3);
  unsigned char* p = tile + row * S::ROWB + c4 * 8;
  *reinterpret_cast<uint2*>(p) = make_uint2(a1, b1);
  *reinterpret_cast<uint2*>(p + S::PLANE) = make_uint2(a2, b2);
  *reinterpret_cast<uint2*>(p + 2 * S::PLANE) = make_uint2(a3, b3);
  unsigned short* q = reinterpret_cast<unsigned short*>(tile + 3 * S::PLANE + (4 * c4) * B::RT + row * 2);
  const unsigned pa[3] = {a1, a2, a3}, pb[3] = {b1, b2, b3};
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    unsigned short* qq = q + pl * (B::TPLANE / 2);
    qq[0] = (unsigned short)(pa[pl] & 0xffffu);
    qq[B::RT / 2] = (unsigned short)(pa[pl] >> 16);
    qq[2 * (B::RT / 2)] = (unsigned short)(pb[pl] & 0xffffu);
    qq[3 * (B::RT / 2)] = (unsigned short)(pb[pl] >> 16);
  }
}

// SIDES: which softmax terms of P are live — 0 both, 1 only the stationary rows' (w_x, lse_x), 2 only the
// streamed rows' (w_y, lse_y): a dead term costs an exp2 and an FMA per score in the exposed part of the loop.
template <int D, bool EXD = false, int SIDES = 0>
__global__ __launch_bounds__(256, 2) void infonce_bwd_b3_kernel(
    const float* __restrict__ x, const float* __restrict__ x_scale, int64_t mx, const float* __restrict__ y,
    const float* __restrict__ y_scale, int64_t ny, float scale2, float out_scale, const float* __restrict__ lse_x,
    const float* __restrict__ w_x, const float* __restrict__ lse_y, const float* __restrict__ w_y, int nsplit,
    int64_t tiles_per_split, float* __restrict__ gpart) {
  using S = ShapeB3<D>;
  using B = BwdB3<D>;
  // d = 128: one tile is 54 KB (row-major + transposed planes): a single LDS buffer (two barriers per tile,
  // two blocks per CU cover each other) instead of the double buffer of d <= 64
  constexpr int NBUF = D <= 64 ? 2 : 1;
  __shared__ __align__(16) unsigned char lds[NBUF][B::TILE_BYTES];
  __shared__ __align__(16) float st_lse[2][kTileJ];
  __shared__ __align__(16) float st_w[2][kTileJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t row_i = (mblk * 4 + wave) * 32 + i32;

  u32x4 bq[1][3][S::KC];
  load_stationary_b3<D>(x, x_scale, mx, row_i, h, scale2, bq[0]);
  const bool on_x = row_i < mx && w_x != nullptr;
  const float wl = on_x ? w_x[row_i] : 0.f;
  const float lse2l = on_x ? lse_x[row_i] * kLog2e : 1.0e30f;    // disabled term: exp2(-huge) = 0, never 0 * inf
  f32x16 gacc[B::CT];
#pragma unroll
  for (int c = 0; c < B::CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[c][r] = 0.f;

  const int64_t total_tiles = (ny + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  float4 regs[S::NLD];
  float s_lse = 0.f, s_w = 0.f;
  auto load_stats = [&](int64_t j0) {
    if (tid < kTileJ) {
      const int64_t j = j0 + tid;
      const bool on = j < ny && w_y != nullptr;
      s_w = on ? w_y[j] : 0.f;
      s_lse = on ? lse_y[j] * kLog2e : 1.0e30f;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) stage_store_b3t_one<D>(lds[buf % NBUF], tid, regs[u], u);
    if (tid < kTileJ) {
      st_lse[buf][tid] = s_lse;
      st_w[buf][tid] = s_w;
    }
  };
  {
    if (tile0 < tile1) {
      stage_load<D>(y, y_scale, ny, tile0 * kTileJ, tid, regs);
      load_stats(tile0 * kTileJ);
      store_tile(0);
    }
    __syncthreads();
    for (int64_t tt = tile0; tt < tile1; ++tt) {
      const int cur = (int)((tt - tile0) & 1);
      const bool more = tt + 1 < tile1;
      if (more) {
        stage_load<D>(y, y_scale, ny, (tt + 1) * kTileJ, tid, regs);
        load_stats((tt + 1) * kTileJ);
      }
      f32x16 acc[1];
      score_tile_b3<D, 1>(lds[cur % NBUF], i32, h, bq, acc);
      const int64_t j0 = tt * kTileJ;
      const bool ragged = j0 + kTileJ > ny;
      const int xr = EXD ? diag_offset(row_i, j0, h) : -1;
      // P in place of the scores
  #pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 lr = *reinterpret_cast<const float4*>(&st_lse[cur][8 * g + 4 * h]);
        const float4 wr = *reinterpret_cast<const float4*>(&st_w[cur][8 * g + 4 * h]);
        const float lre[4] = {lr.x, lr.y, lr.z, lr.w};
        const float wre[4] = {wr.x, wr.y, wr.z, wr.w};
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const bool dead = (ragged && (j0 + acc_row(r, h) >= ny)) || (EXD && xr == e + 8 * g);
          const float sc = dead ? -INFINITY : acc[0][r];
          if (SIDES == 1) acc[0][r] = wl * __builtin_amdgcn_exp2f(sc - lse2l);
          else if (SIDES == 2) acc[0][r] = wre[e] * __builtin_amdgcn_exp2f(sc - lre[e]);
          else acc[0][r] = wl * __builtin_amdgcn_exp2f(sc - lse2l) + wre[e] * __builtin_amdgcn_exp2f(sc - lre[e]);
        }
      }
      // G^T[c][i] += yhat[j][c] * P[j][i], 16 streamed rows per k-chunk, six bf16 terms
      const unsigned char* tbase = lds[cur % NBUF] + 3 * S::PLANE + i32 * B::RT + 8 * h;
  #pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        u32x4 pp[3];
        {
          unsigned q[3][4];
  #pragma unroll
          for (int e = 0; e < 4; ++e) split3(acc[0][8 * kc + 2 * e], acc[0][8 * kc + 2 * e + 1], q[0][e], q[1][e], q[2][e]);
  #pragma unroll
          for (int pl = 0; pl < 3; ++pl) pp[pl] = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
        }
  #pragma unroll
        for (int c = 0; c < B::CT; ++c) {
          u32x4 ya[3];
  #pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const unsigned char* p = tbase + pl * B::TPLANE + (32 * c) * B::RT + 32 * kc;
            const uint2 lo = *reinterpret_cast<const uint2*>(p);
            const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
            ya[pl] = (u32x4){lo.x, lo.y, hi.x, hi.y};
          }
          gacc[c] = mfma_bf16(ya[2], pp[0], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[2], gacc[c]);
          gacc[c] = mfma_bf16(ya[1], pp[1], gacc[c]);
          gacc[c] = mfma_bf16(ya[1], pp[0], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[1], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[0], gacc[c]);
        }
      }
      if (NBUF == 1) __syncthreads();          // every wave is done reading the only buffer
      if (more) store_tile(cur ^ 1);
      __syncthreads();
    }
  }

  // gacc[c] register rr of lane (i, h) is column 32*c + acc_row(rr, h) of row i
  float* gout = gpart + (int64_t)split * mx * D;
  if (row_i < mx) {
#pragma unroll
    for (int c = 0; c < B::CT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(gacc[c][4 * g + 0] * out_scale, gacc[c][4 * g + 1] * out_scale,
                               gacc[c][4 * g + 2] * out_scale, gacc[c][4 * g + 3] * out_scale);
        *reinterpret_cast<float4*>(gout + row_i * D + 32 * c + 8 * g + 4 * h) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// Forward WITH the softmax-weighted row sum ("fwd_o", flash-attention forward):
//   lse[i] = log sum_j exp(s_ij),   o[i, :] = sum_j softmax_j(s_i.)_j * yhat_j
// in one pass over the streamed rows (online max, deferred rescale).  For a row-softmax loss the
// gradient w.r.t. the stationary side is then  dL/dxhat_i = dL/dlse_i * inv_tau * o[i, :]  — no
// backward pass over the M x N tile for that side; the backward recomputes the score tile ONCE
// (streamed-side gradient) instead of twice.  Same tile engine as infonce_bwd_b3_kernel: score
// MFMAs in the forward's order (bitwise the forward's logits), P split in registers, second product
// against the transposed planes.  The two lane halves of a stationary row hold different streamed
// rows of the same k-chunk, so they share one running maximum (one cross-half shuffle per tile).
// Rescaling the accumulators is deferred while no logit exceeds the reference point by more than
// 2^kDefer (wave-uniform branch): after the first tiles it almost never runs.
// ------------------------------------------------------------------------------------------

template <int D, bool EXD = false>
__global__ __launch_bounds__(256, 2) void infonce_fwdo_b3_kernel(
    const float* __restrict__ x, const float* __restrict__ x_scale, int64_t mx, const float* __restrict__ y,
    const float* __restrict__ y_scale, int64_t ny, float scale2, int nsplit, int64_t tiles_per_split,
    float2* __restrict__ part, float* __restrict__ opart) {
  using S = ShapeB3<D>;
  using B = BwdB3<D>;
  constexpr int NBUF = D <= 64 ? 2 : 1;
  __shared__ __align__(16) unsigned char lds[NBUF][B::TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t row_i = (mblk * 4 + wave) * 32 + i32;

  u32x4 bq[1][3][S::KC];
  load_stationary_b3<D>(x, x_scale, mx, row_i, h, scale2, bq[0]);
  float m_run = kNegBig, l_run = 0.f;
  f32x16 gacc[B::CT];
#pragma unroll
  for (int c = 0; c < B::CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[c][r] = 0.f;

  const int64_t total_tiles = (ny + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);

  // scores (log2 domain) -> un-normalised probabilities relative to the running reference point
  auto softmax_p = [&](f32x16& acc, int64_t tt) {
    const int64_t j0 = tt * kTileJ;
    if (EXD || j0 + kTileJ > ny) {                       // wave-uniform: ragged last tile / excluded diagonal
      const int xr = EXD ? diag_offset(row_i, j0, h) : -1;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const bool dead = (j0 + acc_row(r, h) >= ny) || (EXD && xr == e + 8 * g);
          acc[r] = dead ? -INFINITY : acc[r];
        }
    }
    float tmax = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
#pragma unroll
    for (int g = 1; g < 4; ++g)
      tmax = fmaxf(tmax, fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3])));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));        // both halves of a stationary row agree
    if (__any(tmax > m_run + kDefer)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int c = 0; c < B::CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) gacc[c][r] *= alpha;
      m_run = m_new;
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[r] = __builtin_amdgcn_exp2f(acc[r] - m_run);
      sum += acc[r];
    }
    l_run += sum;
  };

  {
    float4 regs[S::NLD];
    if (tile0 < tile1) {
      stage_load<D>(y, y_scale, ny, tile0 * kTileJ, tid, regs);
#pragma unroll
      for (int u = 0; u < S::NLD; ++u) stage_store_b3t_one<D>(lds[0], tid, regs[u], u);
    }
    __syncthreads();
    for (int64_t tt = tile0; tt < tile1; ++tt) {
      const int cur = (int)((tt - tile0) & 1);
      const bool more = tt + 1 < tile1;
      if (more) stage_load<D>(y, y_scale, ny, (tt + 1) * kTileJ, tid, regs);
      f32x16 acc[1];
      score_tile_b3<D, 1>(lds[cur % NBUF], i32, h, bq, acc);
      softmax_p(acc[0], tt);
      const unsigned char* tbase = lds[cur % NBUF] + 3 * S::PLANE + i32 * B::RT + 8 * h;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        u32x4 pp[3];
        {
          unsigned q[3][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split3(acc[0][8 * kc + 2 * e], acc[0][8 * kc + 2 * e + 1], q[0][e], q[1][e], q[2][e]);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) pp[pl] = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
        }
#pragma unroll
        for (int c = 0; c < B::CT; ++c) {
          u32x4 ya[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const unsigned char* p = tbase + pl * B::TPLANE + (32 * c) * B::RT + 32 * kc;
            const uint2 lo = *reinterpret_cast<const uint2*>(p);
            const uint2 hi = *reinterpret_cast<const uint2*>(p + 16);
            ya[pl] = (u32x4){lo.x, lo.y, hi.x, hi.y};
          }
          gacc[c] = mfma_bf16(ya[2], pp[0], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[2], gacc[c]);
          gacc[c] = mfma_bf16(ya[1], pp[1], gacc[c]);
          gacc[c] = mfma_bf16(ya[1], pp[0], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[1], gacc[c]);
          gacc[c] = mfma_bf16(ya[0], pp[0], gacc[c]);
        }
      }
      if (NBUF == 1) __syncthreads();          // every wave is done reading the only buffer
      if (more) {
#pragma unroll
        for (int u = 0; u < S::NLD; ++u) stage_store_b3t_one<D>(lds[(cur ^ 1) % NBUF], tid, regs[u], u);
      }
      __syncthreads();
    }
  }

  // per-split partial: (reference point, sum of both halves) and the un-normalised O rows
  const float l_o = __shfl_xor(l_run, 32, 64);
  if (h == 0 && row_i < mx) part[(int64_t)split * mx + row_i] = make_float2(m_run, l_run + l_o);
  float* gout = opart + (int64_t)split * mx * D;
  if (row_i < mx) {
#pragma unroll
    for (int c = 0; c < B::CT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(gacc[c][4 * g + 0], gacc[c][4 * g + 1], gacc[c][4 * g + 2], gacc[c][4 * g + 3]);
        *reinterpret_cast<float4*>(gout + row_i * D + 32 * c + 8 * g + 4 * h) = v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// Two-product tile loop, software-pipelined ACROSS tiles (d <= 64): the exposed part of the loops above is
// the VALU work between the two MFMA phases (P from the scores + its bf16 split: ~150 instructions per
// tile against ~1500 cycles of MFMA, ~50 % MFMA-busy measured).  Here step t runs
//     phase A   score MFMAs of tile t+1          ||  operand split + LDS stores of tile t+2 (staging)
//     phase B   second-product MFMAs of tile t   ||  P(t+1) from the finished scores + its bf16 split
// so every VALU instruction sits in an MFMA shadow; one barrier per step.  LDS: ONE row-major image of the
// three bf16 planes per tile serves both products — `ds_read_b128` rows for the score operand, the
// transposing `ds_read_b64_tr_b16` for the second product's operand (the 2-byte stores of a transposed copy
// were 8-way bank-conflicted and made the loop LDS-bound) — in a ring of three tiles (t read in B, t+1 read in
// A, t+2 being written in A): 3 x 13.5 KB = 40.5 KB.
//   MODE 0  backward: P = w_x e^{s - lse_x} + w_y e^{s - lse_y}   (infonce_bwd_b3_kernel's arithmetic)
//   MODE 1  forward with the weighted row sum (infonce_fwdo_b3_kernel's arithmetic): online reference point,
//           the rare rescale of the accumulators happens between two B phases.
//   MODE 2  BCE-with-logits forward (lightgcn.py:109-113): row sums of softplus(s) and o_i = sum_j sigmoid(s_ij) y_j
//           (sigmoid <= 1: no reference point, no rescale); MODE 3 its backward, P = (w_x[i] + w_y[j]) sigmoid(s_ij)
//           (SIDES 1 / 2: the weights sit on the stationary / the streamed rows).
// ------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_read_tr16(const unsigned char* p) {      // ds_read_b64_tr_b16; EXEC must be all ones
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p));
}

// max |w| over both weight vectors -> hw[0] = 2^14 / 2^ceil(log2 max|w|) (what the weights are multiplied by before
// they meet e^{s - lse} <= 1, so that P stays in f16 range), hw[1] = 1 / (hw[0] * 2^8) (undoes it, and the streamed
// operand's 2^8, on the way out).  One block.
__global__ __launch_bounds__(1024) void h2_wscale_kernel(const float* __restrict__ w_x, int64_t mx,
                                                         const float* __restrict__ w_y, int64_t ny, float* __restrict__ hw) {
  __shared__ float red[16];
  float m = 0.f;
  if (w_x != nullptr)
    for (int64_t i = threadIdx.x; i < mx; i += 1024) m = fmaxf(m, fabsf(w_x[i]));
  if (w_y != nullptr)
    for (int64_t i = threadIdx.x; i < ny; i += 1024) m = fmaxf(m, fabsf(w_y[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
    int e = 0;
    if (m > 0.f && m < INFINITY) (void)frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1): |w| 2^-e < 1
    e = min(max(e, -100), 100);
    hw[0] = ldexpf(1.0f, 14 - e);
    hw[1] = 1.0f / (hw[0] * EngH2::kSY);
  }
}

// Pre-pass of the two-f16-plane backward loop (MODE 0): a scaled, zero-padded image of the STREAMED rows and of their
// statistics, so that the loop stages a tile with two plain loads per lane — no scale loads, multiplies, end-of-array
// clamps or selects per tile.  The loop's time is its count of vector instructions, and the streamed side of
// a backward launch is read once per row block: mx / 128 times.
//   FOLDW = false:  l_j = lse_j log2 e,  w'_j = w_j hw[0] 2^-l_j,  yhat_j = y_scale_j kSY y_j   (padding: 1e30, 0, zero row)
//   FOLDW = true (statistics on the streamed rows only, SIDES = 2): the weight of row j moves INTO the exponent and its
//   sign into the row,
//     P_ij y_j = w_j e^{s_ij - lse_j} y_j = 2^(c_j s'_ij + e_j) (sgn_j y_j),   s'_ij = x_i . (sgn_j y_j) the score the loop sees,
//     e_j = log2(|w_j| hw[0]) - lse_j log2 e,   c_j = sgn_j kSInv,   yhat_j = sgn_j y_scale_j kSY y_j,
//   so that a probability costs one fma and one exp2 in the loop instead of fma, exp2 and a multiply.  w_j = 0 gives
//   e_j = -inf, i.e. P = 0, like the padding rows (e = -1e30, zero row).
// Layout of `img`: l or e [rows], w' or c [rows], yhat[rows][D], rows = fold_rows(ny).
template <int D, bool FOLDW>
__global__ __launch_bounds__(256) void h2_prestage_kernel(const float* __restrict__ y, const float* __restrict__ y_scale,
                                                          const float* __restrict__ w_y, const float* __restrict__ lse_y,
                                                          int64_t ny, const float* __restrict__ hw, float* __restrict__ img,
                                                          bool write_stats) {
  const int64_t rows = fold_rows(ny);
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= rows * (D / 4)) return;
  const int64_t j = k / (D / 4);
  const int c4 = (int)(k % (D / 4));
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float s0 = FOLDW ? -1.0e30f : 1.0e30f, s1 = FOLDW ? EngH2::kSInv : 0.f;
  if (j < ny) {
    float sgn = 1.0f;
    if (w_y != nullptr) {
      const float w = w_y[j] * hw[0];
      if (FOLDW) {
        sgn = w < 0.f ? -1.0f : 1.0f;
        s0 = fmaf(-lse_y[j], kLog2e, __log2f(fabsf(w)));
        s1 = sgn * EngH2::kSInv;
      } else {
        // statistics on BOTH sides (the only user of this form): the loop factors the common 2^s out of
        //   w_i 2^(s - l_i) + w_j 2^(s - l_j) = 2^s (w_i 2^-l_i + w_j 2^-l_j),
        // so the streamed row's weight arrives multiplied by its 2^-l (l >= every score of the row: no overflow)
        s0 = lse_y[j] * kLog2e;
        s1 = w * __builtin_amdgcn_exp2f(-s0);
      }
    }
    const float sc = sgn * EngH2::kSY * (y_scale != nullptr ? y_scale[j] : 1.0f);
    v = *reinterpret_cast<const float4*>(y + j * D + 4 * c4);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
  }
  *reinterpret_cast<float4*>(img + 2 * rows + j * D + 4 * c4) = v;
  if (c4 == 0 && write_stats) {                            // (w_y == nullptr: every row like the padding)
    img[j] = s0;
    img[rows + j] = s1;
  }
}

// (LDS rows of 36 dwords: the 16 rows of every ds_read_b128 lane group start in 16 different bank quads, but rows q
// and q + 2 of the four that one ds_read_b64_tr_b16 gathers share 8 banks — every transposing read takes two passes,
// 22-28 % of the loop's LDS cycles are conflicts (SQ_LDS_BANK_CONFLICT).  Placing tile row 16kc + 8g + 4h + q in LDS row
// 16kc + 4q + 2g + h removes them and was measured: no change in time (the LDS is 22-27 % busy), so the plain order stays.)
// NW = 8: a 512-thread workgroup of 256 stationary rows.  Its waves 4..7 run the same stream ONE barrier interval behind
// waves 0..3 (two barriers per step: after the score phase and after the P phase), so that on every SIMD one wave is in
// its MFMA-heavy score phase while its partner is in its VALU-heavy P phase, and both halves share one staged tile
// (each lane stages one float4 of it in its own score phase): half the staging work per MFMA.  Ring of four tiles.
template <class E, int D, int MODE, bool EXD, int SIDES, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : E::kMinBlocks) void infonce_pipe_kernel(
    const float* __restrict__ x, const float* __restrict__ x_scale, int64_t mx, const float* __restrict__ y,
    const float* __restrict__ y_scale, int64_t ny, float scale2, float out_scale, const float* __restrict__ lse_x,
    const float* __restrict__ w_x, const float* __restrict__ lse_y, const float* __restrict__ w_y, int nsplit,
    int64_t tiles_per_split, float* __restrict__ gpart, float2* __restrict__ part, const float* __restrict__ hw) {
  using S = ShapeB3<D>;
  using B = BwdB3<D>;
  static_assert(D <= 64 || (D == 128 && std::is_same<E, EngH2>::value), "d = 128: two planes only (registers)");
  constexpr int NPL = E::NPL, NTERM = E::NTERM;
  constexpr int RM = NPL * S::PLANE;
  constexpr int THREADS = 64 * NW;
  constexpr int F4T = kTileJ * D / 4;                        // float4s per streamed tile
  constexpr int NLDW = (F4T + THREADS - 1) / THREADS;        // ... per thread
  constexpr int RING = NW == 8 ? 4 : 3;
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  // PRE (h2_prestage_kernel ran in front): y = the scaled rows, lse_y / w_y = the streamed rows' statistics in the loop's
  // units, all padded to whole tiles; FOLD: the statistics are (e, c), the weights live in the exponent
  constexpr bool PRE = MODE == 0 && std::is_same<E, EngH2>::value;
  constexpr bool FOLD = PRE && SIDES == 2;
  constexpr bool BWD = MODE == 0 || MODE == 3;               // per-row weights (MODE 0: and lse) ride with the tiles
  constexpr bool FWD = MODE == 1 || MODE == 2;               // running row sums + the o accumulators
  constexpr bool NOO = MODE == 2 && SIDES == 1;              // BCE row sums only (no gradient wanted): no second product
  // UNS: BCE on two f16 planes.  Its rows are not unit rows, so the launch scales both sides to unit norm (x_scale /
  // y_scale = 1 / |row|) and every score is UN-SCALED by the two norms before the softplus / sigmoid: lse_x / lse_y carry
  // the norms, w_y the streamed rows' factor of the second product (norm x weight, pre-scaled into f16 range by hw[0]).
  constexpr bool UNS = MODE >= 2 && std::is_same<E, EngH2>::value;
  constexpr bool STATS = (BWD && SIDES != 1) || UNS;         // per-row values of the streamed rows ride with the tiles
  // UNSREF (UNS with a second product): the probabilities sigmoid x (norm x weight) go into f16 planes, whose floor would
  // flush a row whose sigmoids are ALL tiny.  Every stationary row therefore carries a running power-of-two reference 2^q,
  // q ~ log2 of the largest sigmoid seen so far (min(s_max, 0): within one bit of it), the planes hold sigmoid 2^-q x
  // (norm x weight) <= 2^15, the accumulators are rescaled when a tile raises q by more than one (the flash forward's
  // deferred rescale, rare after the first tiles) and the result is multiplied by 2^q on the way out.
  constexpr bool UNSREF = UNS && !NOO;
  static_assert(MODE < 2 || (!EXD && NW == 4), "BCE modes: four waves");
  static_assert(MODE != 3 || SIDES == 1 || SIDES == 2, "BCE backward: weights on one side");
  static_assert(!(UNS && MODE == 3 && SIDES == 1), "two-plane BCE backward: the weights sit on the streamed rows");
  __shared__ __align__(16) unsigned char lds_rm[RING][RM];   // ring: tile t (B, transposed reads), t+1 (A), t+2 (staging)
  __shared__ __align__(16) float st_lse[RING][kTileJ];
  __shared__ __align__(16) float st_w[RING][kTileJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t mblk = blockIdx.x / nsplit;
  const int split = blockIdx.x % nsplit;
  const int64_t row_i = (mblk * NW + wave) * 32 + i32;

  // MODE 0 on EngH2: weights pre-scaled into f16 range (h2_wscale_kernel); hw == nullptr otherwise
  const float w_mul = hw != nullptr ? hw[0] : 1.0f;
  const bool on_x = BWD && row_i < mx && w_x != nullptr;
  const float wl = on_x ? w_x[row_i] * w_mul : 0.f;
  const float lse2l = (MODE == 0 && on_x) ? lse_x[row_i] * kLog2e : 1.0e30f;
  const float rs = (UNS && row_i < mx) ? lse_x[row_i] : 1.0f;   // UNS: the stationary row's norm
  // BOTHF (two planes, statistics on both sides): one exp2 per probability — the unit rows bound the score (|s| <= 28.9
  // in log2 units), so 2^s is taken once and multiplied by w_i 2^-l_i + w_j 2^-l_j (the streamed half premultiplied by
  // h2_prestage_kernel): exp2 + add + multiply instead of two fused multiply-adds, two exp2, two multiplies and an add
  constexpr bool BOTHF = PRE && SIDES == 0;
  const float ax = BOTHF ? wl * __builtin_amdgcn_exp2f(-lse2l) : 0.f;
  // FOLDX (statistics on the stationary rows only): the same move as FOLD inside the kernel — |w_i| into the exponent,
  // sgn w_i into the stationary operand (the scores flip with it, hence c_x) and back out of the gradient row at the end:
  //   P_ij = w_i e^{s_ij - lse_i} = sgn_i 2^(c_x s'_ij + e_x)
  constexpr bool FOLDX = MODE == 0 && SIDES == 1;
  const float sgn_x = FOLDX && wl < 0.f ? -1.0f : 1.0f;
  const float c_x = sgn_x * E::kSInv;
  const float e_x = (MODE == 0 && on_x) ? fmaf(-lse_x[row_i], kLog2e, __log2f(fabsf(wl))) : -1.0e30f;
  u32x4 bq[1][NPL][S::KC];
  load_stationary_e<E, D>(x, x_scale, mx, row_i, h, scale2 * E::kSX * sgn_x, bq[0]);
  float m_run = UNSREF ? -100.0f : kNegBig, l_run = 0.f;   // MODE 1: reference point of the row; UNSREF: q
  f32x16 gacc[B::CT];
#pragma unroll
  for (int c = 0; c < B::CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) gacc[c][r] = 0.f;

  const int64_t total_tiles = (ny + kTileJ - 1) / kTileJ;
  const int64_t tile0 = (int64_t)split * tiles_per_split;
  const int64_t tile1 = min(total_tiles, tile0 + tiles_per_split);
  if (tile0 < tile1) {
    constexpr int NP = 3 * NLDW, NS1 = NTERM * S::KC, NG = B::CT * 2 * NTERM;
    const int64_t last = tile1 - 1;
    StagedRows<NLDW> ra, rb;                                // (raw rows + their scales: multiplied where they are staged)
    float sla = 0.f, swa = 0.f, slb = 0.f, swb = 0.f;
    auto load_tile = [&](int64_t t, StagedRows<NLDW>& r, float& sl, float& sw) {
      const int64_t j0 = min(t, last) * kTileJ;
      // wave-uniform tile base + 32-bit per-thread offsets (a ragged last tile clamps its rows to the last valid
      // one and zeroes their scale): no 64-bit vector arithmetic in the loop
      const int rem = (int)min((int64_t)kTileJ, ny - j0);
      const float* tb = y + j0 * D;
      const float* sp = y_scale != nullptr ? y_scale + j0 : y;               // (no scale: a dummy word through a zero stride)
      const int sstride = y_scale != nullptr ? 1 : 0;
      if (PRE) {                                             // scaled and padded already: uniform base + a constant offset per lane
#pragma unroll
        for (int u = 0; u < NLDW; ++u)
          r.v[u] = *reinterpret_cast<const float4*>(tb + 4u * (unsigned)min(tid + THREADS * u, F4T - 1));
        if (SIDES != 1 && tid < kTileJ) {
          sw = w_y[j0 + tid];
          sl = lse_y[j0 + tid];
        }
        return;
      }
#pragma unroll
      for (int u = 0; u < NLDW; ++u) {
        const int idx = min(tid + THREADS * u, F4T - 1);     // (more threads than float4s: the surplus reloads the last one)
        const int row = idx / (D / 4), c4 = idx % (D / 4);
        const unsigned rr = (unsigned)min(row, rem - 1) & (kTileJ - 1);   // (masked: a provably small offset folds into the load)
        // loads only: the scale is multiplied in where the row is staged, two steps from here (doing it on the spot put
        // a `s_waitcnt vmcnt(0)` behind the loads — a memory round trip in front of every step's first MFMA)
        r.v[u] = *reinterpret_cast<const float4*>(tb + (rr * D + 4u * c4));   // uniform base + 32-bit lane offset
        r.s[u] = sp[rr * (unsigned)sstride];
        r.live[u] = row < rem;
      }
      if ((BWD || UNS) && tid < kTileJ) {
        const int64_t j = j0 + tid;
        const bool on = j < ny && w_y != nullptr;
        sw = on ? w_y[j] * w_mul : 0.f;
        if (UNS) sl = j < ny ? lse_y[j] : 1.0f;             // the streamed row's norm (a masked row: -inf x 1)
        else sl = (MODE == 0 && on) ? lse_y[j] * kLog2e : 1.0e30f;
      }
    };
    // one third of the staging of one float4: split (x, y), split (z, w), row-major plane stores
    auto stage_part = [&](int pi, const StagedRows<NLDW>& st, float4 (&tm)[NLDW], unsigned (&sa)[NLDW][NPL],
                          unsigned (&sb)[NLDW][NPL], unsigned char* rm, int sbuf, float sl, float sw) {
      const int u = pi / 3, k = pi % 3;
      const int idx = tid + THREADS * u;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      if (k == 0) {
        tm[u] = st.v[u];
        if (!PRE) {
          const float sc = st.live[u] ? (y_scale != nullptr ? st.s[u] * E::kSY : E::kSY) : 0.f;
          tm[u].x *= sc; tm[u].y *= sc; tm[u].z *= sc; tm[u].w *= sc;
        }
        E::template split<kSplitForm<MODE>>(tm[u].x, tm[u].y, sa[u]);
      } else if (k == 1) {
        E::template split<kSplitForm<MODE>>(tm[u].z, tm[u].w, sb[u]);
      } else {
        unsigned char* p = rm + row * S::ROWB + c4 * 8;
        if (THREADS * NLDW == F4T || idx < F4T) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint2*>(p + pl * S::PLANE) = make_uint2(sa[u][pl], sb[u][pl]);
        }
        if ((BWD || UNS) && u == 0 && tid < kTileJ) {
          st_lse[sbuf][tid] = sl;
          st_w[sbuf][tid] = sw;
        }
      }
    };
    auto stage_all = [&](const StagedRows<NLDW>& st, unsigned char* rm, int sbuf, float sl, float sw) {
      unsigned sa[NLDW][NPL], sb[NLDW][NPL];
      float4 tm[NLDW];
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) stage_part(pi, st, tm, sa, sb, rm, sbuf, sl, sw);
    };
    // micro-unit m of P(t) from the finished scores of tile t: MODE 0: 0..15 one register each, 16..23 one split3
    // each; MODE 1: the same after `prepare` fixed the reference point.
    float m_use = kNegBig, alpha = 1.0f, psum = 0.f, p_off = 0.f;
    bool rescale = false;
    auto prepare = [&](f32x16& acc, int64_t t, int sbuf) {   // masks (ragged tile / excluded diagonal); MODE 1: reference point
      const int64_t j0 = t * kTileJ;
      const int lim = (int)min((int64_t)kTileJ, ny - j0);   // rows of this tile that exist (wave-uniform)
      // (statistics on the streamed rows only, MODE 0 / SIDES 2: no end-of-array mask — rows behind the end carry w' = 0
      // or e = -1e30, P = 0 whatever their score.  Everywhere else they need it: such a row is staged as a zero row, score
      // 0, and against real scores that are all far below 0 its stationary-side "probability" e^{0 - lse_i} overflows
      // the f16 planes — inf times the zero row.  Under FOLDX the masked score is -inf / c_x: +inf for a negative weight.)
      if (EXD || (lim < kTileJ && !(MODE == 0 && SIDES == 2))) {
        // a real (scalar) branch: if-converted, the 16 selects with their compares would run for every tile
        if (!EXD) asm volatile("" ::: "memory");
        const int xr = EXD ? diag_offset(row_i, j0, h) : -1;
        const float dead_score = FOLDX ? -INFINITY * sgn_x : -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const bool dead = (acc_row(r, h) >= lim) || (EXD && xr == e + 8 * g);
            acc[r] = dead ? dead_score : acc[r];
          }
      }
      if (MODE == 1) {
        float tmax = acc[0];                              // chained: one v_max3_f32 per pair of scores
#pragma unroll
        for (int r = 1; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, acc[r]), acc[r + 1 < 16 ? r + 1 : r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64)) * E::kSInv;
        rescale = __any(tmax > m_run + E::kDeferE);
        m_use = rescale ? fmaxf(m_run, tmax) : m_run;
        alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        psum = 0.f;
        p_off = E::kPExp - m_use;
      }
      if (MODE == 2) psum = 0.f;
      if (UNSREF) {
        // scores un-scaled here (so that their maximum is known before the first probability), q from the largest one
        float smax = -INFINITY;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 l4 = *reinterpret_cast<const float4*>(&st_lse[sbuf][8 * g + 4 * h]);
          acc[4 * g + 0] *= rs * l4.x; acc[4 * g + 1] *= rs * l4.y; acc[4 * g + 2] *= rs * l4.z; acc[4 * g + 3] *= rs * l4.w;
          smax = fmaxf(fmaxf(fmaxf(fmaxf(smax, acc[4 * g]), acc[4 * g + 1]), acc[4 * g + 2]), acc[4 * g + 3]);
        }
        smax = fmaxf(smax, __shfl_xor(smax, 32, 64));       // both lane halves feed the same output row
        const float q_tile = fmaxf(fminf(smax, 0.f), -100.0f);
        rescale = __any(q_tile > m_run + 1.0f);
        m_use = rescale ? fmaxf(m_run, q_tile) : m_run;
        alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        p_off = __builtin_amdgcn_exp2f(-m_use);             // 2^-q: what a probability is multiplied by
      }
    };
    // streamed rows' statistics (MODE 0, SIDES != 1): accumulator register r = 4 g + e of lane half h is tile row
    // 8 g + 4 h + e, so ONE ds_read_b128 per array serves four consecutive P units; group g + 1 is fetched while group g
    // is being used (16 registers).  Per-unit ds_read_b32 pairs were 32 LDS instructions per tile: 9 % of the backward
    // (timing-only build with constants: 1.82 -> 1.66 ms).
    float4 sl4[2], sw4[2];
    auto stats_begin = [&](int sbuf) {
      if (STATS) {
        sl4[0] = *reinterpret_cast<const float4*>(&st_lse[sbuf][4 * h]);
        sw4[0] = *reinterpret_cast<const float4*>(&st_w[sbuf][4 * h]);
      }
    };
    auto p_unit = [&](int m, f32x16& acc, int sbuf, unsigned (&pq)[2][NPL][4]) {
      if (m < 16) {
        const int r = m;
        float lre = 0.f, wre = 0.f;
        if (STATS) {
          const int g = r >> 2, e = r & 3;
          if (e == 0 && g < 3) {
            sl4[(g + 1) & 1] = *reinterpret_cast<const float4*>(&st_lse[sbuf][8 * (g + 1) + 4 * h]);
            sw4[(g + 1) & 1] = *reinterpret_cast<const float4*>(&st_w[sbuf][8 * (g + 1) + 4 * h]);
          }
          const float4 l4 = sl4[g & 1], w4 = sw4[g & 1];
          lre = e == 0 ? l4.x : (e == 1 ? l4.y : (e == 2 ? l4.z : l4.w));
          wre = e == 0 ? w4.x : (e == 1 ? w4.y : (e == 2 ? w4.z : w4.w));
        }
        const float unscale = UNSREF ? 1.0f : (UNS ? rs * lre : E::kSInv);   // accumulator -> log2-domain score (UNSREF: done in `prepare`)
        if (MODE == 1) {
          acc[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], E::kSInv, p_off));
          psum += acc[r];
        } else if (MODE == 2) {
          float sp, sg;
          bce_terms(acc[r] * unscale, sp, sg);
          psum += sp;
          acc[r] = UNSREF ? sg * (wre * p_off) : (UNS ? sg * wre : sg);
        } else {
          const float sc = acc[r];
          if (MODE == 3) {
            acc[r] = (SIDES == 1 ? wl : (UNSREF ? wre * p_off : wre)) * bce_sigmoid(sc * unscale);
          } else if (SIDES == 1) acc[r] = __builtin_amdgcn_exp2f(fmaf(sc, c_x, e_x));
          else if (FOLD) acc[r] = __builtin_amdgcn_exp2f(fmaf(sc, wre, lre));
          else if (SIDES == 2) acc[r] = wre * __builtin_amdgcn_exp2f(fmaf(sc, E::kSInv, -lre));
          else if (BOTHF) acc[r] = __builtin_amdgcn_exp2f(sc * E::kSInv) * (ax + wre);
          else acc[r] = wl * __builtin_amdgcn_exp2f(fmaf(sc, E::kSInv, -lse2l)) + wre * __builtin_amdgcn_exp2f(fmaf(sc, E::kSInv, -lre));
        }
      } else {
        if (NOO) return;                                 // (no second product: nothing to split)
        const int e = m - 16;                            // pair (2e, 2e + 1): k-chunk e / 4, dword e % 4
        unsigned q[NPL];
        E::template split<kSplitForm<MODE>>(acc[2 * e], acc[2 * e + 1], q);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) pq[e >> 2][pl][e & 3] = q[pl];
      }
    };
    auto finish_p = [&]() {                               // MODE 1: between two B phases
      if (MODE == 1) {
        if (rescale) {
          l_run *= alpha;
#pragma unroll
          for (int c = 0; c < B::CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[c][r] *= alpha;
          m_run = m_use;
        }
        l_run += psum;
      }
      if (MODE == 2) l_run += psum;
      if (UNSREF && rescale) {
#pragma unroll
        for (int c = 0; c < B::CT; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) gacc[c][r] *= alpha;
        m_run = m_use;
      }
    };
    auto score_plain = [&](const unsigned char* rm, f32x16& acc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const unsigned char* base = rm + i32 * S::ROWB + h * (S::KH * 2);
#pragma unroll
      for (int c = 0; c < S::KC; ++c) {
        u32x4 ap[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) ap[pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE + 16 * c);
#pragma unroll
        for (int term = 0; term < NTERM; ++term) acc = E::mfma(ap[E::ta(term)], bq[0][E::tb(term)][c], acc);
      }
    };

    // ---- prologue: tiles 0 and 1 staged, P(tile0) ready, tile 2 in registers, tile 3 loading
    unsigned pqa[2][NPL][4], pqb[2][NPL][4];
    f32x16 acc;
    load_tile(tile0, ra, sla, swa);
    stage_all(ra, lds_rm[0], 0, sla, swa);
    load_tile(tile0 + 1, ra, sla, swa);
    stage_all(ra, lds_rm[1], 1, sla, swa);
    load_tile(tile0 + 2, ra, sla, swa);
    __syncthreads();
    score_plain(lds_rm[0], acc);
    prepare(acc, tile0, 0);
    stats_begin(0);
#pragma unroll
    for (int m = 0; m < 24; ++m) p_unit(m, acc, 0, pqa);
    finish_p();
    __syncthreads();

    // step t: tile t's P planes in `pc`, tile t+2 in `st` registers; produces P(t+1) in `pn`, loads tile t+3 to `ld`
    auto step = [&](auto next_c, int64_t t, int k3, unsigned (&pc)[2][NPL][4], unsigned (&pn)[2][NPL][4],
                    const StagedRows<NLDW>& st, float st_l, float st_w_v, StagedRows<NLDW>& ld, float& ld_l, float& ld_w) {
      constexpr bool real_next = decltype(next_c)::value;    // false only for the split's last tile (compile time:
                                                             // no branch may sit between the MFMAs of a phase)
      load_tile(t + 3, ld, ld_l, ld_w);
      // tile t in ring slot k3, t+1 in k3 + 1, t+2 goes to k3 + 2 (planes and per-tile statistics alike)
      const int slot1 = (k3 + 1) % RING, slot2 = (k3 + 2) % RING;
      unsigned char* rm_out = lds_rm[slot2];
      unsigned sa[NLDW][NPL], sb[NLDW][NPL];
      float4 tm[NLDW];
      // phase A: S^T of tile t+1 || staging of tile t+2
      {
        const unsigned char* base = lds_rm[slot1] + i32 * S::ROWB + h * (S::KH * 2);
        u32x4 ap[2][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) ap[0][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE);
#pragma unroll
        for (int c = 0; c < S::KC; ++c) {
          if (c + 1 < S::KC) {
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
              ap[(c + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE + 16 * (c + 1));
          }
#pragma unroll
          for (int term = 0; term < NTERM; ++term) {
            const int slot = c * NTERM + term;
            f32x16 cin = acc;
            if (slot == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) cin[r] = 0.f;
            }
            acc = E::mfma(ap[c & 1][E::ta(term)], bq[0][E::tb(term)][c], cin);
#pragma unroll
            for (int pi = slot * NP / NS1; pi < (slot + 1) * NP / NS1; ++pi)
              stage_part(pi, st, tm, sa, sb, rm_out, slot2, st_l, st_w_v);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (NW == 8) __syncthreads();                      // interval boundary: the partner group moves on to its score phase
      if (BWD || real_next) prepare(acc, t + 1, slot1);
      stats_begin(slot1);
      __builtin_amdgcn_sched_barrier(0);
      // phase B: second product of tile t || P(t+1)
      // A operand yhat^T[feature][tile row] straight from the ROW-MAJOR planes with the transposing LDS read
      // (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, lane 4q + p supplies the address of
      // (row q, columns 4p..4p+3) and receives column `lane & 15`, rows 0..3): no transposed copy of the tile, no
      // 2-byte stores.  Fragment element j of lane half h is tile row 16 kc + 8 (j >> 2) + 4 h + (j & 3), the
      // k order of the accumulator-as-operand P planes.
      const unsigned char* tbase = lds_rm[k3] + (4 * h + ((lane & 15) >> 2)) * S::ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
      auto load_ya = [&](int grp, u32x4 (&ya)[NPL]) {     // grp = kc * CT + c
        const int kc = grp / B::CT, c = grp % B::CT;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const unsigned char* p = tbase + pl * S::PLANE + (16 * kc) * S::ROWB + 64 * c;
          const uint2 lo = lds_read_tr16(p);
          const uint2 hi = lds_read_tr16(p + 8 * S::ROWB);
          ya[pl] = (u32x4){lo.x, lo.y, hi.x, hi.y};
        }
      };
      u32x4 ya[2][NPL];
      if (!NOO) load_ya(0, ya[0]);
#pragma unroll
      for (int grp = 0; grp < 2 * B::CT; ++grp) {
        const int kc = grp / B::CT, c = grp % B::CT;
        if (!NOO && grp + 1 < 2 * B::CT) load_ya(grp + 1, ya[(grp + 1) & 1]);
        u32x4 pp[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) pp[pl] = (u32x4){pc[kc][pl][0], pc[kc][pl][1], pc[kc][pl][2], pc[kc][pl][3]};
#pragma unroll
        for (int term = 0; term < NTERM; ++term) {
          if (!NOO) gacc[c] = E::mfma(ya[grp & 1][E::ta(term)], pp[E::tb(term)], gacc[c]);
          const int slot = grp * NTERM + term;
          if (BWD || real_next) {
#pragma unroll
            for (int m = slot * 24 / NG; m < (slot + 1) * 24 / NG; ++m) p_unit(m, acc, slot1, pn);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (BWD || real_next) finish_p();
      __syncthreads();
    };
    // NW = 8: waves 4..7 take one barrier more before the loop and waves 0..3 one more after it: the same number for
    // all, the second group one interval behind
    // (the group test on a scalar: a barrier inside a vector-divergent branch could be executed with an empty EXEC mask)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if (NW == 8 && wave_s >= 4) __syncthreads();
    int k3 = 0;
    int64_t tt = tile0;
    for (; tt + 2 < tile1; tt += 2) {
      step(std::true_type{}, tt, k3, pqa, pqb, ra, sla, swa, rb, slb, swb);
      k3 = (k3 + 1) % RING;
      step(std::true_type{}, tt + 1, k3, pqb, pqa, rb, slb, swb, ra, sla, swa);
      k3 = (k3 + 1) % RING;
    }
    if (tt + 1 < tile1) {                                  // two tiles left
      step(std::true_type{}, tt, k3, pqa, pqb, ra, sla, swa, rb, slb, swb);
      k3 = (k3 + 1) % RING;
      step(std::false_type{}, tt + 1, k3, pqb, pqa, rb, slb, swb, ra, sla, swa);
    } else {                                               // one tile left
      step(std::false_type{}, tt, k3, pqa, pqb, ra, sla, swa, rb, slb, swb);
    }
    if (NW == 8 && wave_s < 4) __syncthreads();
  }

  float* gout = gpart + (int64_t)split * mx * D;
  if (FWD) {
    const float l_o = __shfl_xor(l_run, 32, 64);
    // the sum carries the scale of P (2^kPExp), the rows P's and the streamed operand's; MODE 2: plain log2-domain sum
    if (h == 0 && row_i < mx)
      part[(int64_t)split * mx + row_i] =
          MODE == 2 ? make_float2(0.f, l_run + l_o) : make_float2(m_run, (l_run + l_o) * __builtin_amdgcn_exp2f(-E::kPExp));
  }
  const float o_mul = out_scale * (hw != nullptr ? hw[1] : 1.0f) * sgn_x * (UNSREF ? __builtin_amdgcn_exp2f(m_run) : 1.0f);
  if (!NOO && row_i < mx) {
#pragma unroll
    for (int c = 0; c < B::CT; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(gacc[c][4 * g + 0] * o_mul, gacc[c][4 * g + 1] * o_mul,
                               gacc[c][4 * g + 2] * o_mul, gacc[c][4 * g + 3] * o_mul);
        *reinterpret_cast<float4*>(gout + row_i * D + 32 * c + 8 * g + 4 * h) = v;
      }
  }
}

// lse and o from the per-split partials: o = sum_s O_s 2^(m_s - m) / sum_s l_s 2^(m_s - m)
__global__ __launch_bounds__(256) void infonce_merge_o_kernel(const float2* __restrict__ part, const float* __restrict__ opart,
                                                              int nsplit, int64_t m_rows, int d, float* __restrict__ lse,
                                                              float* __restrict__ o) {
  const int d4 = d / 4;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m_rows * d4) return;
  const int64_t i = k / d4;
  const int c = (int)(k % d4);
  float m = kNegBig;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part[(int64_t)s * m_rows + i].x);
  float l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < nsplit; ++s) {
    const float2 p = part[(int64_t)s * m_rows + i];
    const float w = __builtin_amdgcn_exp2f(p.x - m);
    l += p.y * w;
    const float4 v = reinterpret_cast<const float4*>(opart)[((int64_t)s * m_rows + i) * d4 + c];
    acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
  }
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  reinterpret_cast<float4*>(o)[i * d4 + c] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (c == 0) lse[i] = (m + __log2f(l)) * kLn2;
}

// g[i, :] = sum over splits (fixed order) of the partial gradients
__global__ __launch_bounds__(256) void bwd_reduce_kernel(const float* __restrict__ gpart, int nsplit, int64_t n4,
                                                         float* __restrict__ g) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
    float4 s = reinterpret_cast<const float4*>(gpart)[k];
    for (int p = 1; p < nsplit; ++p) {
      const float4 v = reinterpret_cast<const float4*>(gpart)[(int64_t)p * n4 + k];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(g)[k] = s;
  }
}

// positive-logit term: gx[i] += c_i * inv_tau * yhat[p_i] (row-exclusive), gy[p_i] += c_i * inv_tau * xhat[i] (atomic)
__global__ __launch_bounds__(256) void pos_bwd_kernel(const float* __restrict__ x, const float* __restrict__ x_scale,
                                                      const float* __restrict__ y, const float* __restrict__ y_scale,
                                                      const int64_t* __restrict__ pos, const float* __restrict__ coef,
                                                      int64_t mx, int64_t ny, int d, float inv_tau, float* gx,
                                                      float* gy) {
  const int lane = threadIdx.x & 63;
  for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < mx; i += (int64_t)gridDim.x * 4) {
    const int64_t p = pos != nullptr ? pos[i] : i;
    if (p < 0 || p >= ny) continue;
    const float c = coef[i] * inv_tau;
    const float sx = x_scale != nullptr ? x_scale[i] : 1.0f;
    const float sy = y_scale != nullptr ? y_scale[p] : 1.0f;
    for (int col = lane; col < d; col += 64) {
      if (gx != nullptr) gx[i * d + col] += c * sy * y[p * d + col];
      if (gy != nullptr) atomicAdd(gy + p * d + col, c * sx * x[i * d + col]);
    }
  }
}

// through F.normalize: out = inv * (ghat - xhat <xhat, ghat>), xhat = x * inv  (out may alias ghat)
__global__ __launch_bounds__(256) void normalize_bwd_kernel(const float* __restrict__ x, const float* __restrict__ inv,
                                                            const float* ghat, int64_t n, int d, float* out) {
  const int l16 = threadIdx.x & 15;
  for (int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); r < n; r += (int64_t)gridDim.x * 16) {
    const float s = inv[r];
    float dot = 0.f;
    for (int c = l16; c < d; c += 16) dot += x[r * d + c] * s * ghat[r * d + c];
    dot = group16_sum(dot);
    for (int c = l16; c < d; c += 16) out[r * d + c] = s * (ghat[r * d + c] - x[r * d + c] * s * dot);
  }
}

FwdPlan plan_bwd_rows(int64_t mx, int64_t ny, int d, int rows_per_block, int64_t resident = 512) {
  FwdPlan p = plan_fwd(mx, ny, rows_per_block, resident);
  // every split keeps an [mx, D] fp32 partial: with many stationary rows the row blocks alone fill
  // the chip, and the partials must stay small (cap 1 GiB)
  const int64_t total_tiles = (ny + kTileJ - 1) / kTileJ;
  while (p.nsplit > 1 && (p.m_blocks >= 3 * resident || (int64_t)p.nsplit * mx * d * 4 > (1ll << 30))) {
    p.tiles_per_split *= 2;
    p.nsplit = (int)((total_tiles + p.tiles_per_split - 1) / p.tiles_per_split);
  }
  return p;
}

template <int D>
FwdPlan plan_bwd(int64_t mx, int64_t ny) {
  return plan_bwd_rows(mx, ny, D, BwdShape<D>::ROWS_PER_BLOCK);
}

// The two-f16-plane flash forward of d = 64 with long tile loops runs as 512-thread workgroups of 256 stationary rows,
// one per CU (infonce_pipe_kernel, NW = 8): measured against the 256-thread form on one box — 2048 x 1M 1.73 -> 1.60 ms,
// 100K x 100K 7.93 -> 7.73 ms; d = 32 (nothing to share: 256 lanes already stage a whole tile) 4-14 % slower.  Hence:
// d = 64 and at least 256 tiles per split.  (The backward used it too in round 2 — 100K x 100K 9.20 -> 8.94 ms — until
// its staging moved into a pre-pass: launch_bwd.)
FwdPlan plan_h2_rows8(int64_t mx, int64_t ny, int d) { return plan_bwd_rows(mx, ny, d, 256, 256); }
bool h2_eight_waves(int d, const FwdPlan& p8) { return d == 64 && p8.tiles_per_split >= 256; }

int32_t reduce_splits(const FwdPlan& p, const float* gpart, int64_t mx, int d, float* g, hipStream_t s) {
  if (p.nsplit <= 1) return GCR_OK;
  const int64_t n4 = mx * d / 4;
  const int64_t want = (n4 + 255) / 256;
  hipLaunchKernelGGL(bwd_reduce_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)), dim3(256), 0, s, gpart, p.nsplit,
                     n4, g);
  return GCR_LAUNCH_STATUS();
}

template <int D>
int32_t launch_bwd(const float* x, const float* x_scale, int64_t mx, const float* y, const float* y_scale, int64_t ny,
                   float inv_tau, const float* lse_x, const float* w_x, const float* lse_y, const float* w_y, float* g,
                   void* workspace, bool exd, bool force_f32, bool unit_rows, hipStream_t s) {
  if constexpr (D <= 128) {
    if (use_b3(D, force_f32)) {
      // d <= 64: the cross-tile pipelined loop (infonce_pipe_b3_kernel); d = 128: single-buffered (one tile with its
      // transposed copy is 54 KB of LDS)
      const bool h2 = use_h2_loop(D, inv_tau, unit_rows, force_f32);
      // (always the 256-thread form: the 512-thread form of the flash forward halves the STAGING work per MFMA, and the
      // backward stages from the pre-scaled image — two loads and two splits per lane.  At 100K x 100K the 512-thread
      // backward, 3-7 % ahead in round 2, is now behind: both sides 8.25 vs 7.44 ms, excluded diagonal 8.93 vs 8.45)
      const FwdPlan p = plan_bwd_rows(mx, ny, D, BwdB3<D>::ROWS_PER_BLOCK);
      float* gpart = p.nsplit == 1 ? g : reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + bwd_header_bytes(ny, D));
      const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
      const bool has_x = w_x != nullptr && lse_x != nullptr, has_y = w_y != nullptr && lse_y != nullptr;
      float* hw = reinterpret_cast<float*>(workspace);
      if (h2) {
        hipLaunchKernelGGL(h2_wscale_kernel, dim3(1), dim3(1024), 0, s, w_x, mx, w_y, ny, hw);
        int32_t st = GCR_LAUNCH_STATUS();
        if (st != GCR_OK) return st;
        {                                                  // the loop's image of the streamed side (h2_prestage_kernel)
          float* img = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + kBwdHeader);
          const int64_t rows = fold_rows(ny);
          const dim3 pg((unsigned)((rows * (D / 4) + 255) / 256));
          const bool foldw = !exd && has_y && !has_x;      // SIDES = 2: weights into the exponent
          const bool stats = exd || has_y || !has_x;       // SIDES != 1: the loop reads the streamed rows' statistics
          if (foldw)
            hipLaunchKernelGGL((h2_prestage_kernel<D, true>), pg, dim3(256), 0, s, y, y_scale, w_y, lse_y, ny, (const float*)hw,
                               img, true);
          else
            hipLaunchKernelGGL((h2_prestage_kernel<D, false>), pg, dim3(256), 0, s, y, y_scale, has_y ? w_y : nullptr, lse_y, ny,
                               (const float*)hw, img, stats);
          st = GCR_LAUNCH_STATUS();
          if (st != GCR_OK) return st;
          if (stats) {
            lse_y = img;
            w_y = img + rows;
          }
          y = img + 2 * rows;
          y_scale = nullptr;
        }
      }
#define GCR_BWD3(EX, SD)                                                                                                \
  if constexpr (D <= 64) {                                                                                              \
    if (h2)                                                                                                             \
      hipLaunchKernelGGL((infonce_pipe_kernel<EngH2, D, 0, EX, SD>), grid, dim3(256), 0, s, x, x_scale, mx, y, y_scale, \
                         ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y, p.nsplit, p.tiles_per_split, gpart,     \
                         (float2*)nullptr, (const float*)hw);                                                           \
    else                                                                                                                \
      hipLaunchKernelGGL((infonce_pipe_kernel<EngB3, D, 0, EX, SD>), grid, dim3(256), 0, s, x, x_scale, mx, y, y_scale, \
                         ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y, p.nsplit, p.tiles_per_split, gpart,     \
                         (float2*)nullptr, (const float*)nullptr);                                                      \
  } else if (h2)                                                                                                        \
    hipLaunchKernelGGL((infonce_pipe_kernel<EngH2, D, 0, EX, SD>), grid, dim3(256), 0, s, x, x_scale, mx, y, y_scale,   \
                       ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y, p.nsplit, p.tiles_per_split, gpart,       \
                       (float2*)nullptr, (const float*)hw);                                                             \
  else                                                                                                                  \
    hipLaunchKernelGGL((infonce_bwd_b3_kernel<D, EX, SD>), grid, dim3(256), 0, s, x, x_scale, mx, y, y_scale,    \
                       ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y, p.nsplit, p.tiles_per_split, gpart)
      if (exd) {
        GCR_BWD3(true, 0);
      } else if (has_x && !has_y) {
        GCR_BWD3(false, 1);
      } else if (has_y && !has_x) {
        GCR_BWD3(false, 2);
      } else {
        GCR_BWD3(false, 0);
      }
#undef GCR_BWD3
      int32_t st = GCR_LAUNCH_STATUS();
      if (st != GCR_OK) return st;
      return reduce_splits(p, gpart, mx, D, g, s);
    }
  }
  const FwdPlan p = plan_bwd<D>(mx, ny);
  float* gpart = p.nsplit == 1 ? g : reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + bwd_header_bytes(ny, D));
  for (int pass = 0; pass < BwdShape<D>::PASSES; ++pass) {
    if (exd)
      hipLaunchKernelGGL((infonce_bwd_kernel<D, true>), dim3((unsigned)(p.m_blocks * p.nsplit)), dim3(256), 0, s, x,
                         x_scale, mx, y, y_scale, ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y,
                         pass * BwdShape<D>::CT, p.nsplit, p.tiles_per_split, gpart);
    else
      hipLaunchKernelGGL((infonce_bwd_kernel<D>), dim3((unsigned)(p.m_blocks * p.nsplit)), dim3(256), 0, s, x, x_scale,
                         mx, y, y_scale, ny, inv_tau * kLog2e, inv_tau, lse_x, w_x, lse_y, w_y, pass * BwdShape<D>::CT,
                         p.nsplit, p.tiles_per_split, gpart);
    int32_t st = GCR_LAUNCH_STATUS();
    if (st != GCR_OK) return st;
  }
  return reduce_splits(p, gpart, mx, D, g, s);
}

int anchors_per_block_for(int d) { return d <= 128 ? 256 : 128; }

template <int D>
int32_t launch_fwd(const float* a, const float* a_scale, int64_t m, const float* b, const float* b_scale, int64_t n,
                   float inv_tau, float* lse, float* col_sum, float col_bound, void* workspace, bool exd, bool force_f32,
                   bool unit_rows, hipStream_t s) {
  float2* part = reinterpret_cast<float2*>(workspace);
  if (col_sum != nullptr) {
    hipError_t err = hipMemsetAsync(col_sum, 0, sizeof(float) * (size_t)n, s);
    if (err != hipSuccess) return gcr_hip_status(err);
  }
  FwdPlan p;
  bool launched = false;
  if constexpr (D <= 128) {
    if (use_b3(D, force_f32)) {
      p = plan_fwd(m, n, ShapeB3<D>::ANCHORS_PER_BLOCK, 512);
      const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
      const float cb2 = col_sum != nullptr ? col_bound * kLog2e : 0.f;
#define GCR_FWD_E(ENG)                                                                                                  \
  if (exd)                                                                                                              \
    hipLaunchKernelGGL((infonce_fwd_e_kernel<ENG, D, false, true>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale,   \
                       n, inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, cb2);                           \
  else if (col_sum != nullptr)                                                                                          \
    hipLaunchKernelGGL((infonce_fwd_e_kernel<ENG, D, true, false>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale,   \
                       n, inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, cb2);                           \
  else                                                                                                                  \
    hipLaunchKernelGGL((infonce_fwd_e_kernel<ENG, D, false, false>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale,  \
                       n, inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, cb2)
      if constexpr (D <= 64) {
        if (use_h2(D, inv_tau, unit_rows, force_f32)) {
          bool eight = false;
          if constexpr (D == 64) {
            // 512 anchors per workgroup, one workgroup per CU: each staged table tile feeds twice the MFMAs
            const FwdPlan p8 = plan_fwd(m, n, 512, 256);
            if (!exd && col_sum == nullptr && m >= 512 && p8.tiles_per_split >= 256) {
              p = p8;
              hipLaunchKernelGGL((infonce_fwd_e_kernel<EngH2, D, false, false, 8>), dim3((unsigned)(p.m_blocks * p.nsplit)),
                                 dim3(512), 0, s, a, a_scale, m, b, b_scale, n, inv_tau * kLog2e, p.nsplit,
                                 p.tiles_per_split, part, col_sum, cb2);
              eight = true;
            }
          }
          if (!eight) {
            GCR_FWD_E(EngH2);
          }
        } else {
          GCR_FWD_E(EngB3);
        }
      } else {
        GCR_FWD_E(EngB3);
      }
#undef GCR_FWD_E
      launched = true;
    }
  }
  if (!launched) {
    p = plan_fwd(m, n, Shape<D>::ANCHORS_PER_BLOCK, D <= 64 ? 768 : 512);
    const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
    if (exd)
      hipLaunchKernelGGL((infonce_fwd_kernel<D, false, true>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale, n,
                         inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, 0.f);
    else if (col_sum != nullptr)
      hipLaunchKernelGGL((infonce_fwd_kernel<D, true>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale, n,
                         inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, col_bound * kLog2e);
    else
      hipLaunchKernelGGL((infonce_fwd_kernel<D, false>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale, n,
                         inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, col_sum, 0.f);
  }
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  hipLaunchKernelGGL(infonce_merge_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, part, p.nsplit, m, lse);
  return GCR_LAUNCH_STATUS();
}


template <int D>
int32_t launch_fwd_o(const float* a, const float* a_scale, int64_t m, const float* b, const float* b_scale, int64_t n,
                     float inv_tau, float* lse, float* o, void* workspace, bool exd, bool unit_rows, hipStream_t s) {
  const bool h2o = use_h2_loop(D, inv_tau, unit_rows, false);
  const FwdPlan p8 = plan_h2_rows8(m, n, D);
  const bool w8 = h2o && h2_eight_waves(D, p8);
  const FwdPlan p = w8 ? p8 : plan_bwd_rows(m, n, D, BwdB3<D>::ROWS_PER_BLOCK);
  float2* part = reinterpret_cast<float2*>(workspace);
  float* opart = reinterpret_cast<float*>(part + (int64_t)p.nsplit * m);
  const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
  const float* none = nullptr;
  const float h2_out = 1.0f / (EngH2::kSY * 16384.0f);               // streamed operand x 2^8, P x 2^kPExp
  if constexpr (D <= 64) {
#define GCR_FWDO(ENG, EX, OS, NWV)                                                                                       \
  hipLaunchKernelGGL((infonce_pipe_kernel<ENG, D, 1, EX, 0, NWV>), grid, dim3(64 * NWV), 0, s, a, a_scale, m, b, b_scale, \
                     n, inv_tau * kLog2e, OS, none, none, none, none, p.nsplit, p.tiles_per_split, opart, part, none)
    if (w8) {
      if constexpr (D == 64) {
        if (exd) GCR_FWDO(EngH2, true, h2_out, 8);
        else GCR_FWDO(EngH2, false, h2_out, 8);
      }
    } else if (h2o) {
      if (exd) GCR_FWDO(EngH2, true, h2_out, 4);
      else GCR_FWDO(EngH2, false, h2_out, 4);
    } else {
      if (exd) GCR_FWDO(EngB3, true, 1.0f, 4);
      else GCR_FWDO(EngB3, false, 1.0f, 4);
    }
  } else if (h2o) {
    if (exd) GCR_FWDO(EngH2, true, h2_out, 4);
    else GCR_FWDO(EngH2, false, h2_out, 4);
#undef GCR_FWDO
  } else {
    if (exd)
      hipLaunchKernelGGL((infonce_fwdo_b3_kernel<D, true>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale, n,
                         inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, opart);
    else
      hipLaunchKernelGGL((infonce_fwdo_b3_kernel<D, false>), grid, dim3(256), 0, s, a, a_scale, m, b, b_scale, n,
                         inv_tau * kLog2e, p.nsplit, p.tiles_per_split, part, opart);
  }
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  const int64_t threads = m * (D / 4);
  hipLaunchKernelGGL(infonce_merge_o_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, part, opart,
                     p.nsplit, m, D, lse, o);
  return GCR_LAUNCH_STATUS();
}

bool dim_supported(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

// ------------------------------------------------------------------------------------------
// BCE-with-logits over the all-pairs score matrix (lightgcn.py:109-113, loss_type == "bce"):
//   scores = user_vecs @ item_emb.T;  labels one-hot at pos_i;  F.binary_cross_entropy_with_logits(scores, labels)
//   = ( sum_ij softplus(s_ij) - sum_i s_{i, pos_i} ) / (M N).
// The M x N part is the row sum of softplus (forward) and (sigmoid(s)) weighted row / column sums of the operands
// (gradients) — the InfoNCE tile engine with another epilogue; rows are NOT normalised here, so the launches run on
// three bf16 planes (d <= 64, the pipelined two-product loop, MODE 2 / 3) or the f32 MFMA (d = 128, 256, or forced).
// The positive-logit term is O(M d) and lives in the host-side op (functional.bce_softplus_rowsum's callers).
// ------------------------------------------------------------------------------------------
// rowsum[i] = ln2 * sum_s part[s][i].y ;  o[i, :] = sum_s opart[s][i, :]   (fixed order)
__global__ __launch_bounds__(256) void bce_merge_kernel(const float2* __restrict__ part, const float* __restrict__ opart,
                                                        int nsplit, int64_t m_rows, int d, float* __restrict__ rowsum,
                                                        float* __restrict__ o) {
  const int d4 = o != nullptr ? d / 4 : 1;
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= m_rows * d4) return;
  const int64_t i = k / d4;
  const int c = (int)(k % d4);
  if (c == 0) {
    float l = 0.f;
    for (int s = 0; s < nsplit; ++s) l += part[(int64_t)s * m_rows + i].y;
    rowsum[i] = l * kLn2;
  }
  if (o != nullptr) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nsplit; ++s) {
      const float4 v = reinterpret_cast<const float4*>(opart)[((int64_t)s * m_rows + i) * d4 + c];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(o)[i * d4 + c] = acc;
  }
}

bool bce_pipe(int d, bool force_f32) { return d <= 64 && use_b3(d, force_f32); }

// Pre-pass of the two-f16-plane BCE loops (EngH2, UNS): per row  inv = 1 / max(|x|, 1e-12)  (the operand scale that makes it a
// unit row),  nrm = max(|x|, 1e-12)  (what un-scales its scores) and  f = nrm * w  (the row's factor in the second product:
// sum_j sigmoid(s_ij) w_j y_j = sum_j (sigmoid(s_ij) nrm_j w_j) yhat_j).  D / 4 lanes per row, one float4 per lane.
template <int D>
__global__ __launch_bounds__(256) void bce_row_scales_kernel(const float* __restrict__ x, int64_t rows, const float* __restrict__ w,
                                                             float* __restrict__ inv, float* __restrict__ nrm, float* __restrict__ f) {
  constexpr int LPR = D / 4, GROUPS = 256 / LPR;
  const int gl = threadIdx.x % LPR;
  for (int64_t r = (int64_t)blockIdx.x * GROUPS + threadIdx.x / LPR; r < rows; r += (int64_t)gridDim.x * GROUPS) {
    const float4 v = *reinterpret_cast<const float4*>(x + r * D + 4 * gl);
    float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (gl == 0) {
      const float n = fmaxf(sqrtf(ss), 1.0e-12f);
      inv[r] = 1.0f / n;
      nrm[r] = n;
      if (f != nullptr) f[r] = n * (w != nullptr ? w[r] : 1.0f);
    }
  }
}

template <int D>
void launch_bce_row_scales(const float* x, int64_t rows, const float* w, float* inv, float* nrm, float* f, hipStream_t s) {
  const int64_t want = (rows + 4 * (1024 / D) - 1) / (4 * (1024 / D));
  hipLaunchKernelGGL((bce_row_scales_kernel<D>), dim3((unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want))), dim3(256), 0, s, x,
                     rows, w, inv, nrm, f);
}

inline int64_t align256_i64(int64_t x) { return (x + 255) & ~(int64_t)255; }
// scratch of the two-plane path behind the partials: inv / nrm of both sides, f of the streamed side, hw[2]
inline int64_t bce_scales_bytes(int64_t m, int64_t n) { return align256_i64((2 * m + 3 * n + 8) * (int64_t)sizeof(float)); }

FwdPlan plan_bce_fwd(int64_t m, int64_t n, int d, bool force_f32) {
  if (bce_pipe(d, force_f32)) return plan_bwd_rows(m, n, d, 128);
  return plan_fwd(m, n, d <= 128 ? 256 : 128, d <= 64 ? 768 : 512);
}

template <int D>
int32_t launch_bce_fwd(const float* a, int64_t m, const float* b, int64_t n, float* rowsum, float* o, void* workspace,
                       int64_t core_bytes, bool force_f32, bool two_planes, hipStream_t s) {
  const FwdPlan p = plan_bce_fwd(m, n, D, force_f32);
  float2* part = reinterpret_cast<float2*>(workspace);
  float* opart = reinterpret_cast<float*>(part + (int64_t)p.nsplit * m);
  const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
  const float* none = nullptr;
  bool with_o = false;
  if constexpr (D <= 64) {
    if (bce_pipe(D, force_f32) && two_planes) {
      // two f16 planes: both sides scaled to unit rows, scores un-scaled by the norms, o's streamed factor = the norm
      float* sc = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + core_bytes);
      float *inv_a = sc, *nrm_a = sc + m, *inv_b = sc + 2 * m, *nrm_b = inv_b + n, *f_b = nrm_b + n, *hw = f_b + n;
      launch_bce_row_scales<D>(a, m, nullptr, inv_a, nrm_a, nullptr, s);
      launch_bce_row_scales<D>(b, n, nullptr, inv_b, nrm_b, o != nullptr ? f_b : nullptr, s);
      if (o != nullptr) {
        hipLaunchKernelGGL(h2_wscale_kernel, dim3(1), dim3(1024), 0, s, none, (int64_t)0, (const float*)f_b, n, hw);
        hipLaunchKernelGGL((infonce_pipe_kernel<EngH2, D, 2, false, 0, 4>), grid, dim3(256), 0, s, a, (const float*)inv_a, m, b,
                           (const float*)inv_b, n, kLog2e, 1.0f, (const float*)nrm_a, none, (const float*)nrm_b,
                           (const float*)f_b, p.nsplit, p.tiles_per_split, opart, part, (const float*)hw);
      } else {
        hipLaunchKernelGGL((infonce_pipe_kernel<EngH2, D, 2, false, 1, 4>), grid, dim3(256), 0, s, a, (const float*)inv_a, m, b,
                           (const float*)inv_b, n, kLog2e, 1.0f, (const float*)nrm_a, none, (const float*)nrm_b, none,
                           p.nsplit, p.tiles_per_split, opart, part, none);
      }
      with_o = true;
    } else if (bce_pipe(D, force_f32)) {
      if (o != nullptr)
        hipLaunchKernelGGL((infonce_pipe_kernel<EngB3, D, 2, false, 0, 4>), grid, dim3(256), 0, s, a, none, m, b, none, n,
                           kLog2e, 1.0f, none, none, none, none, p.nsplit, p.tiles_per_split, opart, part, none);
      else                                                  // row sums only: the same loop without its second product
        hipLaunchKernelGGL((infonce_pipe_kernel<EngB3, D, 2, false, 1, 4>), grid, dim3(256), 0, s, a, none, m, b, none, n,
                           kLog2e, 1.0f, none, none, none, none, p.nsplit, p.tiles_per_split, opart, part, none);
      with_o = true;
    }
  }
  if (!with_o) {
    if (o != nullptr) return GCR_EUNSUPPORTED;
    hipLaunchKernelGGL((infonce_fwd_kernel<D, false, false, true>), grid, dim3(256), 0, s, a, none, m, b, none, n, kLog2e,
                       p.nsplit, p.tiles_per_split, part, (float*)nullptr, 0.f);
  }
  int32_t st = GCR_LAUNCH_STATUS();
  if (st != GCR_OK) return st;
  const int64_t threads = m * (o != nullptr ? D / 4 : 1);
  hipLaunchKernelGGL(bce_merge_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, part, opart, p.nsplit, m, D,
                     rowsum, o);
  return GCR_LAUNCH_STATUS();
}

template <int D>
int32_t launch_bce_bwd(const float* x, int64_t mx, const float* y, int64_t ny, const float* w_x, const float* w_y, float* g,
                       void* workspace, int64_t core_bytes, bool force_f32, bool two_planes, hipStream_t s) {
  const float* none = nullptr;
  if constexpr (D <= 64) {
    if (bce_pipe(D, force_f32)) {
      const FwdPlan p = plan_bwd_rows(mx, ny, D, BwdB3<D>::ROWS_PER_BLOCK);
      float* gpart = p.nsplit == 1 ? g : reinterpret_cast<float*>(workspace);
      const dim3 grid((unsigned)(p.m_blocks * p.nsplit));
      if (two_planes && w_y != nullptr) {                  // (weights on the stationary rows stay on three planes)
        float* sc = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + core_bytes);
        float *inv_x = sc, *nrm_x = sc + mx, *inv_y = sc + 2 * mx, *nrm_y = inv_y + ny, *f_y = nrm_y + ny, *hw = f_y + ny;
        launch_bce_row_scales<D>(x, mx, nullptr, inv_x, nrm_x, nullptr, s);
        launch_bce_row_scales<D>(y, ny, w_y, inv_y, nrm_y, f_y, s);
        hipLaunchKernelGGL(h2_wscale_kernel, dim3(1), dim3(1024), 0, s, none, (int64_t)0, (const float*)f_y, ny, hw);
        hipLaunchKernelGGL((infonce_pipe_kernel<EngH2, D, 3, false, 2, 4>), grid, dim3(256), 0, s, x, (const float*)inv_x, mx, y,
                           (const float*)inv_y, ny, kLog2e, 1.0f, (const float*)nrm_x, none, (const float*)nrm_y,
                           (const float*)f_y, p.nsplit, p.tiles_per_split, gpart, (float2*)nullptr, (const float*)hw);
        int32_t st = GCR_LAUNCH_STATUS();
        if (st != GCR_OK) return st;
        return reduce_splits(p, gpart, mx, D, g, s);
      }
      if (w_x != nullptr)
        hipLaunchKernelGGL((infonce_pipe_kernel<EngB3, D, 3, false, 1, 4>), grid, dim3(256), 0, s, x, none, mx, y, none, ny,
                           kLog2e, 1.0f, none, w_x, none, none, p.nsplit, p.tiles_per_split, gpart, (float2*)nullptr, none);
      else
        hipLaunchKernelGGL((infonce_pipe_kernel<EngB3, D, 3, false, 2, 4>), grid, dim3(256), 0, s, x, none, mx, y, none, ny,
                           kLog2e, 1.0f, none, none, none, w_y, p.nsplit, p.tiles_per_split, gpart, (float2*)nullptr, none);
      int32_t st = GCR_LAUNCH_STATUS();
      if (st != GCR_OK) return st;
      return reduce_splits(p, gpart, mx, D, g, s);
    }
  }
  const FwdPlan p = plan_bwd<D>(mx, ny);
  float* gpart = p.nsplit == 1 ? g : reinterpret_cast<float*>(workspace);
  for (int pass = 0; pass < BwdShape<D>::PASSES; ++pass) {
    hipLaunchKernelGGL((infonce_bwd_kernel<D, false, true>), dim3((unsigned)(p.m_blocks * p.nsplit)), dim3(256), 0, s, x,
                       none, mx, y, none, ny, kLog2e, 1.0f, none, w_x, none, w_y, pass * BwdShape<D>::CT, p.nsplit,
                       p.tiles_per_split, gpart);
    int32_t st = GCR_LAUNCH_STATUS();
    if (st != GCR_OK) return st;
  }
  return reduce_splits(p, gpart, mx, D, g, s);
}

// ------------------------------------------------------------------------------------------
// Nearest-centroid assignment for the NCL prototype step (ncl.py:340-356: faiss.Kmeans.train +
// index.search(x, 1)): argmin_k ||x_i - c_k||^2 = argmax_k (<x_i, c_k> - 0.5 ||c_k||^2).  Same MFMA
// tile engine as the InfoNCE forward (points stationary on the lanes, centroids streamed through
// LDS); the epilogue is an in-lane running arg-max instead of an exp2-sum.  Ties go to the
// smaller centroid id.
// ------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? 3 : 2)) void kmeans_assign_kernel(
    const float* __restrict__ x, int64_t n, const float* __restrict__ cent, const float* __restrict__ half_sq,
    int64_t k, int64_t* __restrict__ assign, float* __restrict__ best_out) {
  using S = Shape<D>;
  __shared__ __align__(16) float lds[2][kTileJ * S::STRIDE];
  __shared__ __align__(16) float st_bias[2][kTileJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * S::NT);
  float bfrag[S::NT][S::KH];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) load_stationary<D>(x, nullptr, n, i0 + 32 * t + i32, h, 1.0f, bfrag[t]);
  float best[S::NT];
  int bidx[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    best[t] = -INFINITY;
    bidx[t] = 0x7fffffff;
  }
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  float4 regs[S::NLD];
  float bias = 0.f;
  auto load_bias = [&](int64_t j0) {
    if (tid < kTileJ) bias = (j0 + tid < k) ? -half_sq[j0 + tid] : -INFINITY;  // rows past k never win
  };
  stage_load<D>(cent, nullptr, k, 0, tid, regs);
  load_bias(0);
  stage_store<D>(lds[0], tid, regs);
  if (tid < kTileJ) st_bias[0][tid] = bias;
  __syncthreads();
  for (int64_t tt = 0; tt < tiles; ++tt) {
    const int cur = (int)(tt & 1);
    const int64_t nxt = tt + 1 < tiles ? tt + 1 : tt;
    stage_load<D>(cent, nullptr, k, nxt * kTileJ, tid, regs);
    load_bias(nxt * kTileJ);
    f32x16 acc[S::NT];
    score_tile<D>(lds[cur], i32, h, bfrag, acc);
    const int j0 = (int)(tt * kTileJ);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *reinterpret_cast<const float4*>(&st_bias[cur][8 * g + 4 * h]);
      const float be[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        const int j = j0 + acc_row(r, h);
#pragma unroll
        for (int t = 0; t < S::NT; ++t) {
          const float v = acc[t][r] + be[e];
          const bool better = v > best[t] || (v == best[t] && j < bidx[t]);
          best[t] = better ? v : best[t];
          bidx[t] = better ? j : bidx[t];
        }
      }
    }
    stage_store<D>(lds[cur ^ 1], tid, regs);
    if (tid < kTileJ) st_bias[cur ^ 1][tid] = bias;
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const float v_o = __shfl_xor(best[t], 32, 64);
    const int j_o = __shfl_xor(bidx[t], 32, 64);
    const bool other = v_o > best[t] || (v_o == best[t] && j_o < bidx[t]);
    const float v = other ? v_o : best[t];
    const int j = other ? j_o : bidx[t];
    const int64_t row = i0 + 32 * t + i32;
    if (h == 0 && row < n) {
      assign[row] = j;
      if (best_out != nullptr) best_out[row] = v;
    }
  }
}

// The same on the split-operand engine (d <= 128), software-pipelined by hand like the InfoNCE forward.
// -0.5 ||c||^2 rides in as the MFMA C operand of the tile's first product; the running arg-max costs
// three VALU instructions per score (compare, select the register number, max) plus one tile-number
// select per tile: a lane walks its rows in increasing centroid id, so a strict `>` keeps the smallest
// id among equal scores; the two lane halves are merged with an explicit id comparison at the end.
template <int D>
__global__ __launch_bounds__(256, 2) void kmeans_assign_b3_kernel(
    const float* __restrict__ x, int64_t n, const float* __restrict__ cent, const float* __restrict__ half_sq,
    int64_t k, int64_t* __restrict__ assign, float* __restrict__ best_out, float* __restrict__ acc_sums = nullptr,
    float* __restrict__ acc_counts = nullptr, int n_copies = 1) {
  using S = ShapeB3<D>;
  constexpr int NS = 6 * S::KC * S::NT, NU = 16 * S::NT + S::NLD;
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
  __shared__ __align__(16) unsigned char lds[2][3 * S::PLANE];
  __shared__ __align__(16) float st_bias[2][kTileJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * (32 * S::NT);
  u32x4 bq[S::NT][3][S::KC];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) load_stationary_b3<D>(x, nullptr, n, i0 + 32 * t + i32, h, 1.0f, bq[t]);
  float best[S::NT];
  int btile[S::NT], breg[S::NT];
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    best[t] = -INFINITY;
    btile[t] = 0;
    breg[t] = 0;
  }
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  const int64_t last = tiles - 1;
  // One centroid tile's share of a lane between its global loads and its LDS stores.  The loads are LOADS ONLY — the rows
  // as they are (rows past k repeat the last centroid: their bias is -inf, they never win), ½|c|² as it is, from every
  // lane without a branch — and a tile is loaded a whole step before it is staged.  (The bias used to be negated and
  // selected on the spot inside `if (tid < 32)`, the rows multiplied by their 0/1 scale on the spot: every tile opened with
  // `global_load ...; s_waitcnt vmcnt(0)` in wave 0, a memory round trip that the other three waves sat out at the
  // barrier — half of the 1.9 us a tile took.)
  struct Staged {
    float4 v[S::NLD];
    float hb;
    bool live;
  };
  Staged ga, gb;
  auto load_tile = [&](int64_t t, Staged& g) {
    const int64_t j0 = min(t, last) * kTileJ;
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx / (D / 4), c4 = idx % (D / 4);
      const int64_t jj = min(j0 + row, k - 1);
      g.v[u] = *reinterpret_cast<const float4*>(cent + jj * D + 4 * c4);
    }
    const int64_t jb = j0 + (tid & (kTileJ - 1));
    g.hb = half_sq[min(jb, k - 1)];
    g.live = jb < k;
  };
  auto store_tile = [&](unsigned char* out, int sbuf, const Staged& g) {
#pragma unroll
    for (int u = 0; u < S::NLD; ++u) stage_store_b3_one<D>(out, tid, g.v[u], u);
    if (tid < kTileJ) st_bias[sbuf][tid] = g.live ? -g.hb : -INFINITY;          // rows past k never win
  };
  // one tile: scores of the tile in lds[buf] into `nxt` (C = bias), optionally interleaved with the
  // arg-max over `cur` (the previous tile, number tt) and the staging of the tile held in `st`
  auto tile = [&](auto with_cur, const f32x16 (&cur)[S::NT], f32x16 (&nxt)[S::NT], int64_t tt, int buf, const Staged& st) {
    constexpr bool CUR = decltype(with_cur)::value;
    const unsigned char* base = lds[buf] + i32 * S::ROWB + h * (S::KH * 2);
    unsigned char* out = lds[buf ^ 1];
    float binit[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bb = *reinterpret_cast<const float4*>(&st_bias[buf][8 * g + 4 * h]);
      binit[4 * g + 0] = bb.x; binit[4 * g + 1] = bb.y; binit[4 * g + 2] = bb.z; binit[4 * g + 3] = bb.w;
    }
    float bprev[S::NT];
#pragma unroll
    for (int t = 0; t < S::NT; ++t) bprev[t] = best[t];
    auto micro = [&](int m) {
      if (m < 16 * S::NT) {
        const int r = m / S::NT, t = m % S::NT;
        const float v = cur[t][r];
        breg[t] = v > best[t] ? r : breg[t];
        best[t] = fmaxf(best[t], v);
      } else {
        const int u = m - 16 * S::NT;
        stage_store_b3_one<D>(out, tid, st.v[u], u);
        if (u == 0 && tid < kTileJ) st_bias[buf ^ 1][tid] = st.live ? -st.hb : -INFINITY;
      }
    };
    u32x4 ap[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) ap[0][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE);
#pragma unroll
    for (int c = 0; c < S::KC; ++c) {
      if (c + 1 < S::KC) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          ap[(c + 1) & 1][pl] = *reinterpret_cast<const u32x4*>(base + pl * S::PLANE + 16 * (c + 1));
      }
#pragma unroll
      for (int term = 0; term < 6; ++term) {
#pragma unroll
        for (int t = 0; t < S::NT; ++t) {
          const int slot = (c * 6 + term) * S::NT + t;
          f32x16 cin = nxt[t];
          if (c == 0 && term == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cin[r] = binit[r];
          }
          nxt[t] = mfma_bf16(ap[c & 1][TA[term]], bq[t][TB[term]][c], cin);
          if (CUR) {
#pragma unroll
            for (int u = slot * NU / NS; u < (slot + 1) * NU / NS; ++u) micro(u);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (CUR) {
#pragma unroll
      for (int t = 0; t < S::NT; ++t) {
        btile[t] = best[t] > bprev[t] ? (int)tt : btile[t];
        asm volatile("" : "+v"(best[t]), "+v"(breg[t]), "+v"(btile[t]));
      }
    }
  };
  auto argmax_only = [&](const f32x16 (&cur)[S::NT], int64_t tt) {
#pragma unroll
    for (int t = 0; t < S::NT; ++t) {
      const float bp = best[t];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        breg[t] = cur[t][r] > best[t] ? r : breg[t];
        best[t] = fmaxf(best[t], cur[t][r]);
      }
      btile[t] = best[t] > bp ? (int)tt : btile[t];
    }
  };
  f32x16 acc_a[S::NT], acc_b[S::NT];
  load_tile(0, ga);
  load_tile(1, gb);
  store_tile(lds[0], 0, ga);
  load_tile(2, ga);
  __syncthreads();
  tile(std::false_type{}, acc_b, acc_a, 0, 0, gb);      // scores of tile 0; nothing to reduce yet
  store_tile(lds[1], 1, gb);
  __syncthreads();
  int64_t tt = 0;
  for (; tt + 2 <= last; tt += 2) {                     // step tt stages tile tt + 2 (loaded a step ago), loads tile tt + 3
    load_tile(tt + 3, gb);
    tile(std::true_type{}, acc_a, acc_b, tt, 1, ga);
    __syncthreads();
    load_tile(tt + 4, ga);
    tile(std::true_type{}, acc_b, acc_a, tt + 1, 0, gb);
    __syncthreads();
  }
  if (tt < last) {
    tile(std::true_type{}, acc_a, acc_b, tt, 1, ga);
    __syncthreads();
    argmax_only(acc_b, last);
  } else {
    argmax_only(acc_a, last);
  }
#pragma unroll
  for (int t = 0; t < S::NT; ++t) {
    const int jl = best[t] > -INFINITY ? btile[t] * kTileJ + (breg[t] & 3) + 8 * (breg[t] >> 2) + 4 * h : 0x7fffffff;
    const float v_o = __shfl_xor(best[t], 32, 64);
    const int j_o = __shfl_xor(jl, 32, 64);
    const bool other = v_o > best[t] || (v_o == best[t] && j_o < jl);
    const float v = other ? v_o : best[t];
    const int j = other ? j_o : jl;
    const int64_t row = i0 + 32 * t + i32;
    if (h == 0 && row < n) {
      if (assign != nullptr) assign[row] = j;
      if (best_out != nullptr) best_out[row] = v;
    }
    btile[t] = j;                                         // kept for the fused centroid update below
  }
  // Lloyd update fused into the search (gcr_kmeans_assign_accumulate_f32): every point's row is added to its cluster's
  // sum straight away — 256-B float-atomic rows into private copy blockIdx % n_copies — instead of a second kernel that
  // re-reads the assignment.  The rows are re-read from L2 (the registers hold them as bf16 planes); 8 loads in flight.
  if (acc_sums != nullptr) {
    float* __restrict__ sb = acc_sums + (int64_t)(blockIdx.x % n_copies) * k * D;
    float* __restrict__ cb = acc_counts + (int64_t)(blockIdx.x % n_copies) * k;
    constexpr int NVX = (D + 63) / 64;
#pragma unroll
    for (int t = 0; t < S::NT; ++t) {
      // all rows of the tile in flight at once (d <= 64: 32 loads; a batch of 8 per round trip left the wave waiting on
      // eight dependent L2 round trips per tile: ~20 of the iteration's 53 us), then the row atomics
      constexpr int PB = NVX == 1 ? 32 : 8;
      for (int p0 = 0; p0 < 32; p0 += PB) {
        float xv[PB][NVX];
        int cj[PB];
#pragma unroll
        for (int q = 0; q < PB; ++q) {
          const int64_t row = i0 + 32 * t + p0 + q;
          cj[q] = __builtin_amdgcn_readlane(btile[t], p0 + q);
          if (row >= n || cj[q] < 0 || cj[q] >= k) cj[q] = -1;
          const float* xp = x + (row < n ? row : 0) * D;
#pragma unroll
          for (int c = 0; c < NVX; ++c) xv[q][c] = (lane + 64 * c < D) ? xp[lane + 64 * c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < PB; ++q) {
          if (cj[q] < 0) continue;                        // wave-uniform
#pragma unroll
          for (int c = 0; c < NVX; ++c)
            if (lane + 64 * c < D) atomicAdd(sb + (int64_t)cj[q] * D + lane + 64 * c, xv[q][c]);
          if (lane == 0) atomicAdd(cb + cj[q], 1.0f);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The e_step's search at NCL's sizes (76.8 K sampled points x 300 centroids, 25 times per table and step) is a LATENCY
// problem: the launch above puts ~1 wave on every SIMD, and each of its ten centroid tiles costs a wave the tile's
// staging (split + LDS stores), a workgroup barrier and the first operand reads, none of it hidden — 19 us of search
// for 7 us of matrix work.  Here the centroids are split into their three bf16 planes ONCE per Lloyd iteration, by
// kmeans_image_kernel, into an image in MFMA-fragment order: for tile T, k-chunk c, plane p the 64 lanes' operand
// registers are 1 KB of consecutive memory — one coalesced global_load_dwordx4 per fragment, served by L2 (120 KB for
// k = 300).  A wave then needs no LDS, no barrier and no partner: 32 points stationary in registers (NT = 1: twice the
// waves of the tiled kernel, so that the 1024 SIMDs hold 2-3 each), the next tile's twelve fragments in flight while
// this tile's 24 MFMAs run, the running arg-max of the previous tile in their shadow.
// ------------------------------------------------------------------------------------------
// image[((T * KC + c) * 3 + p) * 64 + lane] = the 8 bf16 of plane p, features [h * KH + 8 c, + 8) of centroid 32 T + i32
// (lane = 32 h + i32); bias[32 T + r] = -0.5 |c|^2, -inf past k (such a row never wins).
template <int D>
__global__ __launch_bounds__(256) void kmeans_image_kernel(const float* __restrict__ cent, const float* __restrict__ half_sq,
                                                           int64_t k, u32x4* __restrict__ image, float* __restrict__ bias) {
  using S = ShapeB3<D>;
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread per (tile, chunk, lane)
  if (idx < tiles * kTileJ) bias[idx] = idx < k ? -half_sq[idx] : -INFINITY;
  if (idx >= tiles * S::KC * 64) return;
  const int lane = (int)(idx & 63);
  const int c = (int)((idx >> 6) % S::KC);
  const int64_t T = (idx >> 6) / S::KC;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t row = T * kTileJ + i32;
  float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
  if (row < k) {
    const float* p = cent + row * D + h * S::KH + 8 * c;
    v0 = *reinterpret_cast<const float4*>(p);
    v1 = *reinterpret_cast<const float4*>(p + 4);
  }
  unsigned q[3][4];
  split3(v0.x, v0.y, q[0][0], q[1][0], q[2][0]);
  split3(v0.z, v0.w, q[0][1], q[1][1], q[2][1]);
  split3(v1.x, v1.y, q[0][2], q[1][2], q[2][2]);
  split3(v1.z, v1.w, q[0][3], q[1][3], q[2][3]);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl)
    image[((T * S::KC + c) * 3 + pl) * 64 + lane] = (u32x4){q[pl][0], q[pl][1], q[pl][2], q[pl][3]};
}

// LOWREG: the same search at <= 128 registers per lane (four waves per SIMD): fragments are fetched one k-chunk ahead
// (three 16-B loads in flight per lane instead of two tiles' worth), one accumulator, the arg-max right behind its tile —
// each wave now stalls on L2 latency, and the other three waves of the SIMD cover it.  What it buys is CO-RESIDENCE: a
// wave of this kernel fits into the registers the InfoNCE loops of the same training step leave free on a SIMD (2 x 176-192
// of 512), so the e_step's chain of small launches no longer waits for those grids to drain.
// INCR (gcr_kmeans_search_image_incr_f32): the cluster sums are kept in 64-bit FIXED POINT across the Lloyd iterations and
// only the points whose assignment CHANGED touch them — the row leaves its old cluster's sum and joins the new one's.
// Integer adds are exact and commute, so the incremental sums equal a fresh accumulation bit for bit, in any order: no
// drift over the 25 iterations, and the e_step becomes run-to-run reproducible (float row atomics are not).  After the first
// iterations a few per cent of the points move, and the accumulation — 15-19 of an iteration's ~46 us as float row atomics
// at the memory-side unit's rate — all but disappears (3-4 full passes' worth over 25 iterations instead of 25).
// q = round(x * qscale[0]), qscale[0] a power of two with |q| < 2^30 (set from max |x| by the caller).
template <int D, bool LOWREG = false, bool INCR = false>
__global__ __launch_bounds__(256, LOWREG ? 4 : 2) void kmeans_search_img_kernel(
    const float* __restrict__ x, int64_t n, const u32x4* __restrict__ image, const float* __restrict__ bias, int64_t k,
    int64_t* __restrict__ assign, float* __restrict__ acc_sums, float* __restrict__ acc_counts, int n_copies,
    int32_t* __restrict__ prev_assign = nullptr, const float* __restrict__ qscale = nullptr,
    long long* __restrict__ sums_q = nullptr, int32_t* __restrict__ counts_i = nullptr) {
  using S = ShapeB3<D>;
  constexpr int KC = S::KC, NF = 3 * KC;                   // fragments per centroid tile
  constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // (streamed plane, stationary plane), smallest terms first
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i32 = lane & 31, h = lane >> 5;
  const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (i0 >= n) return;                                      // (whole waves only: nothing below synchronises across waves)
  u32x4 bq[3][KC];
  load_stationary_b3<D>(x, nullptr, n, i0 + i32, h, 1.0f, bq);
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  float best = -INFINITY;
  int btile = 0, breg = 0;
  u32x4 fa[NF], fb[NF];
  float4 ba[4], bb[4];
  auto load_tile = [&](int64_t t, u32x4 (&f)[NF], float4 (&b4)[4]) {
    const int64_t tt = min(t, tiles - 1);
    const u32x4* base = image + tt * (NF * 64) + lane;
#pragma unroll
    for (int q = 0; q < NF; ++q) f[q] = base[q * 64];
#pragma unroll
    for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const float4*>(bias + tt * kTileJ + 8 * g + 4 * h);
  };
  auto scores = [&](const u32x4 (&f)[NF], const float4 (&b4)[4], f32x16& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[4 * g + 0] = b4[g].x; acc[4 * g + 1] = b4[g].y; acc[4 * g + 2] = b4[g].z; acc[4 * g + 3] = b4[g].w;
    }
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int term = 0; term < 6; ++term) acc = mfma_bf16(f[c * 3 + TA[term]], bq[TB[term]][c], acc);
  };
  auto argmax = [&](const f32x16& acc, int64_t t) {
    const float bp = best;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      breg = acc[r] > best ? r : breg;
      best = fmaxf(best, acc[r]);
    }
    btile = best > bp ? (int)t : btile;                     // a lane walks its rows in increasing centroid id: strict > keeps the smallest
  };
  if constexpr (LOWREG) {
    const u32x4* fp = image + lane;
    u32x4 cur[3], nxt[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) cur[pl] = fp[pl * 64];
    for (int64_t t = 0; t < tiles; ++t) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + t * kTileJ + 8 * g + 4 * h);
        acc[4 * g + 0] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        // the next chunk's three fragments (the next tile's first ones behind the last chunk; the image is read once past
        // its end by the last tile: clamped)
        const int64_t nq = min((t * KC + c + 1) * 3, (tiles * KC - 1) * 3);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) nxt[pl] = fp[(nq + pl) * 64];
#pragma unroll
        for (int term = 0; term < 6; ++term) acc = mfma_bf16(cur[TA[term]], bq[TB[term]][c], acc);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) cur[pl] = nxt[pl];
      }
      argmax(acc, t);
    }
  } else {
  f32x16 acc_a, acc_b;
  load_tile(0, fa, ba);
  load_tile(1, fb, bb);
  scores(fa, ba, acc_a);
  int64_t t = 0;
  for (; t + 2 < tiles; t += 2) {                           // tile t's scores are in acc_a, tile t + 1's fragments in fb
    load_tile(t + 2, fa, ba);
    scores(fb, bb, acc_b);
    argmax(acc_a, t);
    load_tile(t + 3, fb, bb);
    scores(fa, ba, acc_a);
    argmax(acc_b, t + 1);
  }
  argmax(acc_a, t);
  if (t + 1 < tiles) {
    scores(fb, bb, acc_b);
    argmax(acc_b, t + 1);
  }
  }
  const int jl = best > -INFINITY ? btile * kTileJ + (breg & 3) + 8 * (breg >> 2) + 4 * h : 0x7fffffff;
  const float v_o = __shfl_xor(best, 32, 64);
  const int j_o = __shfl_xor(jl, 32, 64);
  const bool other = v_o > best || (v_o == best && j_o < jl);
  const int j = other ? j_o : jl;
  const int64_t row = i0 + i32;
  if (h == 0 && row < n && assign != nullptr) assign[row] = j;
  if constexpr (INCR) {
    const float qs = qscale[0];
    // private copy = workgroup % n_copies (the first iterations move every point: 76.8 K 512-B rows onto a few hundred
    // rows of ONE copy serialise in the memory-side atomic unit: 144 us for the first launch); integer sums of the copies
    // add up exactly, whichever copy a row joined and whichever it leaves
    sums_q += (int64_t)(blockIdx.x % n_copies) * k * D;
    counts_i += (int64_t)(blockIdx.x % n_copies) * k;
    const int old_l = (h == 0 && row < n) ? prev_assign[row] : -1;
    const bool moved_l = h == 0 && row < n && old_l != j;
    if (moved_l) prev_assign[row] = j;
    unsigned long long moved = __ballot(moved_l);           // bit q: point q of the wave changed cluster
    while (moved != 0ull) {                                 // wave-uniform walk over the moved points only
      const int q = __builtin_ctzll(moved);
      moved &= moved - 1ull;
      const int cn = __builtin_amdgcn_readlane(j, q), co = __builtin_amdgcn_readlane(old_l, q);
      const float* xp = x + (i0 + q) * D;
#pragma unroll
      for (int c = 0; c < (D + 63) / 64; ++c) {
        if (lane + 64 * c < D) {
          const long long v = __float2ll_rn(xp[lane + 64 * c] * qs);
          if (cn >= 0 && cn < k) atomicAdd(reinterpret_cast<unsigned long long*>(sums_q + (int64_t)cn * D + lane + 64 * c), (unsigned long long)v);
          if (co >= 0 && co < k) atomicAdd(reinterpret_cast<unsigned long long*>(sums_q + (int64_t)co * D + lane + 64 * c), (unsigned long long)(-v));
        }
      }
      if (lane == 0) {
        if (cn >= 0 && cn < k) atomicAdd(counts_i + cn, 1);
        if (co >= 0 && co < k) atomicAdd(counts_i + co, -1);
      }
    }
    return;
  }
  if (acc_sums == nullptr) return;
  // Lloyd update fused in, as in kmeans_assign_b3_kernel: every point's row goes to its cluster's sum as one 256-B
  // float-atomic row (private copy = workgroup % n_copies), all 32 rows of the wave loaded before the first atomic
  float* __restrict__ sb = acc_sums + (int64_t)(blockIdx.x % n_copies) * k * D;
  float* __restrict__ cb = acc_counts + (int64_t)(blockIdx.x % n_copies) * k;
  constexpr int NVX = (D + 63) / 64;
  constexpr int PB = 32;                                    // rows in flight (the point planes are dead by now)
#pragma unroll
  for (int p0 = 0; p0 < 32; p0 += PB) {
    float xv[PB][NVX];
    int cj[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int64_t r = i0 + p0 + q;
      cj[q] = __builtin_amdgcn_readlane(j, p0 + q);
      if (r >= n || cj[q] < 0 || cj[q] >= k) cj[q] = -1;
      const float* xp = x + (r < n ? r : 0) * D;
#pragma unroll
      for (int c = 0; c < NVX; ++c) xv[q][c] = (lane + 64 * c < D) ? xp[lane + 64 * c] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      if (cj[q] < 0) continue;                              // wave-uniform
#pragma unroll
      for (int c = 0; c < NVX; ++c)
        if (lane + 64 * c < D) atomicAdd(sb + (int64_t)cj[q] * D + lane + 64 * c, xv[q][c]);
      if (lane == 0) atomicAdd(cb + cj[q], 1.0f);
    }
  }
}

// (Tried and removed in round 4: the image shared by a workgroup's four waves through a four-slot LDS ring filled by LDS-DMA
// three tiles ahead — global_load_lds issued from inline assembly so that hipcc does not drain it with vmcnt(0), counted
// vmcnt + one raw s_barrier per tile, 64 points per wave: an eighth of the L2 traffic, same assignments, 29.4 us against
// 27.4 us for the search alone and 8.74 against 8.42 ms for the NCL iteration.  The three designs — LDS staging with a
// barrier per tile 32 us, per-wave L2 streaming 27 us, LDS-DMA ring 29 us — all sit at ~2 us per centroid tile for 0.75 us
// of matrix work at 2.4 GHz: what they share is 2400 wave-tiles on 1024 SIMDs (2.3 per SIMD: a makespan of 3) and the
// clock the chip holds under 1024 SIMDs of back-to-back bf16 MFMAs, not their operand path.)
}  // namespace

extern "C" int64_t gcr_kmeans_image_bytes(int64_t k, int32_t d) {
  if (k < 1 || !(d == 32 || d == 64 || d == 128)) return 0;
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  return tiles * (3 * (d / 16) * 64 * 16 + kTileJ * (int64_t)sizeof(float));      // fragments, then the bias rows
}

extern "C" int32_t gcr_kmeans_centroid_image_f32(const float* centroids, const float* half_sqnorm, int64_t k, int32_t d,
                                                 void* image, void* stream) {
  GCR_CHECK_ARG(k >= 1 && k < (1ll << 31));
  if (!(d == 32 || d == 64 || d == 128)) return GCR_EUNSUPPORTED;
  GCR_CHECK_ARG(centroids && half_sqnorm && image);
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  u32x4* img = reinterpret_cast<u32x4*>(image);
  float* bias = reinterpret_cast<float*>(img + tiles * 3 * (d / 16) * 64);
  const int64_t threads = tiles * (d / 16) * 64;           // >= tiles * 32: covers the bias rows too
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((threads + 255) / 256));
  switch (d) {
    case 32: hipLaunchKernelGGL((kmeans_image_kernel<32>), grid, dim3(256), 0, s, centroids, half_sqnorm, k, img, bias); break;
    case 64: hipLaunchKernelGGL((kmeans_image_kernel<64>), grid, dim3(256), 0, s, centroids, half_sqnorm, k, img, bias); break;
    default: hipLaunchKernelGGL((kmeans_image_kernel<128>), grid, dim3(256), 0, s, centroids, half_sqnorm, k, img, bias); break;
  }
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_search_image_f32(const float* x, int64_t n, const void* image, int64_t k, int32_t d,
                                               int64_t* assign, float* sums, float* counts, int32_t n_copies, uint32_t flags,
                                               void* stream) {
  GCR_CHECK_ARG(n >= 0 && k >= 1 && k < (1ll << 31) && (flags & ~(uint32_t)GCR_KMEANS_SEARCH_LOW_REGISTERS) == 0);
  if (!(d == 32 || d == 64)) return GCR_EUNSUPPORTED;      // (d = 128: the point planes + two tiles of fragments exceed 256 registers)
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x && image && (assign || sums));
  GCR_CHECK_ARG((sums == nullptr) == (counts == nullptr) && (sums == nullptr || (n_copies >= 1 && n_copies <= 64)));
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  const u32x4* img = reinterpret_cast<const u32x4*>(image);
  const float* bias = reinterpret_cast<const float*>(img + tiles * 3 * (d / 16) * 64);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 127) / 128));
  const bool low = (flags & GCR_KMEANS_SEARCH_LOW_REGISTERS) != 0;
  if (d == 32) {
    if (low) hipLaunchKernelGGL((kmeans_search_img_kernel<32, true>), grid, dim3(256), 0, s, x, n, img, bias, k, assign, sums, counts, (int)n_copies);
    else hipLaunchKernelGGL((kmeans_search_img_kernel<32, false>), grid, dim3(256), 0, s, x, n, img, bias, k, assign, sums, counts, (int)n_copies);
  } else {
    if (low) hipLaunchKernelGGL((kmeans_search_img_kernel<64, true>), grid, dim3(256), 0, s, x, n, img, bias, k, assign, sums, counts, (int)n_copies);
    else hipLaunchKernelGGL((kmeans_search_img_kernel<64, false>), grid, dim3(256), 0, s, x, n, img, bias, k, assign, sums, counts, (int)n_copies);
  }
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_search_image_incr_f32(const float* x, int64_t n, const void* image, int64_t k, int32_t d,
                                                    int32_t* prev_assign, const float* qscale, int64_t* sums_q,
                                                    int32_t* counts, int32_t n_copies, uint32_t flags, void* stream) {
  GCR_CHECK_ARG(n >= 0 && k >= 1 && k < (1ll << 31) && (flags & ~(uint32_t)GCR_KMEANS_SEARCH_LOW_REGISTERS) == 0);
  GCR_CHECK_ARG(n_copies >= 1 && n_copies <= 64);
  if (!(d == 32 || d == 64)) return GCR_EUNSUPPORTED;
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x && image && prev_assign && qscale && sums_q && counts);
  const int64_t tiles = (k + kTileJ - 1) / kTileJ;
  const u32x4* img = reinterpret_cast<const u32x4*>(image);
  const float* bias = reinterpret_cast<const float*>(img + tiles * 3 * (d / 16) * 64);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 127) / 128));
  const bool low = (flags & GCR_KMEANS_SEARCH_LOW_REGISTERS) != 0;
  long long* sq = reinterpret_cast<long long*>(sums_q);

#define GCR_KMI(DD, LOW)                                                                                                \
  hipLaunchKernelGGL((kmeans_search_img_kernel<DD, LOW, true>), grid, dim3(256), 0, s, x, n, img, bias, k, (int64_t*)nullptr, \
                     (float*)nullptr, (float*)nullptr, (int)n_copies, prev_assign, qscale, sq, counts)
  if (d == 32) {
    if (low) GCR_KMI(32, true); else GCR_KMI(32, false);
  } else {
    if (low) GCR_KMI(64, true); else GCR_KMI(64, false);
  }
#undef GCR_KMI
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_assign_f32(const float* x, int64_t n, const float* centroids, const float* half_sqnorm,
                                         int64_t k, int32_t d, int64_t* assign, float* best_score, void* stream) {
  GCR_CHECK_ARG(n >= 0 && k >= 1 && k < (1ll << 31));
  if (!dim_supported(d)) return GCR_EUNSUPPORTED;
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x && centroids && half_sqnorm && assign);
  hipStream_t s = (hipStream_t)stream;
#define GCR_KM(DD)                                                                                             \
  hipLaunchKernelGGL((kmeans_assign_kernel<DD>), dim3((unsigned)((n + Shape<DD>::ANCHORS_PER_BLOCK - 1) /      \
                                                                 Shape<DD>::ANCHORS_PER_BLOCK)),               \
                     dim3(256), 0, s, x, n, centroids, half_sqnorm, k, assign, best_score)
#define GCR_KM3(DD)                                                                                            \
  hipLaunchKernelGGL((kmeans_assign_b3_kernel<DD>), dim3((unsigned)((n + ShapeB3<DD>::ANCHORS_PER_BLOCK - 1) / \
                                                                    ShapeB3<DD>::ANCHORS_PER_BLOCK)),          \
                     dim3(256), 0, s, x, n, centroids, half_sqnorm, k, assign, best_score)
  const bool b3 = use_b3(d);
  switch (d) {
    case 32: if (b3) GCR_KM3(32); else GCR_KM(32); break;
    case 64: if (b3) GCR_KM3(64); else GCR_KM(64); break;
    case 128: if (b3) GCR_KM3(128); else GCR_KM(128); break;
    default: GCR_KM(256); break;
  }
#undef GCR_KM
#undef GCR_KM3
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_kmeans_assign_accumulate_f32(const float* x, int64_t n, const float* centroids,
                                                    const float* half_sqnorm, int64_t k, int32_t d, int64_t* assign,
                                                    float* sums, float* counts, int32_t n_copies, void* stream) {
  GCR_CHECK_ARG(n >= 0 && k >= 1 && k < (1ll << 31) && n_copies >= 1 && n_copies <= 64);
  if (!dim_supported(d) || !use_b3(d)) return GCR_EUNSUPPORTED;      // split-operand engine only (d <= 128)
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x && centroids && half_sqnorm && sums && counts);
  hipStream_t s = (hipStream_t)stream;
#define GCR_KMA(DD)                                                                                            \
  hipLaunchKernelGGL((kmeans_assign_b3_kernel<DD>), dim3((unsigned)((n + ShapeB3<DD>::ANCHORS_PER_BLOCK - 1) / \
                                                                    ShapeB3<DD>::ANCHORS_PER_BLOCK)),          \
                     dim3(256), 0, s, x, n, centroids, half_sqnorm, k, assign, (float*)nullptr, sums, counts, (int)n_copies)
  switch (d) {
    case 32: GCR_KMA(32); break;
    case 64: GCR_KMA(64); break;
    default: GCR_KMA(128); break;
  }
#undef GCR_KMA
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_infonce_engine(int32_t d) { return dim_supported(d) && use_b3(d) ? 1 : 0; }

extern "C" int64_t gcr_infonce_fwd_workspace_bytes(int64_t m, int64_t n, int32_t d) {
  if (m <= 0 || n <= 0 || !dim_supported(d)) return 0;
  // large enough for either engine (the choice is made per call)
  const FwdPlan p = plan_fwd(m, n, anchors_per_block_for(d), d <= 64 ? 768 : 512);
  int64_t nsplit = p.nsplit;
  if (d <= 128) {
    const FwdPlan q = plan_fwd(m, n, d <= 64 ? 256 : 128, 512);
    if (q.nsplit > nsplit) nsplit = q.nsplit;
    if (d == 64) {                                        // the 512-anchor form of the two-plane format
      const FwdPlan q8 = plan_fwd(m, n, 512, 256);
      if (q8.nsplit > nsplit) nsplit = q8.nsplit;
    }
  }
  return nsplit * m * (int64_t)sizeof(float2);
}

extern "C" int32_t gcr_infonce_fwd_ex_f32(const float* a, const float* a_scale, int64_t m, const float* b,
                                          const float* b_scale, int64_t n, int32_t d, float inv_tau, float* lse,
                                          float* col_sum, float col_bound, void* workspace, uint32_t flags,
                                          void* stream) {
  GCR_CHECK_ARG(m >= 0 && n >= 1);
  GCR_CHECK_ARG((flags & ~(uint32_t)(GCR_INFONCE_EXCLUDE_DIAGONAL | GCR_INFONCE_ENGINE_F32 | GCR_INFONCE_UNIT_ROWS)) == 0);
  const bool exd = (flags & GCR_INFONCE_EXCLUDE_DIAGONAL) != 0;
  const bool force_f32 = (flags & GCR_INFONCE_ENGINE_F32) != 0;
  const bool unit = (flags & GCR_INFONCE_UNIT_ROWS) != 0;
  GCR_CHECK_ARG(!(exd && col_sum != nullptr));
  if (!dim_supported(d)) return GCR_EUNSUPPORTED;
  if (m == 0) return GCR_OK;
  GCR_CHECK_ARG(a != nullptr && b != nullptr && lse != nullptr && workspace != nullptr);
  GCR_CHECK_ARG(m < (1ll << 40) && n < (1ll << 40));
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 32: return launch_fwd<32>(a, a_scale, m, b, b_scale, n, inv_tau, lse, col_sum, col_bound, workspace, exd, force_f32, unit, s);
    case 64: return launch_fwd<64>(a, a_scale, m, b, b_scale, n, inv_tau, lse, col_sum, col_bound, workspace, exd, force_f32, unit, s);
    case 128: return launch_fwd<128>(a, a_scale, m, b, b_scale, n, inv_tau, lse, col_sum, col_bound, workspace, exd, force_f32, unit, s);
    default: return launch_fwd<256>(a, a_scale, m, b, b_scale, n, inv_tau, lse, col_sum, col_bound, workspace, exd, force_f32, unit, s);
  }
}

extern "C" int32_t gcr_infonce_fwd_o_supported(int32_t d, uint32_t flags) {
  return (d == 32 || d == 64 || d == 128) && use_b3(d, (flags & GCR_INFONCE_ENGINE_F32) != 0) ? 1 : 0;
}

extern "C" int64_t gcr_infonce_fwd_o_workspace_bytes(int64_t m, int64_t n, int32_t d) {
  if (m <= 0 || n <= 0 || !gcr_infonce_fwd_o_supported(d, 0)) return 0;
  const FwdPlan p = plan_bwd_rows(m, n, d, 128);
  int64_t nsplit = p.nsplit;
  if (d <= 64) {                                          // either operand format (chosen per call)
    const FwdPlan q = plan_h2_rows8(m, n, d);
    if (q.nsplit > nsplit) nsplit = q.nsplit;
  }
  return nsplit * m * ((int64_t)sizeof(float2) + (int64_t)d * (int64_t)sizeof(float));
}

extern "C" int32_t gcr_infonce_fwd_o_f32(const float* a, const float* a_scale, int64_t m, const float* b,
                                         const float* b_scale, int64_t n, int32_t d, float inv_tau, float* lse,
                                         float* o, void* workspace, uint32_t flags, void* stream) {
  GCR_CHECK_ARG(m >= 0 && n >= 1);
  GCR_CHECK_ARG((flags & ~(uint32_t)(GCR_INFONCE_EXCLUDE_DIAGONAL | GCR_INFONCE_ENGINE_F32 | GCR_INFONCE_UNIT_ROWS)) == 0);
  if (!gcr_infonce_fwd_o_supported(d, flags)) return GCR_EUNSUPPORTED;
  if (m == 0) return GCR_OK;
  GCR_CHECK_ARG(a != nullptr && b != nullptr && lse != nullptr && o != nullptr && workspace != nullptr);
  const bool exd = (flags & GCR_INFONCE_EXCLUDE_DIAGONAL) != 0;
  const bool unit = (flags & GCR_INFONCE_UNIT_ROWS) != 0;
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 32: return launch_fwd_o<32>(a, a_scale, m, b, b_scale, n, inv_tau, lse, o, workspace, exd, unit, s);
    case 64: return launch_fwd_o<64>(a, a_scale, m, b, b_scale, n, inv_tau, lse, o, workspace, exd, unit, s);
    default: return launch_fwd_o<128>(a, a_scale, m, b, b_scale, n, inv_tau, lse, o, workspace, exd, unit, s);
  }
}

extern "C" int32_t gcr_infonce_fwd_f32(const float* a, const float* a_scale, int64_t m, const float* b,
                                       const float* b_scale, int64_t n, int32_t d, float inv_tau, float* lse,
                                       float* col_sum, float col_bound, void* workspace, void* stream) {
  return gcr_infonce_fwd_ex_f32(a, a_scale, m, b, b_scale, n, d, inv_tau, lse, col_sum, col_bound, workspace, 0u, stream);
}

extern "C" int32_t gcr_row_inv_norm_f32(const float* x, int64_t n, int32_t d, float eps, float* out, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && out != nullptr);
  const int64_t want = (n + 15) / 16;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0,
                     (hipStream_t)stream, x, n, d, eps, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_pos_logit_f32(const float* a, const float* a_scale, const float* b, const float* b_scale,
                                     const int64_t* pos, int64_t m, int64_t n, int32_t d, float scale, float* out,
                                     void* stream) {
  GCR_CHECK_ARG(m >= 0 && n >= 0 && d >= 1);
  if (m == 0) return GCR_OK;
  GCR_CHECK_ARG(a != nullptr && b != nullptr && out != nullptr);
  const int64_t want = (m + 15) / 16;
  hipLaunchKernelGGL(pos_logit_kernel, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0, (hipStream_t)stream,
                     a, a_scale, b, b_scale, pos, m, n, d, scale, out);
  return GCR_LAUNCH_STATUS();
}

extern "C" int64_t gcr_infonce_bwd_workspace_bytes(int64_t mx, int64_t ny, int32_t d) {
  if (mx <= 0 || ny <= 0 || !dim_supported(d)) return 0;
  FwdPlan p;
  switch (d) {
    case 32: p = plan_bwd<32>(mx, ny); break;
    case 64: p = plan_bwd<64>(mx, ny); break;
    case 128: p = plan_bwd<128>(mx, ny); break;
    default: p = plan_bwd<256>(mx, ny); break;
  }
  int64_t nsplit = p.nsplit;
  if (d <= 128) {   // either engine (chosen per call)
    const FwdPlan q = plan_bwd_rows(mx, ny, d, 128);
    if (q.nsplit > nsplit) nsplit = q.nsplit;
  }
  return (nsplit > 1 ? nsplit * mx * d * (int64_t)sizeof(float) : 0) + bwd_header_bytes(ny, d);
}

extern "C" int32_t gcr_infonce_bwd_ex_f32(const float* x, const float* x_scale, int64_t mx, const float* y,
                                          const float* y_scale, int64_t ny, int32_t d, float inv_tau,
                                          const float* lse_x, const float* w_x, const float* lse_y, const float* w_y,
                                          float* g, void* workspace, uint32_t flags, void* stream) {
  GCR_CHECK_ARG(mx >= 0 && ny >= 1);
  GCR_CHECK_ARG((flags & ~(uint32_t)(GCR_INFONCE_EXCLUDE_DIAGONAL | GCR_INFONCE_ENGINE_F32 | GCR_INFONCE_UNIT_ROWS)) == 0);
  const bool exd = (flags & GCR_INFONCE_EXCLUDE_DIAGONAL) != 0;
  const bool force_f32 = (flags & GCR_INFONCE_ENGINE_F32) != 0;
  const bool unit = (flags & GCR_INFONCE_UNIT_ROWS) != 0;
  if (!dim_supported(d)) return GCR_EUNSUPPORTED;
  if (mx == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && y != nullptr && g != nullptr);
  GCR_CHECK_ARG((w_x == nullptr) == (lse_x == nullptr) && (w_y == nullptr) == (lse_y == nullptr));
  GCR_CHECK_ARG(workspace != nullptr || gcr_infonce_bwd_workspace_bytes(mx, ny, d) == 0);
  hipStream_t s = (hipStream_t)stream;
  switch (d) {
    case 32: return launch_bwd<32>(x, x_scale, mx, y, y_scale, ny, inv_tau, lse_x, w_x, lse_y, w_y, g, workspace, exd, force_f32, unit, s);
    case 64: return launch_bwd<64>(x, x_scale, mx, y, y_scale, ny, inv_tau, lse_x, w_x, lse_y, w_y, g, workspace, exd, force_f32, unit, s);
    case 128: return launch_bwd<128>(x, x_scale, mx, y, y_scale, ny, inv_tau, lse_x, w_x, lse_y, w_y, g, workspace, exd, force_f32, unit, s);
    default: return launch_bwd<256>(x, x_scale, mx, y, y_scale, ny, inv_tau, lse_x, w_x, lse_y, w_y, g, workspace, exd, force_f32, unit, s);
  }
}

extern "C" int32_t gcr_infonce_bwd_f32(const float* x, const float* x_scale, int64_t mx, const float* y,
                                       const float* y_scale, int64_t ny, int32_t d, float inv_tau, const float* lse_x,
                                       const float* w_x, const float* lse_y, const float* w_y, float* g,
                                       void* workspace, void* stream) {
  return gcr_infonce_bwd_ex_f32(x, x_scale, mx, y, y_scale, ny, d, inv_tau, lse_x, w_x, lse_y, w_y, g, workspace, 0u,
                                stream);
}

extern "C" int32_t gcr_infonce_pos_bwd_f32(const float* x, const float* x_scale, const float* y, const float* y_scale,
                                           const int64_t* pos, const float* coef, int64_t mx, int64_t ny, int32_t d,
                                           float inv_tau, float* gx, float* gy, void* stream) {
  GCR_CHECK_ARG(mx >= 0 && ny >= 0 && d >= 1);
  if (mx == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && y != nullptr && coef != nullptr && (gx != nullptr || gy != nullptr));
  const int64_t want = (mx + 3) / 4;
  hipLaunchKernelGGL(pos_bwd_kernel, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0, (hipStream_t)stream, x,
                     x_scale, y, y_scale, pos, coef, mx, ny, d, inv_tau, gx, gy);
  return GCR_LAUNCH_STATUS();
}

extern "C" int32_t gcr_normalize_bwd_f32(const float* x, const float* inv_norm, const float* ghat, int64_t n,
                                         int32_t d, float* out, void* stream) {
  GCR_CHECK_ARG(n >= 0 && d >= 1);
  if (n == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && inv_norm != nullptr && ghat != nullptr && out != nullptr);
  const int64_t want = (n + 15) / 16;
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3((unsigned)(want > 8192 ? 8192 : want)), dim3(256), 0,
                     (hipStream_t)stream, x, inv_norm, ghat, n, d, out);
  return GCR_LAUNCH_STATUS();
}

// ---- BCE-with-logits over all pairs (lightgcn.py:109-113) ----
extern "C" int32_t gcr_bce_fwd_o_supported(int32_t d, uint32_t flags) {
  return (d == 32 || d == 64) && bce_pipe(d, (flags & GCR_INFONCE_ENGINE_F32) != 0) ? 1 : 0;
}

static int64_t bce_fwd_core_bytes(int64_t m, int64_t n, int32_t d) {
  const FwdPlan p = plan_bce_fwd(m, n, d, false), q = plan_bce_fwd(m, n, d, true);   // either engine (chosen per call)
  const int64_t nsplit = p.nsplit > q.nsplit ? p.nsplit : q.nsplit;
  return align256_i64(nsplit * m * ((int64_t)sizeof(float2) + (d <= 64 ? (int64_t)d * (int64_t)sizeof(float) : 0)));
}

extern "C" int64_t gcr_bce_fwd_workspace_bytes(int64_t m, int64_t n, int32_t d) {
  if (m <= 0 || n <= 0 || !dim_supported(d)) return 0;
  return bce_fwd_core_bytes(m, n, d) + (d <= 64 ? bce_scales_bytes(m, n) : 0);
}

extern "C" int32_t gcr_bce_fwd_f32(const float* a, int64_t m, const float* b, int64_t n, int32_t d, float* row_softplus,
                                   float* o, void* workspace, uint32_t flags, void* stream) {
  GCR_CHECK_ARG(m >= 0 && n >= 1);
  GCR_CHECK_ARG((flags & ~(uint32_t)(GCR_INFONCE_ENGINE_F32 | GCR_BCE_TWO_PLANES)) == 0);
  if (!dim_supported(d)) return GCR_EUNSUPPORTED;
  if (o != nullptr && !gcr_bce_fwd_o_supported(d, flags)) return GCR_EUNSUPPORTED;
  if (m == 0) return GCR_OK;
  GCR_CHECK_ARG(a != nullptr && b != nullptr && row_softplus != nullptr && workspace != nullptr);
  GCR_CHECK_ARG(m < (1ll << 40) && n < (1ll << 40));
  const bool f32 = (flags & GCR_INFONCE_ENGINE_F32) != 0, two = (flags & GCR_BCE_TWO_PLANES) != 0;
  hipStream_t s = (hipStream_t)stream;
  const int64_t core = bce_fwd_core_bytes(m, n, d);
  switch (d) {
    case 32: return launch_bce_fwd<32>(a, m, b, n, row_softplus, o, workspace, core, f32, two, s);
    case 64: return launch_bce_fwd<64>(a, m, b, n, row_softplus, o, workspace, core, f32, two, s);
    case 128: return launch_bce_fwd<128>(a, m, b, n, row_softplus, o, workspace, core, f32, two, s);
    default: return launch_bce_fwd<256>(a, m, b, n, row_softplus, o, workspace, core, f32, two, s);
  }
}

static int64_t bce_bwd_core_bytes(int64_t nsplit, int64_t mx, int32_t d) {
  return nsplit > 1 ? align256_i64(nsplit * mx * d * (int64_t)sizeof(float)) : 0;
}

extern "C" int64_t gcr_bce_bwd_workspace_bytes(int64_t mx, int64_t ny, int32_t d) {
  if (mx <= 0 || ny <= 0 || !dim_supported(d)) return 0;
  FwdPlan p;
  switch (d) {
    case 32: p = plan_bwd<32>(mx, ny); break;
    case 64: p = plan_bwd<64>(mx, ny); break;
    case 128: p = plan_bwd<128>(mx, ny); break;
    default: p = plan_bwd<256>(mx, ny); break;
  }
  int64_t nsplit = p.nsplit;
  if (d <= 64) {
    const FwdPlan q = plan_bwd_rows(mx, ny, d, 128);
    if (q.nsplit > nsplit) nsplit = q.nsplit;
  }
  return bce_bwd_core_bytes(nsplit, mx, d) + (d <= 64 ? bce_scales_bytes(mx, ny) : 0);
}

extern "C" int32_t gcr_bce_bwd_f32(const float* x, int64_t mx, const float* y, int64_t ny, int32_t d, const float* w_x,
                                   const float* w_y, float* g, void* workspace, uint32_t flags, void* stream) {
  GCR_CHECK_ARG(mx >= 0 && ny >= 1);
  GCR_CHECK_ARG((flags & ~(uint32_t)(GCR_INFONCE_ENGINE_F32 | GCR_BCE_TWO_PLANES)) == 0);
  if (!dim_supported(d)) return GCR_EUNSUPPORTED;
  if (mx == 0) return GCR_OK;
  GCR_CHECK_ARG(x != nullptr && y != nullptr && g != nullptr);
  GCR_CHECK_ARG((w_x != nullptr) != (w_y != nullptr));     // the weights sit on exactly one side
  GCR_CHECK_ARG(workspace != nullptr || gcr_bce_bwd_workspace_bytes(mx, ny, d) == 0);
  const bool f32 = (flags & GCR_INFONCE_ENGINE_F32) != 0, two = (flags & GCR_BCE_TWO_PLANES) != 0;
  hipStream_t s = (hipStream_t)stream;
  // (the split partials sit in front of the two-plane scratch; their size follows the plan actually used)
  int64_t core = gcr_bce_bwd_workspace_bytes(mx, ny, d) - (d <= 64 ? bce_scales_bytes(mx, ny) : 0);
  switch (d) {
    case 32: return launch_bce_bwd<32>(x, mx, y, ny, w_x, w_y, g, workspace, core, f32, two, s);
    case 64: return launch_bce_bwd<64>(x, mx, y, ny, w_x, w_y, g, workspace, core, f32, two, s);
    case 128: return launch_bce_bwd<128>(x, mx, y, ny, w_x, w_y, g, workspace, core, f32, two, s);
    default: return launch_bce_bwd<256>(x, mx, y, ny, w_x, w_y, g, workspace, core, f32, two, s);
  }
}
