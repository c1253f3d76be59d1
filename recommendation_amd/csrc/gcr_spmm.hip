// CSR x dense SpMM for the LightGCN message pass on gfx950 (MI355X).
//
// Replaces torch.sparse.mm(A, emb) (ncl.py:419, directau.py:290, selfcf.py:479, sept.py:223,
// buir.py:317, mhcn.py:440-456) and LGConv (lightgcn.py:25), with the layer combine of
// lightgcn.py:26 / ncl.py:421 and the row normalise of sept.py:224 fused into the epilogue.
//
// Mapping (HBM-bound gather; no MFMA on purpose):
//   * one 64-lane wavefront per *partition* of <= L consecutive non-zeros (host plan, see
//     gcr_spmm_plan_*): a partition is either up to 64 whole rows or one chunk of a long row, so
//     every wave moves the same number of bytes whatever the degree skew;
//   * lane l owns feature columns l, l+64, ...: with d = 64 one neighbour row is exactly one
//     coalesced 256-B wave load whose base address is wave-uniform (SGPR base + lane*4);
//   * column ids / values of 64 non-zeros are fetched with one coalesced vector load each and
//     handed out with v_readlane (scalar), so all control flow (row boundaries, edge mask,
//     tails) is scalar and divergence-free;
//   * UNR (16 at d <= 64) gathers are issued back to back before the first FMA to keep >= 4 KiB per
//     wave in flight; at 8 waves/SIMD that is >= 128 KiB per CU;
//   * long rows: chunk partial sums go to a workspace and are added in chunk order by
//     spmm_long_rows (deterministic; no float atomics).
#include "gcr_common.h"

namespace {

// gathers in flight per wave before the first FMA: interleaved A/B on MI355X (scripts/perf_spmm_ab.py)
// 4 -> 0.647 ms, 8 -> 0.612 ms, 16 -> 0.598 ms per cfg2 layer (cfg4: 7.52 / 7.06 / 6.91 ms)
template <int NV>
constexpr int unroll_for() { return NV == 1 ? 16 : 8; }

struct Epilogue {
  float val_scale;
  float* y;
  const float* acc_in;
  float* acc_out;
  float acc_scale;
  uint32_t flags;
  float* inv_norm_out;
  float* y_raw;          // second output under ROW_L2NORM: the product before the row normalise (mhcn.py:440-442)
  const float* acc_in2;  // second addend of the combine, with its own scale (the per-layer gradient of the Horner backward)
  float acc_in2_scale;
  const uint32_t* col_bits;   // host-side dispatch only (COLMASK instantiation): never read through the struct on the device
};

// ACC2: the second-addend form is its own instantiation — read unconditionally, its pointer and scale cost the hot
// instantiation 9 SGPRs (96 -> 105), which is one wave per SIMD of occupancy and 4 % of the cfg2 layer time
template <int NV, bool D64, bool ACC2>
__device__ __forceinline__ void store_row(const Epilogue& ep, int64_t row, int d, int lane, float (&acc)[NV]) {
  float yv[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) yv[v] = acc[v] * ep.val_scale;
  if (ep.flags & GCR_SPMM_ROW_L2NORM) {
    if (ep.y_raw != nullptr) {
      const int64_t rb = row * (int64_t)d;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int c = lane + 64 * v;
        if (D64 || c < d) ep.y_raw[rb + c] = yv[v];
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) ss = fmaf(yv[v], yv[v], ss);
    ss = gcr_wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int v = 0; v < NV; ++v) yv[v] *= inv;
    if (ep.inv_norm_out != nullptr && lane == 0) ep.inv_norm_out[row] = inv;
  }
  const int64_t base = row * (int64_t)d;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c = lane + 64 * v;
    if (D64 || c < d) {
      if (ep.y != nullptr) ep.y[base + c] = yv[v];
      if (ep.acc_out != nullptr) {
        float prev = ep.acc_in != nullptr ? ep.acc_in[base + c] : 0.f;
        if (ACC2) prev = fmaf(ep.acc_in2[base + c], ep.acc_in2_scale, prev);
        ep.acc_out[base + c] = (prev + yv[v]) * ep.acc_scale;
      }
    }
  }
}

// COLMASK: `keep_bits` is a bitmap over the COLUMNS (bit c set = row c of x may be non-zero) instead of over the stored
// non-zeros: non-zeros whose column is clear are skipped before their 256-B gather — the first launch of a backward pass
// whose incoming gradient has a few thousand non-zero rows of a million (the NCL step: DESIGN 4.5) reads the CSR and
// writes its output, but gathers almost nothing.
template <int NV, bool D64, bool HAS_VAL, bool MASKED, int UNR, bool ACC2, bool COLMASK = false>
__global__ __launch_bounds__(256) void spmm_parts(const int64_t* __restrict__ desc, int64_t n_parts,
                                                  const int64_t* __restrict__ rowptr,
                                                  const int32_t* __restrict__ col,
                                                  const float* __restrict__ val,
                                                  const uint32_t* __restrict__ keep_bits,
                                                  const float* __restrict__ x, int d, Epilogue ep,
                                                  float* __restrict__ partials) {
  const int lane = threadIdx.x & 63;
  // workgroups go round-robin over the 8 XCDs, so neighbouring partitions (cheap user rows, expensive item
  // rows) are mixed on every XCD; giving each XCD one contiguous eighth was measured slower
  const int64_t part = (int64_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  if (part >= n_parts) return;
  const int64_t nnz0 = desc[4 * part + 0];
  const int64_t nnz1 = desc[4 * part + 1];
  const int64_t rowinfo = desc[4 * part + 2];
  const int64_t slot = desc[4 * part + 3];
  const int row0 = (int)(rowinfo & 0xffffffffll);
  const int nrows = (int)(rowinfo >> 32);
  const int n = (int)(nnz1 - nnz0);
  const bool whole_rows = slot < 0;

  float acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;

  // local end offsets of the (<= 64) rows of a whole-row partition, one per lane
  int ends_v = 0x7fffffff;
  int cur = 0, cur_end = 0x7fffffff;
  if (whole_rows) {
    if (lane < nrows) ends_v = (int)(rowptr[row0 + lane + 1] - nnz0);
    cur_end = gcr_readlane_i(ends_v, 0);
  }

  auto flush = [&]() {
    store_row<NV, D64, ACC2>(ep, (int64_t)row0 + cur, d, lane, acc);
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    ++cur;
    cur_end = cur < nrows ? gcr_readlane_i(ends_v, cur) : 0x7fffffff;
  };

  for (int b = 0; b < n; b += 64) {
    const int m = min(64, n - b);
    int cv = 0;
    float vv = 0.f;
    bool keep = lane < m;
    if (keep) {
      const int64_t e = nnz0 + b + lane;
      cv = col[e];
      vv = HAS_VAL ? val[e] : 1.0f;
      if (MASKED && !COLMASK) keep = (keep_bits[e >> 5] >> (e & 31)) & 1u;
      if (COLMASK) keep = (keep_bits[cv >> 5] >> (cv & 31)) & 1u;
    }
    unsigned long long todo = (MASKED || COLMASK) ? __ballot(keep) : (m == 64 ? ~0ull : ((1ull << m) - 1ull));
    int cnt = __builtin_popcountll(todo);

    while (cnt >= UNR) {
      int js[UNR];
      float xr[UNR][NV];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        js[u] = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const float* xp = x + (int64_t)gcr_readlane_i(cv, js[u]) * d;
#pragma unroll
        for (int v = 0; v < NV; ++v) xr[u][v] = (D64 || lane + 64 * v < d) ? xp[lane + 64 * v] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (whole_rows) {
          while (b + js[u] >= cur_end) flush();
        }
        const float w = gcr_readlane_f(vv, js[u]);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = fmaf(w, xr[u][v], acc[v]);
      }
      cnt -= UNR;
    }
    while (cnt > 0) {
      const int j = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      const float* xp = x + (int64_t)gcr_readlane_i(cv, j) * d;
      float xr[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) xr[v] = (D64 || lane + 64 * v < d) ? xp[lane + 64 * v] : 0.f;
      if (whole_rows) {
        while (b + j >= cur_end) flush();
      }
      const float w = gcr_readlane_f(vv, j);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = fmaf(w, xr[v], acc[v]);
      --cnt;
    }
  }

  if (whole_rows) {
    while (cur < nrows) flush();
  } else {
    const int64_t base = slot * (int64_t)d;
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (D64 || lane + 64 * v < d) partials[base + lane + 64 * v] = acc[v];
  }
}

// one 4-wave block per long row: wave w adds chunk partials w, w+4, ... (4 loads in flight each),
// the four wave sums are combined through LDS in wave order, then the common epilogue.  The order
// of additions is fixed by the plan, so the result stays bitwise reproducible.  (One wave per row
// took 24 us per cfg2 layer — hub rows have ~100 chunks — i.e. 3 % of the layer.)
template <int NV, bool D64, bool ACC2>
__global__ __launch_bounds__(256) void spmm_long_rows(const int32_t* __restrict__ long_row,
                                                      const int32_t* __restrict__ long_slot0, int64_t n_long,
                                                      const float* __restrict__ partials, int d, Epilogue ep) {
  __shared__ float red[3][NV * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t i = blockIdx.x;
  const int s0 = long_slot0[i], s1 = long_slot0[i + 1];
  float acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
  int s = s0 + wave;
  for (; s + 12 < s1; s += 16) {
    float t[4][NV];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v)
        t[u][v] = (D64 || lane + 64 * v < d) ? partials[(int64_t)(s + 4 * u) * d + lane + 64 * v] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += t[u][v];
  }
  for (; s < s1; s += 4)
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (D64 || lane + 64 * v < d) acc[v] += partials[(int64_t)s * d + lane + 64 * v];
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) red[wave - 1][v * 64 + lane] = acc[v];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] += red[w][v * 64 + lane];
    store_row<NV, D64, ACC2>(ep, (int64_t)long_row[i], d, lane, acc);
  }
}

__global__ void csr_validate_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                    int64_t n_rows, int64_t n_cols, int64_t nnz,
                                    unsigned long long* __restrict__ n_errors) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride)
    bad += rowptr[i + 1] < rowptr[i];
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride)
    bad += (col[e] < 0) | ((int64_t)col[e] >= n_cols);
  if (blockIdx.x == 0 && threadIdx.x == 0) bad += (rowptr[0] != 0) + (rowptr[n_rows] != nnz);
  if (bad) atomicAdd(n_errors, bad);
}

template <int NV, bool D64, bool ACC2>
int32_t launch_spmm_a(const int64_t* desc, int64_t n_parts, const int32_t* long_row, const int32_t* long_slot0,
                    int64_t n_long, const int64_t* rowptr, const int32_t* col, const float* val,
                    const uint32_t* keep_bits, const float* x, int d, const Epilogue& ep, float* partials,
                    hipStream_t stream) {
  const unsigned blocks = (unsigned)((n_parts + 3) / 4);
  if (blocks > 0) {
#define GCR_SPMM_LAUNCH(HV, MK)                                                                             \
  hipLaunchKernelGGL((spmm_parts<NV, D64, HV, MK, unroll_for<NV>(), ACC2>), dim3(blocks), dim3(256), 0, stream, desc, n_parts, \
                     rowptr, col, val, keep_bits, x, d, ep, partials)
    if (ep.col_bits != nullptr) {
      if (val != nullptr)
        hipLaunchKernelGGL((spmm_parts<NV, D64, true, false, unroll_for<NV>(), ACC2, true>), dim3(blocks), dim3(256), 0,
                           stream, desc, n_parts, rowptr, col, val, ep.col_bits, x, d, ep, partials);
      else
        hipLaunchKernelGGL((spmm_parts<NV, D64, false, false, unroll_for<NV>(), ACC2, true>), dim3(blocks), dim3(256), 0,
                           stream, desc, n_parts, rowptr, col, val, ep.col_bits, x, d, ep, partials);
    } else if (val != nullptr) {
      if (keep_bits != nullptr) GCR_SPMM_LAUNCH(true, true);
      else GCR_SPMM_LAUNCH(true, false);
    } else {
      if (keep_bits != nullptr) GCR_SPMM_LAUNCH(false, true);
      else GCR_SPMM_LAUNCH(false, false);
    }
#undef GCR_SPMM_LAUNCH
    int32_t st = GCR_LAUNCH_STATUS();
    if (st != GCR_OK) return st;
  }
  if (n_long > 0) {
    hipLaunchKernelGGL((spmm_long_rows<NV, D64, ACC2>), dim3((unsigned)n_long), dim3(256), 0, stream,
                       long_row, long_slot0, n_long, partials, d, ep);
    return GCR_LAUNCH_STATUS();
  }
  return GCR_OK;
}

template <int NV, bool D64>
int32_t launch_spmm(const int64_t* desc, int64_t n_parts, const int32_t* long_row, const int32_t* long_slot0,
                    int64_t n_long, const int64_t* rowptr, const int32_t* col, const float* val,
                    const uint32_t* keep_bits, const float* x, int d, const Epilogue& ep, float* partials,
                    hipStream_t stream) {
  if (ep.acc_in2 != nullptr)
    return launch_spmm_a<NV, D64, true>(desc, n_parts, long_row, long_slot0, n_long, rowptr, col, val, keep_bits, x, d, ep,
                                        partials, stream);
  return launch_spmm_a<NV, D64, false>(desc, n_parts, long_row, long_slot0, n_long, rowptr, col, val, keep_bits, x, d, ep,
                                       partials, stream);
}

}  // namespace

extern "C" int32_t gcr_spmm_csr_acc2_f32(const int64_t* desc, int64_t n_parts, const int32_t* long_row,
                                         const int32_t* long_slot0, int64_t n_long_rows, const int64_t* rowptr,
                                         const int32_t* col, const float* val, const uint32_t* keep_bits,
                                         float val_scale, const float* x, int32_t d, float* y, const float* acc_in,
                                         const float* acc_in2, float acc_in2_scale, float* acc_out, float acc_scale,
                                         uint32_t flags, float* inv_norm_out, float* partials, int64_t n_rows,
                                         int64_t n_cols, const uint32_t* col_active_bits, void* stream) {
  GCR_CHECK_ARG(n_parts >= 0 && n_long_rows >= 0 && n_rows >= 0 && n_cols >= 0);
  GCR_CHECK_ARG(n_parts < (1ll << 31) - 4 && n_rows < (1ll << 31) && n_cols < (1ll << 31));
  GCR_CHECK_ARG(d >= 1 && d <= 256);
  if (n_rows == 0 || n_parts == 0) return GCR_OK;
  GCR_CHECK_ARG(desc != nullptr && rowptr != nullptr && x != nullptr);
  GCR_CHECK_ARG(y != nullptr || acc_out != nullptr);
  GCR_CHECK_ARG(acc_in2 == nullptr || acc_out != nullptr);
  GCR_CHECK_ARG(col_active_bits == nullptr || keep_bits == nullptr);       // one predicate per launch
  GCR_CHECK_ARG(n_long_rows == 0 || (long_row != nullptr && long_slot0 != nullptr && partials != nullptr));
  GCR_CHECK_ARG((flags & ~GCR_SPMM_ROW_L2NORM) == 0);
  Epilogue ep{val_scale, y, acc_in, acc_out, acc_scale, flags, inv_norm_out, nullptr, acc_in2, acc_in2_scale, col_active_bits};
  hipStream_t s = (hipStream_t)stream;
#define GCR_GO(NV, D64) \
  return launch_spmm<NV, D64>(desc, n_parts, long_row, long_slot0, n_long_rows, rowptr, col, val, keep_bits, x, d, ep, partials, s)
  if (d == 64) GCR_GO(1, true);
  if (d <= 64) GCR_GO(1, false);
  if (d <= 128) GCR_GO(2, false);
  if (d <= 192) GCR_GO(3, false);
  GCR_GO(4, false);
#undef GCR_GO
}

extern "C" int32_t gcr_spmm_csr_f32(const int64_t* desc, int64_t n_parts, const int32_t* long_row,
                                    const int32_t* long_slot0, int64_t n_long_rows, const int64_t* rowptr,
                                    const int32_t* col, const float* val, const uint32_t* keep_bits,
                                    float val_scale, const float* x, int32_t d, float* y, const float* acc_in,
                                    float* acc_out, float acc_scale, uint32_t flags, float* inv_norm_out,
                                    float* partials, int64_t n_rows, int64_t n_cols, void* stream) {
  return gcr_spmm_csr_acc2_f32(desc, n_parts, long_row, long_slot0, n_long_rows, rowptr, col, val, keep_bits, val_scale,
                               x, d, y, acc_in, nullptr, 0.f, acc_out, acc_scale, flags, inv_norm_out, partials, n_rows,
                               n_cols, nullptr, stream);
}

extern "C" int32_t gcr_spmm_csr_dual_f32(const int64_t* desc, int64_t n_parts, const int32_t* long_row,
                                         const int32_t* long_slot0, int64_t n_long_rows, const int64_t* rowptr,
                                         const int32_t* col, const float* val, const uint32_t* keep_bits,
                                         float val_scale, const float* x, int32_t d, float* y_raw, float* y_norm,
                                         float* inv_norm_out, float* partials, int64_t n_rows, int64_t n_cols,
                                         void* stream) {
  GCR_CHECK_ARG(n_parts >= 0 && n_long_rows >= 0 && n_rows >= 0 && n_cols >= 0);
  GCR_CHECK_ARG(n_parts < (1ll << 31) - 4 && n_rows < (1ll << 31) && n_cols < (1ll << 31));
  GCR_CHECK_ARG(d >= 1 && d <= 256);
  if (n_rows == 0 || n_parts == 0) return GCR_OK;
  GCR_CHECK_ARG(desc != nullptr && rowptr != nullptr && x != nullptr && y_raw != nullptr && y_norm != nullptr);
  GCR_CHECK_ARG(y_raw != y_norm);
  GCR_CHECK_ARG(n_long_rows == 0 || (long_row != nullptr && long_slot0 != nullptr && partials != nullptr));
  Epilogue ep{val_scale, y_norm, nullptr, nullptr, 1.0f, GCR_SPMM_ROW_L2NORM, inv_norm_out, y_raw, nullptr, 0.f, nullptr};
  hipStream_t s = (hipStream_t)stream;
#define GCR_GO(NV, D64) \
  return launch_spmm<NV, D64>(desc, n_parts, long_row, long_slot0, n_long_rows, rowptr, col, val, keep_bits, x, d, ep, partials, s)
  if (d == 64) GCR_GO(1, true);
  if (d <= 64) GCR_GO(1, false);
  if (d <= 128) GCR_GO(2, false);
  if (d <= 192) GCR_GO(3, false);
  GCR_GO(4, false);
#undef GCR_GO
}

// The dual launch with the layer-list accumulation folded in (mhcn.py:440-457 appends the normalised product of every layer
// to a list that is summed afterwards): acc_out = acc_in + normalize(A x) from the same pass; y_norm may be NULL — the backward
// rebuilds the normalised rows from y_raw and inv_norm_out (gcr_normalize_bwd_raw_f32).
extern "C" int32_t gcr_spmm_csr_dual_acc_f32(const int64_t* desc, int64_t n_parts, const int32_t* long_row,
                                             const int32_t* long_slot0, int64_t n_long_rows, const int64_t* rowptr,
                                             const int32_t* col, const float* val, const uint32_t* keep_bits,
                                             float val_scale, const float* x, int32_t d, float* y_raw, float* y_norm,
                                             const float* acc_in, float* acc_out, float* inv_norm_out, float* partials,
                                             int64_t n_rows, int64_t n_cols, void* stream) {
  GCR_CHECK_ARG(n_parts >= 0 && n_long_rows >= 0 && n_rows >= 0 && n_cols >= 0);
  GCR_CHECK_ARG(n_parts < (1ll << 31) - 4 && n_rows < (1ll << 31) && n_cols < (1ll << 31));
  GCR_CHECK_ARG(d >= 1 && d <= 256);
  if (n_rows == 0 || n_parts == 0) return GCR_OK;
  GCR_CHECK_ARG(desc != nullptr && rowptr != nullptr && x != nullptr && y_raw != nullptr && acc_out != nullptr);
  GCR_CHECK_ARG(y_raw != y_norm && y_raw != acc_out && (y_norm != nullptr || inv_norm_out != nullptr));
  GCR_CHECK_ARG(n_long_rows == 0 || (long_row != nullptr && long_slot0 != nullptr && partials != nullptr));
  Epilogue ep{val_scale, y_norm, acc_in, acc_out, 1.0f, GCR_SPMM_ROW_L2NORM, inv_norm_out, y_raw, nullptr, 0.f, nullptr};
  hipStream_t s = (hipStream_t)stream;
#define GCR_GO(NV, D64) \
  return launch_spmm<NV, D64>(desc, n_parts, long_row, long_slot0, n_long_rows, rowptr, col, val, keep_bits, x, d, ep, partials, s)
  if (d == 64) GCR_GO(1, true);
  if (d <= 64) GCR_GO(1, false);
  if (d <= 128) GCR_GO(2, false);
  if (d <= 192) GCR_GO(3, false);
  GCR_GO(4, false);
#undef GCR_GO
}

extern "C" int32_t gcr_csr_validate(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int64_t n_cols,
                                    int64_t nnz, int64_t* n_errors_dev, void* stream) {
  GCR_CHECK_ARG(rowptr != nullptr && n_errors_dev != nullptr && n_rows >= 0 && nnz >= 0 && n_cols >= 0);
  GCR_CHECK_ARG(nnz == 0 || col != nullptr);
  hipLaunchKernelGGL(csr_validate_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, rowptr, col, n_rows,
                     n_cols, nnz, (unsigned long long*)n_errors_dev);
  return GCR_LAUNCH_STATUS();
}
