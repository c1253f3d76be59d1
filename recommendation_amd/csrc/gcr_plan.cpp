// Host-side SpMM work planner (one-off per graph) and library identification.
//
// The reference rebuilds nothing per step either: its adjacency is made once in
// Interaction._build_adj (ncl.py:74-85) / normalize_graph_mat (selfcf.py:240-255).  The plan
// turns the degree skew of the bipartite graph (Zipf item popularity: a few item rows hold
// most non-zeros) into equal-sized units of work for the 64-lane waves of gcr_spmm.hip.
#include <stdint.h>

#include "gcr.h"

namespace {

constexpr int kMaxRowsPerPart = 64;  // one row-end offset per lane

// Calls emit(nnz0, nnz1, row0, nrows, slot) for every partition and long(row, slot0, slot1)
// for every split row; returns the number of partial slots.
template <class EmitPart, class EmitLong>
int64_t walk(const int64_t* rowptr, int64_t n_rows, int32_t L, EmitPart emit, EmitLong emit_long) {
  int64_t slots = 0;
  int64_t cur_row0 = 0, cur_nnz0 = rowptr[0], cur_rows = 0;
  auto close = [&](int64_t row_end) {
    if (cur_rows > 0) emit(cur_nnz0, rowptr[row_end], cur_row0, cur_rows, (int64_t)-1);
    cur_rows = 0;
  };
  for (int64_t r = 0; r < n_rows; ++r) {
    const int64_t deg = rowptr[r + 1] - rowptr[r];
    if (deg > L) {
      close(r);
      const int64_t chunks = (deg + L - 1) / L;
      const int64_t slot0 = slots;
      for (int64_t c = 0; c < chunks; ++c) {
        // equal chunks: boundaries at floor(c * deg / chunks)
        const int64_t a = rowptr[r] + (deg * c) / chunks;
        const int64_t b = rowptr[r] + (deg * (c + 1)) / chunks;
        emit(a, b, r, (int64_t)1, slots++);
      }
      emit_long(r, slot0, slots);
      cur_row0 = r + 1;
      cur_nnz0 = rowptr[r + 1];
      continue;
    }
    if (cur_rows == 0) {
      cur_row0 = r;
      cur_nnz0 = rowptr[r];
    } else if (cur_rows == kMaxRowsPerPart || (rowptr[r + 1] - cur_nnz0) > L) {
      close(r);
      cur_row0 = r;
      cur_nnz0 = rowptr[r];
    }
    ++cur_rows;
  }
  close(n_rows);
  return slots;
}

}  // namespace

extern "C" int32_t gcr_version(void) { return 100; }
extern "C" const char* gcr_arch(void) { return "gfx950"; }

extern "C" int32_t gcr_spmm_plan_size_host(const int64_t* rowptr_host, int64_t n_rows, int32_t nnz_per_part,
                                           int64_t* n_parts, int64_t* n_long_rows, int64_t* n_slots) {
  if (rowptr_host == nullptr || n_parts == nullptr || n_long_rows == nullptr || n_slots == nullptr) return GCR_EINVAL;
  if (n_rows < 0 || n_rows >= (1ll << 31) || nnz_per_part < 64 || nnz_per_part > (1 << 20)) return GCR_EINVAL;
  for (int64_t r = 0; r < n_rows; ++r)
    if (rowptr_host[r + 1] < rowptr_host[r]) return GCR_EINVAL;
  int64_t parts = 0, longs = 0;
  *n_slots = walk(
      rowptr_host, n_rows, nnz_per_part, [&](int64_t, int64_t, int64_t, int64_t, int64_t) { ++parts; },
      [&](int64_t, int64_t, int64_t) { ++longs; });
  *n_parts = parts;
  *n_long_rows = longs;
  return GCR_OK;
}

extern "C" int32_t gcr_spmm_plan_fill_host(const int64_t* rowptr_host, int64_t n_rows, int32_t nnz_per_part,
                                           int64_t* desc_host, int32_t* long_row_host, int32_t* long_slot0_host) {
  if (rowptr_host == nullptr || desc_host == nullptr) return GCR_EINVAL;
  if (n_rows < 0 || n_rows >= (1ll << 31) || nnz_per_part < 64 || nnz_per_part > (1 << 20)) return GCR_EINVAL;
  int64_t p = 0, l = 0;
  int64_t slots = walk(
      rowptr_host, n_rows, nnz_per_part,
      [&](int64_t a, int64_t b, int64_t row0, int64_t nrows, int64_t slot) {
        desc_host[4 * p + 0] = a;
        desc_host[4 * p + 1] = b;
        desc_host[4 * p + 2] = row0 | (nrows << 32);
        desc_host[4 * p + 3] = slot;
        ++p;
      },
      [&](int64_t row, int64_t s0, int64_t) {
        long_row_host[l] = (int32_t)row;
        long_slot0_host[l] = (int32_t)s0;
        ++l;
      });
  if (l > 0) long_slot0_host[l] = (int32_t)slots;
  return slots < (1ll << 31) ? GCR_OK : GCR_EUNSUPPORTED;
}
