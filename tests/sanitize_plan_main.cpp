// Driver for tests/test_sanitizers_cpu.py: the host-side SpMM planner (recommendation_amd/csrc/gcr_plan.cpp) built
// with -fsanitize=address,undefined and run over adversarial row-pointer arrays; every non-zero must be covered
// exactly once and every buffer access must stay inside what gcr_spmm_plan_size_host reported.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gcr.h"

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static int check(const std::vector<int64_t>& rowptr, int32_t L) {
  const int64_t n_rows = (int64_t)rowptr.size() - 1;
  int64_t n_parts = -1, n_long = -1, n_slots = -1;
  if (gcr_spmm_plan_size_host(rowptr.data(), n_rows, L, &n_parts, &n_long, &n_slots) != GCR_OK) return 1;
  std::vector<int64_t> desc((size_t)n_parts * 4 + 1, -7);          // exact sizes: an overrun trips the sanitizer
  std::vector<int32_t> long_row((size_t)n_long + 1, -7), long_slot0((size_t)n_long + 2, -7);
  if (gcr_spmm_plan_fill_host(rowptr.data(), n_rows, L, desc.data(), long_row.data(), long_slot0.data()) != GCR_OK) return 2;
  const int64_t nnz = rowptr[n_rows];
  std::vector<unsigned char> seen((size_t)nnz, 0);
  for (int64_t p = 0; p < n_parts; ++p) {
    const int64_t b = desc[4 * p], e = desc[4 * p + 1];
    if (b < 0 || e < b || e > nnz || e - b > L) return 3;
    for (int64_t k = b; k < e; ++k) {
      if (seen[(size_t)k]) return 4;
      seen[(size_t)k] = 1;
    }
  }
  for (int64_t k = 0; k < nnz; ++k)
    if (!seen[(size_t)k]) return 5;
  if (desc[(size_t)n_parts * 4] != -7 || long_row[(size_t)n_long] != -7) return 6;   // guards untouched
  return 0;
}

int main() {
  int cases = 0;
  for (int32_t L : {64, 256, 4096}) {
    for (int shape = 0; shape < 6; ++shape) {
      for (int rep = 0; rep < 8; ++rep) {
        const int64_t n_rows = shape == 0 ? 0 : (shape == 1 ? 1 : 1 + (int64_t)(rnd() % 3000));
        std::vector<int64_t> rowptr((size_t)n_rows + 1, 0);
        for (int64_t r = 0; r < n_rows; ++r) {
          int64_t deg;
          switch (shape) {
            case 2: deg = 0; break;                                       // all rows empty
            case 3: deg = (int64_t)(rnd() % 5); break;                    // short rows, many empty
            case 4: deg = (rnd() % 50 == 0) ? (int64_t)(rnd() % 60000) : (int64_t)(rnd() % 8); break;   // a few huge rows
            case 5: deg = L + (int64_t)(rnd() % 3) - 1; break;            // rows right at the partition size
            default: deg = (int64_t)(rnd() % 200); break;
          }
          rowptr[(size_t)r + 1] = rowptr[(size_t)r] + deg;
        }
        const int rc = check(rowptr, L);
        if (rc != 0) {
          std::printf("FAIL shape %d rep %d L %d rows %lld: code %d\n", shape, rep, (int)L, (long long)n_rows, rc);
          return 1;
        }
        ++cases;
      }
    }
  }
  std::printf("planner under ASan/UBSan: %d cases OK\n", cases);
  return 0;
}
