/* TEST INFRASTRUCTURE ONLY — plain-C restatement of the reference hot path.
 *
 * Used (a) by tests as a second, independent checker at sizes numpy is too slow for, and
 * (b) by bench.py's `cpu_baseline` leg ("kind": "port") timed on the GPU box's host cores.
 * Never linked into, or called by, recommendation_amd/.  Pinned against the numpy oracle, which
 * is itself pinned against the reference-generated golden fixtures (tests/test_oracle_c.py).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/oracle.c -o oracle/liboracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ncl.py:419 torch.sparse.mm(A, emb) on the canonical CSR; fp32 accumulate like the CPU
 * PyTorch path.  val == NULL -> all ones (raw adjacency, ncl.py:74-85). */
void orc_spmm_csr_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                      const float* x, int d, float* y) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t r = 0; r < n_rows; ++r) {
    float* yr = y + r * d;
    for (int c = 0; c < d; ++c) yr[c] = 0.f;
    for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const float w = val ? val[e] : 1.0f;
      const float* xr = x + (int64_t)col[e] * d;
      for (int c = 0; c < d; ++c) yr[c] += w * xr[c];
    }
  }
}

/* The COO formulation the reference actually runs (uncoalesced COO, ncl.py:203-209): serial
 * scatter-add in COO order, as a single-thread restatement of the same arithmetic. */
void orc_spmm_coo_f32(const int64_t* row, const int64_t* col, const float* val, int64_t nnz, int64_t n_rows,
                      const float* x, int d, float* y) {
  memset(y, 0, sizeof(float) * (size_t)n_rows * d);
  for (int64_t e = 0; e < nnz; ++e) {
    float* yr = y + row[e] * d;
    const float* xr = x + col[e] * d;
    const float w = val[e];
    for (int c = 0; c < d; ++c) yr[c] += w * xr[c];
  }
}

/* ncl.py:415-422 / lightgcn.py:21-27: K layers + sum (scale = 1) or mean (scale = 1/(K+1)).
 * work: 2 * n * d floats. */
void orc_lightgcn_propagate_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                const float* x0, int d, int n_layers, float scale, float* out, float* work) {
  float* cur = work;
  float* nxt = work + n * d;
  memcpy(out, x0, sizeof(float) * (size_t)n * d);
  const float* src = x0;
  for (int k = 0; k < n_layers; ++k) {
    orc_spmm_csr_f32(rowptr, col, val, n, src, d, nxt);
#pragma omp parallel for
    for (int64_t i = 0; i < n * d; ++i) out[i] += nxt[i];
    float* t = cur;
    cur = nxt;
    nxt = t;
    src = cur;
  }
#pragma omp parallel for
  for (int64_t i = 0; i < n * d; ++i) out[i] *= scale;
}

static void normalize_rows(const float* x, int64_t n, int d, float* out) {
#pragma omp parallel for
  for (int64_t i = 0; i < n; ++i) {
    double ss = 0;
    for (int c = 0; c < d; ++c) ss += (double)x[i * d + c] * x[i * d + c];
    const float inv = 1.0f / fmaxf((float)sqrt(ss), 1e-12f);
    for (int c = 0; c < d; ++c) out[i * d + c] = x[i * d + c] * inv;
  }
}

/* Row logsumexp of S = a b^T * inv_tau and the positive logit s[i, pos[i]] — the common core of
 * InfoNCE (ncl.py:125-130), info_nce_loss (gcl.py:28-35), ssl_layer_loss (ncl.py:358-367) and
 * batch_softmax_loss (ssl4rec.py:25-30).  fp32 dot products like torch's CPU GEMM, fp64 LSE.
 * scratch: (m + n) * d floats when normalize != 0. */
void orc_row_lse_f32(const float* a, int64_t m, const float* b, int64_t n, int d, const int64_t* pos,
                     float inv_tau, int normalize, float* scratch, double* lse, double* pos_logit) {
  const float* an = a;
  const float* bn = b;
  if (normalize) {
    normalize_rows(a, m, d, scratch);
    normalize_rows(b, n, d, scratch + m * d);
    an = scratch;
    bn = scratch + m * d;
  }
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t i = 0; i < m; ++i) {
    double mx = -INFINITY, sum = 0;
    const float* ai = an + i * d;
    for (int64_t j = 0; j < n; ++j) {
      const float* bj = bn + j * d;
      float dot = 0.f;
      for (int c = 0; c < d; ++c) dot += ai[c] * bj[c];
      const double s = (double)(dot * inv_tau);
      if (pos && pos[i] == j) pos_logit[i] = s;
      if (s > mx) {
        sum = sum * exp(mx - s) + 1.0;
        mx = s;
      } else {
        sum += exp(s - mx);
      }
    }
    lse[i] = mx + log(sum);
  }
}

/* BPR over gathered rows: ncl.py:116-120 (variant 0: -log(1e-5 + sigmoid)), gcl.py:221 /
 * sept.py:34-38 (1: -logsigmoid), lightgcn.py:95-108 (2: -log(sigmoid), n_neg negatives averaged). */
double orc_bpr_loss_f32(const float* user_tab, const float* item_tab, int d, const int64_t* u, const int64_t* i,
                        const int64_t* j, int64_t batch, int n_neg, int variant) {
  double total = 0;
#pragma omp parallel for reduction(+ : total)
  for (int64_t b = 0; b < batch; ++b) {
    const float* ue = user_tab + u[b] * d;
    const float* pe = item_tab + i[b] * d;
    float pos = 0.f, neg = 0.f;
    for (int c = 0; c < d; ++c) pos += ue[c] * pe[c];
    for (int k = 0; k < n_neg; ++k) {
      const float* ne = item_tab + j[b * n_neg + k] * d;
      float s = 0.f;
      for (int c = 0; c < d; ++c) s += ue[c] * ne[c];
      neg += s;
    }
    const double x = (double)pos - (double)neg / n_neg;
    const double sg = 1.0 / (1.0 + exp(-x));
    double l;
    if (variant == 0) l = -log(10e-6 + sg);
    else if (variant == 1) l = (x > 0 ? log1p(exp(-x)) : -x + log1p(exp(x)));
    else l = -log(sg);
    total += l;
  }
  return total / (double)batch;
}

/* Philox-4x32-10 (Salmon et al. 2011) */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

#define ORC_STREAM_NEG 0x4E454753u
#define ORC_STREAM_EDGE 0x45444745u

/* Counter-RNG restatement of the sampler contract (ncl.py:91-114, gcl.py:111-125,
 * ssl4rec.py:33-50, lightgcn.py:91-94); see oracle_np.neg_sample_uniform. */
void orc_neg_sample(const int64_t* user_rowptr, const int32_t* user_items_sorted, const int64_t* u_idx,
                    int64_t batch, int n_negs, int64_t num_items, uint64_t seed, uint64_t offset, int max_trials,
                    int64_t* out) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma omp parallel for
  for (int64_t s = 0; s < batch * n_negs; ++s) {
    const int64_t user = u_idx[s / n_negs];
    const int64_t lo0 = user_rowptr[user], hi0 = user_rowptr[user + 1];
    const uint64_t slot = offset + (uint64_t)s;
    int64_t res = -1;
    const int draws = max_trials > 0 ? max_trials : 1;
    uint32_t c[4];
    for (int t = 0; t < draws; ++t) {
      if ((t & 3) == 0) {
        c[0] = (uint32_t)slot; c[1] = (uint32_t)(slot >> 32); c[2] = (uint32_t)(t >> 2); c[3] = ORC_STREAM_NEG;
        philox4x32_10(c, k0, k1);
      }
      const int64_t cand = (int64_t)(((uint64_t)c[t & 3] * (uint64_t)num_items) >> 32);
      if (max_trials == 0) { res = cand; break; }
      int64_t lo = lo0, hi = hi0;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (user_items_sorted[mid] < cand) lo = mid + 1; else hi = mid;
      }
      if (!(lo < hi0 && user_items_sorted[lo] == cand)) { res = cand; break; }
    }
    out[s] = res;
  }
}

/* gcl.py:22-25 Bernoulli keep mask `rand >= pe`; see oracle_np.edge_keep_mask. */
void orc_edge_keep_mask(int64_t nnz, float pe, uint64_t seed, uint8_t* keep) {
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma omp parallel for
  for (int64_t blk = 0; blk < (nnz + 3) / 4; ++blk) {
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)((uint64_t)blk >> 32), 0u, ORC_STREAM_EDGE};
    philox4x32_10(c, k0, k1);
    for (int w = 0; w < 4 && blk * 4 + w < nnz; ++w) {
      const float u = (float)(c[w] >> 8) * 5.9604644775390625e-8f; /* 2^-24 */
      keep[blk * 4 + w] = u >= pe;
    }
  }
}
