/*
 * gcr.h — C ABI of libgcr (graph-contrastive recommender hot path, MI355X / gfx950).
 *
 * The reference (Cmint22/Recommendation) has no FFI of its own: its op boundary is the stock
 * PyTorch call inside each model class.  Each entry point below names the
 * reference call site(s) it replaces.  A maintainer binds them with ctypes;
 * recommendation_amd/_lib.py is that binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; buffers are caller-owned
 *     and only borrowed for the call; no hidden allocations, workspaces are passed in;
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it; no global
 *     mutable state, re-entrant across streams;
 *   - dense matrices are row-major contiguous fp32 with row stride d; column ids int32,
 *     row pointers int64;
 *   - return value: 0 = ok, GCR_EINVAL.. = argument error (nothing launched),
 *     -(1000 + hipError_t) = runtime error.
 */
#ifndef GCR_H
#define GCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCR_OK 0
#define GCR_EINVAL (-1)      /* null pointer / negative size / unsupported d */
#define GCR_EUNSUPPORTED (-2)
#define GCR_HIP_ERROR_BASE (-1000)

/* library / build identification: returns e.g. 100 for 0.1.0; arch string is "gfx950". */
int32_t gcr_version(void);
const char* gcr_arch(void);

/* ---------------------------------------------------------------------------------------------
 * SpMM work plan (host, one-off per graph — the reference also builds its adjacency once:
 * ncl.py:74-85, selfcf.py:291-306).  Rows are packed into partitions of <= nnz_per_part
 * non-zeros (<= 64 whole rows each); rows longer than nnz_per_part are split into equal chunks
 * whose partial sums are combined in a fixed order (deterministic, no atomics).
 * desc is [n_parts][4] int64: {nnz_begin, nnz_end, row0 | (nrows << 32), slot (or -1)}.
 * --------------------------------------------------------------------------------------------- */
int32_t gcr_spmm_plan_size_host(const int64_t* rowptr_host, int64_t n_rows, int32_t nnz_per_part,
                                int64_t* n_parts, int64_t* n_long_rows, int64_t* n_slots);
int32_t gcr_spmm_plan_fill_host(const int64_t* rowptr_host, int64_t n_rows, int32_t nnz_per_part,
                                int64_t* desc_host, int32_t* long_row_host, int32_t* long_slot0_host);

/* flags for gcr_spmm_csr_f32 */
#define GCR_SPMM_ROW_L2NORM 1u /* y <- y / max(||y||_2, 1e-12) per row (sept.py:224, sept_social.py:373-374) */

/*
 * y = epilogue( val_scale * A[keep] x )            A: CSR [n_rows, n_cols], x: [n_cols, d]
 * replaces  torch.sparse.mm(self.sparse_norm_adj, emb)   ncl.py:419, directau.py:290,
 *           selfcf.py:479, buir.py:317,334, sept.py:223, sept_social.py:373,382,
 *           mhcn.py:440-456,494 and LGConv()(x, edge_index) lightgcn.py:25 (after gcn_norm).
 * Fused layer combine (lightgcn.py:26 `x += out`, ncl.py:421 mean over layers):
 *           acc_out[r] = (acc_in[r] + y[r]) * acc_scale     when acc_out != NULL
 * val == NULL means all-ones values (the raw 0/1 adjacency of ncl.py:74-85).
 * keep_bits == NULL means no edge mask; otherwise bit e of the little-endian uint32 bitmap keeps
 * non-zero e (gcl.py:22-25 / sept.py:55-61 / buir.py:300-309 edge dropout consumed as a predicate).
 * inv_norm_out (optional, [n_rows]) receives 1/max(||.||,1e-12) under GCR_SPMM_ROW_L2NORM.
 * y may be NULL when only acc_out is wanted.  partials: fp32 workspace [n_slots, d].
 */
int32_t gcr_spmm_csr_f32(const int64_t* desc, int64_t n_parts,
                         const int32_t* long_row, const int32_t* long_slot0, int64_t n_long_rows,
                         const int64_t* rowptr, const int32_t* col, const float* val,
                         const uint32_t* keep_bits, float val_scale,
                         const float* x, int32_t d,
                         float* y, const float* acc_in, float* acc_out, float acc_scale,
                         uint32_t flags, float* inv_norm_out, float* partials,
                         int64_t n_rows, int64_t n_cols, void* stream);

/*
 * gcr_spmm_csr_f32 with a second addend in the combine:
 *           acc_out[r] = (acc_in[r] + acc_in2_scale * acc_in2[r] + y[r]) * acc_scale
 * (acc_in2 == NULL: exactly gcr_spmm_csr_f32).  The backward of the K-layer message pass with per-layer outputs
 * (ncl.py:415-422 returns `all_emb`; ncl.py:319-322 reads emb_list[2 * hyper_layers]) is the Horner recurrence
 * h_k = g_final + g_k / c + A^T h_{k+1}: the per-layer gradient g_k rides on the epilogue instead of a separate
 * [n_rows, d] add pass per layer.
 * col_active_bits (optional, excludes keep_bits): a little-endian uint32 bitmap over the COLUMNS, bit c clear = row c of x
 * is known to be zero; non-zeros with a clear column are skipped before their gather.  The first launch of a backward
 * pass whose incoming gradient touches a few thousand rows (the batch rows of ncl.py:314-317) then moves the CSR and its
 * output but gathers almost nothing.  gcr_bitmap_set builds the bitmap from an index list.
 */
int32_t gcr_spmm_csr_acc2_f32(const int64_t* desc, int64_t n_parts,
                              const int32_t* long_row, const int32_t* long_slot0, int64_t n_long_rows,
                              const int64_t* rowptr, const int32_t* col, const float* val,
                              const uint32_t* keep_bits, float val_scale,
                              const float* x, int32_t d,
                              float* y, const float* acc_in, const float* acc_in2, float acc_in2_scale,
                              float* acc_out, float acc_scale,
                              uint32_t flags, float* inv_norm_out, float* partials,
                              int64_t n_rows, int64_t n_cols, const uint32_t* col_active_bits, void* stream);
/* bits[idx[i] >> 5] |= 1 << (idx[i] & 31) for every idx[i] in [0, n_bits) (atomic OR; the caller zeroes `bits`). */
int32_t gcr_bitmap_set(const int64_t* idx, int64_t n, int64_t n_bits, uint32_t* bits, void* stream);

/*
 * The same product with TWO outputs from one launch:
 *   y_raw  = val_scale * A[keep] x                     (fed to the next layer)
 *   y_norm = y_raw / max(||y_raw||_2, 1e-12) per row   (appended to the layer list)
 * replaces  user_embeddings_c1 = torch.sparse.mm(self.H_s, user_embeddings_c1);
 *           norm_embeddings_c1 = F.normalize(user_embeddings_c1, p=2, dim=1)     univariate/mhcn.py:440-457
 * (SEPT feeds the NORMALISED rows forward, sept.py:223-224: that is gcr_spmm_csr_f32 with
 * GCR_SPMM_ROW_L2NORM).  inv_norm_out (optional, [n_rows]) as above.  y_raw != y_norm.
 */
int32_t gcr_spmm_csr_dual_f32(const int64_t* desc, int64_t n_parts,
                              const int32_t* long_row, const int32_t* long_slot0, int64_t n_long_rows,
                              const int64_t* rowptr, const int32_t* col, const float* val,
                              const uint32_t* keep_bits, float val_scale,
                              const float* x, int32_t d, float* y_raw, float* y_norm, float* inv_norm_out,
                              float* partials, int64_t n_rows, int64_t n_cols, void* stream);

/* The dual launch with the layer-list accumulation folded in (univariate/mhcn.py:440-457 appends the normalised product of
 * every layer to a list that is summed afterwards):  acc_out = acc_in + normalize(A x)  from the same pass (acc_in may be NULL,
 * acc_out may alias acc_in).  y_norm may be NULL — inv_norm_out is then required and the backward rebuilds the normalised
 * rows from y_raw (gcr_normalize_bwd_raw_f32): three [n_rows, d] passes instead of the five of dual launch + add. */
int32_t gcr_spmm_csr_dual_acc_f32(const int64_t* desc, int64_t n_parts,
                                  const int32_t* long_row, const int32_t* long_slot0, int64_t n_long_rows,
                                  const int64_t* rowptr, const int32_t* col, const float* val,
                                  const uint32_t* keep_bits, float val_scale,
                                  const float* x, int32_t d, float* y_raw, float* y_norm, const float* acc_in,
                                  float* acc_out, float* inv_norm_out, float* partials, int64_t n_rows, int64_t n_cols,
                                  void* stream);

/* Counts structural errors of a CSR on the device (rowptr not monotone / not ending at nnz,
 * col outside [0, n_cols)); *n_errors_dev is a device int64 the caller zeroes and reads back. */
int32_t gcr_csr_validate(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int64_t n_cols,
                         int64_t nnz, int64_t* n_errors_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BPR pairwise loss over gathered rows, fused with the squared norms the regularisers need.
 * --------------------------------------------------------------------------------------------- */
#define GCR_BPR_NCL 0         /* -log(1e-5 + sigmoid(x))  ncl.py:116-120 (literal 10e-6), mhcn.py:35-39 */
#define GCR_BPR_LOGSIGMOID 1  /* -logsigmoid(x)           gcl.py:221, sept.py:34-38 */
#define GCR_BPR_LOG_SIGMOID 2 /* -log(sigmoid(x))         lightgcn.py:108 */

/* floats of workspace gcr_bpr_fwd_f32 needs for `batch` samples */
int64_t gcr_bpr_workspace_floats(int64_t batch);

/*
 * x_b = <U[u_b], I[i_b]> - mean_k <U[u_b], I[j_{b,k}]>          (n_neg negatives per sample)
 * replaces  rec_user_emb[user_idx] / rec_item_emb[pos_idx] / [neg_idx] + bpr_loss  ncl.py:314-317,
 *           lightgcn.py:95-108, gcl.py:216-221, sept.py:236-240, mhcn.py:527-530.
 * Outputs: dloss_dx[b] = d loss_b / d x_b (saved for backward; NaN for a skipped sample); sums[5] =
 *   { sum_b loss_b, sum_b |U[u_b]|^2, sum_b |I[i_b]|^2, sum_{b,k} |I[j_bk]|^2, #samples skipped
 *     because an id was out of range } — the caller forms mean / regulariser variants
 *   (lightgcn.py:118, gcl.py:222, ncl.py:122-123, sept.py:241) from these.
 * Deterministic (fixed-order block reduction).  j_idx is [batch * n_neg] row-major.
 */
int32_t gcr_bpr_fwd_f32(const float* user_tab, const float* item_tab, int32_t d,
                        const int64_t* u_idx, const int64_t* i_idx, const int64_t* j_idx,
                        int64_t batch, int32_t n_neg, int32_t variant, int64_t n_users, int64_t n_items,
                        float* dloss_dx, float* sums, float* workspace, void* stream);

/*
 * Backward: grad_sums (device, 4 floats) is dL/d sums[0..3] of the forward; the row gradients are
 * accumulated (float atomics, duplicate ids add up) into the dense tables grad_user [n_users, d] /
 * grad_item [n_items, d], which the caller zeroes or already holds other gradient terms in.
 */
int32_t gcr_bpr_bwd_f32(const float* user_tab, const float* item_tab, int32_t d,
                        const int64_t* u_idx, const int64_t* i_idx, const int64_t* j_idx,
                        int64_t batch, int32_t n_neg, int64_t n_users, int64_t n_items,
                        const float* dloss_dx, const float* grad_sums,
                        float* grad_user, float* grad_item, void* stream);

/*
 * Backward for large batches (lightgcn.py:95-118 trains on all E edges at once): the same gradients from
 * three SORTED orders of the samples instead of three row atomics per sample.  gcr_sort_index produces an
 * order: keys_sorted[k] = idx[perm[k]] ascending (stable), ids outside [0, n_keys) get the key n_keys and
 * sort last; n and n_keys < 2^31.  (keys_u, perm_u) = order of u_idx, (keys_i, perm_i) of i_idx,
 * (keys_j, perm_j) of the flattened j_idx [batch * n_neg]; orders of index arrays that do not change
 * between steps can be reused.  One row atomic per run of equal keys and 64-entry chunk.
 * keys_i == perm_i == NULL (i_idx may then be NULL too): the POSITIVE-pair parts are left to the caller — grad_user gets
 * only - coef * mean_k I[j_bk] (+ the |U[u]|^2 term), grad_item only the negatives' rows (+ the |I[j]|^2 term).  For a
 * batch that is the training graph's own edge list those parts are one SpMM with per-edge coefficients on the graph's
 * structure (functional.bpr_edge_sums: gcr_spmm_csr_f32 with val = dL/dx per edge), not two sorted scatters.
 */
/* keys_u == perm_u == NULL (only together with keys_i == NULL): the users' rows are left to the caller as well — for the
 * graph's own edge list in user-major order the negatives form a CSR block on the graph's user row pointer:
 * gcr_bpr_neg_block_f32 writes its columns (int32 [batch * n_neg], dropped samples -> row 0 with value 0), its values
 * - g dL/dx / n_neg and the dropped samples per user; one gcr_spmm_csr_f32 launch on that block adds
 * - g dl_e mean_k I[j_ek] to grad_user.  Only the negatives' item rows remain a sorted scatter. */
int32_t gcr_bpr_neg_block_f32(const float* dloss_dx, const int64_t* j_idx, const int64_t* u_idx, int64_t batch, int32_t n_neg,
                              int64_t n_items, const float* grad_sums, int32_t* col, float* val, float* dropped_per_user,
                              uint32_t* sort_key, uint64_t* sort_payload, void* stream);
/* ... and the negatives' ITEM rows from a sort that carries its payload (lightgcn.py:91-93 draws fresh negatives every
 * step, so this sort cannot be cached): with sort_key / sort_payload non-NULL gcr_bpr_neg_block_f32 also writes, per
 * negative slot, key = j (n_items for a dropped slot: it sorts last) and payload = (u << 32) | bits(- g dL/dx / n_neg);
 * gcr_sort_pairs_u64 orders both by key (stable radix sort over the bits n_keys needs; n < 2^31);
 * gcr_bpr_neg_items_sorted_f32 walks the sorted arrays in 64-entry chunks and adds, per run of equal keys,
 * sum coef U[u] + 2 g_3 (#entries) I[j] to grad_item[j] with one row atomic — key, user and coefficient arrive by coalesced
 * loads instead of perm -> sample -> (u_idx, dloss_dx) as in gcr_bpr_bwd_sorted_f32's third launch. */
/* (col, val, dropped_per_user may be NULL together when only the sort key / payload are wanted — any batch of triples, not
 * only a graph's edge list; gcr_bpr_bwd_sorted_f32 with keys_j == perm_j == NULL then leaves the negatives' item rows to
 * gcr_bpr_neg_items_sorted_f32.) */
int64_t gcr_sort_pairs_u64_workspace_bytes(int64_t n);
int32_t gcr_sort_pairs_u64(const uint32_t* keys, const uint64_t* payload, int64_t n, int64_t n_keys, uint32_t* keys_sorted,
                           uint64_t* payload_sorted, void* workspace, void* stream);
int32_t gcr_bpr_neg_items_sorted_f32(const float* user_tab, const float* item_tab, int32_t d, const uint32_t* keys_sorted,
                                     const uint64_t* payload_sorted, int64_t n_entries, int64_t n_users, int64_t n_items,
                                     const float* grad_sums, float* grad_item, void* stream);
int64_t gcr_sort_index_workspace_bytes(int64_t n);
int32_t gcr_sort_index(const int64_t* idx, int64_t n, int64_t n_keys, uint32_t* keys_sorted, int32_t* perm,
                       void* workspace, void* stream);
int32_t gcr_bpr_bwd_sorted_f32(const float* user_tab, const float* item_tab, int32_t d,
                               const int64_t* u_idx, const int64_t* i_idx, const int64_t* j_idx,
                               int64_t batch, int32_t n_neg, int64_t n_users, int64_t n_items,
                               const float* dloss_dx, const float* grad_sums,
                               const uint32_t* keys_u, const int32_t* perm_u,
                               const uint32_t* keys_i, const int32_t* perm_i,
                               const uint32_t* keys_j, const int32_t* perm_j,
                               float* grad_user, float* grad_item, void* stream);

/*
 * The per-non-zero values of that coefficient operator, for a symmetric bipartite CSR with the users first (2 * n_pairs
 * non-zeros, the user rows' n_pairs first): val[e] = grad_sums[0] * dloss_dx[pair(e)], pair(e) = e in the user-major half,
 * mirror[e] (the position of the transposed non-zero, CsrGraph.mirror_perm) in the item-major half; a NaN in dloss_dx (a
 * sample the forward dropped) gives 0 and adds 1 to dropped_per_item[its positive item] (zero on entry).
 */
int32_t gcr_bpr_edge_values_f32(const float* dloss_dx, const int64_t* mirror, const int32_t* col, int64_t n_users,
                                int64_t n_pairs, const float* grad_sums, float* val, float* dropped_per_item,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Counter-based RNG (Philox-4x32-10): negative sampler and edge-dropout bitmaps.
 * --------------------------------------------------------------------------------------------- */
/*
 * out[b * n_negs + k] = first of the draws  mulhi(philox(ctr = (slot, trial/4, 'NEGS'), key = seed)
 * [trial % 4], num_items),  slot = offset + b * n_negs + k,  that is not in the user's sorted
 * training row user_items_sorted[user_rowptr[u] .. user_rowptr[u+1]);  max_trials == 0 disables
 * rejection (lightgcn.py:91-94); no acceptable draw within max_trials -> -1 (ncl.py:110-112).
 * replaces  next_batch_pairwise  ncl.py:91-114, gcl.py:111-125, ssl4rec.py:33-50, sept.py:12-30.
 */
int32_t gcr_neg_sample(const int64_t* user_rowptr, const int32_t* user_items_sorted, const int64_t* u_idx,
                       int64_t batch, int32_t n_negs, int64_t n_users, int64_t num_items,
                       uint64_t seed, uint64_t offset, int32_t max_trials, int64_t* out, void* stream);

/*
 * Bernoulli keep bitmap over nnz edges: bit e = (u_e >= pe), u_e the 24-bit uniform of word id%4 of
 * philox(ctr = (id/4, 0, 'EDGE'), key = seed), id = edge_id[e] (or e when edge_id == NULL).
 * Passing the canonical id of every non-zero lets A and A^T carry the same mask without a gather.
 * replaces  EdgeRemoving.__call__  gcl.py:22-25 (`rand >= pe`), buir.py:300-309 (mask part).
 * bits: uint32[(nnz + 31) / 32], unused high bits cleared.
 */
int32_t gcr_edge_mask_bits(int64_t nnz, float pe, uint64_t seed, const int64_t* edge_id, uint32_t* bits,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * InfoNCE / prototype contrast: all-pairs logits on the matrix cores, fused 1/tau scale + row
 * logsumexp; the M x N score matrix is never materialised.  d must be 32, 64, 128 or 256
 * (GCR_EUNSUPPORTED otherwise; the host wrapper zero-pads other widths).  f32 in, f32 out, f32
 * accuracy on both engines: the f32 MFMA (v_mfma_f32_32x32x2_f32), or for d <= 128 the bf16 MFMA on
 * three error-free bf16 planes per operand (csrc/gcr_infonce.hip, "split-operand engine").
 * --------------------------------------------------------------------------------------------- */
/* engine the InfoNCE / k-means kernels use for width d by default: 0 = f32 MFMA, 1 = split-operand bf16
 * (d <= 128); GCR_INFONCE_ENGINE_F32 in the flags of the _ex entry points selects 0 for one call */
int32_t gcr_infonce_engine(int32_t d);


/* out[r] = 1 / max(||x_r||_2, eps): the denominator of F.normalize (ncl.py:127, gcl.py:29-30). */
int32_t gcr_row_inv_norm_f32(const float* x, int64_t n, int32_t d, float eps, float* out, void* stream);

/* bytes of workspace gcr_infonce_fwd_f32 needs */
int64_t gcr_infonce_fwd_workspace_bytes(int64_t m, int64_t n, int32_t d);

/*
 * lse[i] = log sum_j exp( inv_tau * a_scale[i] * b_scale[j] * <a_i, b_j> ),  i < m, j < n.
 * a_scale / b_scale are optional per-row multipliers (the inverse norms when b_cos / normalize).
 * replaces  F.log_softmax(view1 @ view2.T / t, dim=1)           ncl.py:128-129, ssl4rec.py:22-23
 *           F.cross_entropy(sim, labels) / (sim.T, labels)       gcl.py:31-34
 *           torch.exp(torch.matmul(norm_cu, F.normalize(iu).T) / t).sum(1)   ncl.py:363-366
 *           torch.exp(user_emb @ item_emb.T / t).sum(dim=1)      ssl4rec.py:29
 *
 * col_sum (optional, [n]): in the SAME pass also col_sum[j] = sum_i exp(s_ij - col_bound) over all
 * anchors (float atomics), from which the column logsumexp of gcl.py:34 (`cross_entropy(sim.T)`)
 * is col_bound + log(col_sum[j]).  col_bound must bound every logit from above (unit-norm rows:
 * inv_tau) and keep exp(-2 col_bound) representable; pass NULL / 0 otherwise and run a second
 * call with a and b swapped.
 */
int32_t gcr_infonce_fwd_f32(const float* a, const float* a_scale, int64_t m,
                            const float* b, const float* b_scale, int64_t n, int32_t d,
                            float inv_tau, float* lse, float* col_sum, float col_bound,
                            void* workspace, void* stream);

/*
 * Same with flags.  GCR_INFONCE_EXCLUDE_DIAGONAL: the pair (i, j = i) is left out of every sum, i.e.
 * lse[i] = log sum_{j != i} exp(s_ij) — the intra-view negatives of PyGCL's DualBranchContrast
 * (univariate/grace.py:396-404: `neg_mask = 1 - eye` over the anchors themselves), a and b then being the
 * same view.  Not combinable with col_sum.
 */
#define GCR_INFONCE_EXCLUDE_DIAGONAL 1u
/* GCR_INFONCE_ENGINE_F32: run this problem on the f32 MFMA engine instead of the default (the split-operand
 * bf16 engine for d <= 128).  The engine is an argument, never read from the environment; forward and
 * backward of one problem must pass the same flag (the backward recomputes the forward's logits). */
#define GCR_INFONCE_ENGINE_F32 4u
/* GCR_INFONCE_UNIT_ROWS: the caller promises |a_scale[i] * a_i|_2 <= 1 and |b_scale[j] * b_j|_2 <= 1 — rows normalised
 * by the scales, as every contrast loss of the reference does (F.normalize, ncl.py:127, gcl.py:29-30).  With the
 * operand range known, the split-operand launches (gcr_infonce_fwd_ex_f32, gcr_infonce_fwd_o_f32,
 * gcr_infonce_bwd_ex_f32) of d <= 64 — the two-product launches gcr_infonce_fwd_o_f32 / gcr_infonce_bwd_ex_f32 also of
 * d = 128 — and inv_tau <= 20 run on TWO f16 planes per operand and three product terms instead of three bf16 planes and
 * six (csrc/gcr_infonce.hip, EngH2): half the matrix-core work, operands rounded at 2^-22 instead of 2^-27 — a few f32
 * roundings per product, inside the 1e-5 of every parity test.  Without the promise nothing changes. */
#define GCR_INFONCE_UNIT_ROWS 2u
int32_t gcr_infonce_fwd_ex_f32(const float* a, const float* a_scale, int64_t m,
                               const float* b, const float* b_scale, int64_t n, int32_t d,
                               float inv_tau, float* lse, float* col_sum, float col_bound,
                               void* workspace, uint32_t flags, void* stream);

/*
 * Forward WITH the softmax-weighted row sum (flash-attention forward), one pass:
 *   lse[i] = log sum_j exp(s_ij),    o[i, :] = sum_j exp(s_ij - lse[i]) * (b_scale[j] * b_j)      o: [m, d]
 * For the row-softmax losses (ncl.py:125-130 InfoNCE, ncl.py:358-375 ssl_layer_loss / ProtoNCE_loss,
 * ssl4rec.py:25-30) the gradient w.r.t. the scaled anchor row is  dL/dlse[i] * inv_tau * o[i, :]  (plus the
 * positive-logit term), so training needs no backward pass over the M x N tile for the anchor side: the
 * backward (gcr_infonce_bwd_f32 with the table stationary) recomputes the score tile once, not twice.
 * Split-operand engine only: d in {32, 64, 128} and no GCR_INFONCE_ENGINE_F32 (gcr_infonce_fwd_o_supported
 * tells; GCR_EUNSUPPORTED otherwise — the caller then uses gcr_infonce_fwd_ex_f32 + two backward calls).
 * flags: GCR_INFONCE_EXCLUDE_DIAGONAL as above.
 */
int32_t gcr_infonce_fwd_o_supported(int32_t d, uint32_t flags);
int64_t gcr_infonce_fwd_o_workspace_bytes(int64_t m, int64_t n, int32_t d);
int32_t gcr_infonce_fwd_o_f32(const float* a, const float* a_scale, int64_t m,
                              const float* b, const float* b_scale, int64_t n, int32_t d,
                              float inv_tau, float* lse, float* o, void* workspace, uint32_t flags, void* stream);

/*
 * out[i] = scale * a_scale[i] * b_scale[p] * <a_i, b_p>, p = pos[i] (pos == NULL: p = i) — the
 * positive logit (the diagonal of ncl.py:129 / gcl.py:32-34, `(norm_cu * norm_iu).sum(1) / t`
 * ncl.py:363).  An out-of-range pos yields NaN for that row.  Any d.
 */
int32_t gcr_pos_logit_f32(const float* a, const float* a_scale, const float* b, const float* b_scale,
                          const int64_t* pos, int64_t m, int64_t n, int32_t d, float scale, float* out,
                          void* stream);

/* bytes of workspace gcr_infonce_bwd_f32 needs (0 when no column split is used) */
int64_t gcr_infonce_bwd_workspace_bytes(int64_t mx, int64_t ny, int32_t d);

/*
 * Flash-style backward of the softmax part (autograd of the expressions listed at
 * gcr_infonce_fwd_f32).  With s_ij = inv_tau * <xhat_i, yhat_j>, xhat = x * x_scale, yhat = y * y_scale:
 *   P_ij   = w_x[i] * exp(s_ij - lse_x[i]) + w_y[j] * exp(s_ij - lse_y[j])
 *   g[i,:] = inv_tau * sum_j P_ij * yhat_j          (gradient w.r.t. xhat_i; the positive-logit
 *                                                    term is gcr_infonce_pos_bwd_f32)
 * (w_x, lse_x) is the row-softmax side (upstream dL/dlse of the rows of x, their forward lse);
 * (w_y, lse_y) the column side (dL/dlse of the rows of y).  Either pair may be NULL.  Call twice
 * with x and y swapped for both input gradients.  Scores are recomputed, never stored.
 */
int32_t gcr_infonce_bwd_f32(const float* x, const float* x_scale, int64_t mx,
                            const float* y, const float* y_scale, int64_t ny, int32_t d, float inv_tau,
                            const float* lse_x, const float* w_x, const float* lse_y, const float* w_y,
                            float* g, void* workspace, void* stream);
/* Same with flags: GCR_INFONCE_EXCLUDE_DIAGONAL gives P_ii = 0 (backward of gcr_infonce_fwd_ex_f32). */
int32_t gcr_infonce_bwd_ex_f32(const float* x, const float* x_scale, int64_t mx,
                               const float* y, const float* y_scale, int64_t ny, int32_t d, float inv_tau,
                               const float* lse_x, const float* w_x, const float* lse_y, const float* w_y,
                               float* g, void* workspace, uint32_t flags, void* stream);

/*
 * Positive-logit term: gx[i,:] += coef[i] * inv_tau * yhat[p_i,:]  (plain add, row i is exclusive)
 *                      gy[p_i,:] += coef[i] * inv_tau * xhat[i,:]  (float atomics, p_i may repeat)
 * p_i = pos[i] (pos == NULL: p_i = i); gx or gy may be NULL.  Any d.
 */
int32_t gcr_infonce_pos_bwd_f32(const float* x, const float* x_scale, const float* y, const float* y_scale,
                                const int64_t* pos, const float* coef, int64_t mx, int64_t ny, int32_t d,
                                float inv_tau, float* gx, float* gy, void* stream);

/* Backward through F.normalize: out = inv_norm * (ghat - xhat <xhat, ghat>), xhat = x * inv_norm;
 * out may alias ghat.  Any d. */
int32_t gcr_normalize_bwd_f32(const float* x, const float* inv_norm, const float* ghat, int64_t n, int32_t d,
                              float* out, void* stream);

/* Backward through `F.normalize(A x, p=2, dim=1)` from the SAVED NORMALISED rows n = normalize(z) and inv = 1 / max(|z|, eps)
 * (what gcr_spmm_csr_f32 / _dual_f32 write with GCR_SPMM_ROW_L2NORM; sept.py:223-224, mhcn.py:440-457):
 *   out = (g_n - n <n, g_n>) * inv + g_raw          g_raw optional (the gradient of the raw product, mhcn.py:440-442)
 * one pass; out may alias g_n or g_raw.  d a multiple of 4, <= 256. */
int32_t gcr_normalize_bwd_n_f32(const float* n_rows_normalised, const float* inv_norm, const float* g_n,
                                const float* g_raw, int64_t rows, int32_t d, float* out, void* stream);
/* The same from the RAW rows z = A x (a forward that kept only z and 1 / |z|: gcr_spmm_csr_dual_acc_f32 without y_norm):
 * n = z * inv_norm on the fly. */
int32_t gcr_normalize_bwd_raw_f32(const float* z_raw, const float* inv_norm, const float* g_n, const float* g_raw,
                                  int64_t rows, int32_t d, float* out, void* stream);

/* out[dx, dg] = X^T G for tall row-major X [n, dx], G [n, dg]: the weight gradient of `em @ W` in MHCN's gating /
 * attention (mhcn.py:404-420; W is d x d, n = #users), rows split over the chip on the f32 MFMA, partials summed in a
 * fixed order (bitwise reproducible).  dx, dg in {32, 64, 96, 128} (GCR_EUNSUPPORTED otherwise). */
int64_t gcr_gram_tn_workspace_bytes(int64_t n, int32_t dx, int32_t dg);
int32_t gcr_gram_tn_f32(const float* x, const float* g, int64_t n, int32_t dx, int32_t dg, float* out, void* workspace,
                        void* stream);

/* out[r] = <x[r, :], v> for row-major x [n, d], v [d] — the channel-attention logits of mhcn.py:414 as one product per channel. */
int32_t gcr_rows_dot_vec_f32(const float* x, const float* v, int64_t n, int32_t d, float* out, void* stream);

/* out[c] = sum_r w[r] * x[r][c] for tall x [n, d], w [n] (d <= 256): the gradient of v in the channel-attention logits
 * `em @ v` (mhcn.py:414 with v = attention_mat attention^T); fixed-order partial sums. */
int64_t gcr_weighted_colsum_workspace_bytes(int64_t n, int32_t d);
int32_t gcr_weighted_colsum_f32(const float* x, const float* w, int64_t n, int32_t d, float* out, void* workspace,
                                void* stream);

/* MHCN's gates around the library GEMM z = em W (univariate/mhcn.py:404-411 `self_gating`, `self_supervised_gating`):
 *   forward   out = em * sigmoid(z + bias)                                   bias [d] or NULL
 *   backward  d_em = g * sig,  d_z = g * em * sig * (1 - sig),  d_bias[c] = sum_r d_z[r][c]   (sig recomputed)
 * one pass each over row-major [n, d] arrays, d <= 256; d_bias from fixed-order partial sums. */
int32_t gcr_gate_fwd_f32(const float* em, const float* z, const float* bias, int64_t n, int32_t d, float* out, void* stream);
int64_t gcr_gate_bwd_workspace_bytes(int64_t n, int32_t d);
int32_t gcr_gate_bwd_f32(const float* g, const float* em, const float* z, const float* bias, int64_t n, int32_t d,
                         float* d_em, float* d_z, float* d_bias, void* workspace, void* stream);

/* MHCN's channel attention (univariate/mhcn.py:413-420) over three channel tables e_k [n, d] with the logit vector
 * v = attention_mat attention^T [d]:  score[k][r] = softmax_k <e_k[r], v>,  mixed[r] = sum_k score[k][r] e_k[r]
 * (+ extra_scale * extra[r], the `+ simple / 2` of mhcn.py:443; extra may be NULL).  score is [3, n].
 * Backward for g = d mixed:  d_e_k = score_k g + q_k v with q_k = score_k (<g, e_k> - sum_m score_m <g, e_m>),
 * d_extra = extra_scale g (NULL: not written), d_v[c] = sum_r sum_k q_k[r] e_k[r][c] (fixed-order partial sums).
 * d in {32, 64, 128, 256} (gcr_channel_mix_supported; GCR_EUNSUPPORTED otherwise). */
int32_t gcr_channel_mix_supported(int32_t d);
int32_t gcr_channel_mix_fwd_f32(const float* e1, const float* e2, const float* e3, const float* v, const float* extra,
                                float extra_scale, int64_t n, int32_t d, float* mixed, float* score, void* stream);
int64_t gcr_channel_mix_bwd_workspace_bytes(int64_t n, int32_t d);
int32_t gcr_channel_mix_bwd_f32(const float* g, const float* e1, const float* e2, const float* e3, const float* v,
                                const float* score, float extra_scale, int64_t n, int32_t d, float* d_e1, float* d_e2,
                                float* d_e3, float* d_extra, float* d_v, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BCE-with-logits over the all-pairs score matrix — the `loss_type == "bce"` branch of LightGCN's training step:
 *   scores = torch.matmul(user_vecs, item_emb.t()); labels = one-hot at pos_i;                 lightgcn.py:110-112
 *   loss = F.binary_cross_entropy_with_logits(scores, labels)                                  lightgcn.py:113
 *        = ( sum_ij softplus(s_ij) - sum_i s_{i, pos_i} ) / (M N),   s_ij = <a_i, b_j>  (no temperature, rows NOT normalised).
 * The M x N part never reaches memory: row sums of softplus in the forward, sigmoid-weighted operand sums in the
 * backward, on the InfoNCE tile engine (three bf16 planes for d <= 64; the f32 MFMA for d = 128 / 256 or with
 * GCR_INFONCE_ENGINE_F32 in the flags).  d in {32, 64, 128, 256}.  The O(M d) positive-logit term is
 * gcr_pos_logit_f32 / gcr_infonce_pos_bwd_f32 (or one SpMM when the batch is the graph's edge list).
 * --------------------------------------------------------------------------------------------- */
/* GCR_BCE_TWO_PLANES (flags of gcr_bce_fwd_f32 / gcr_bce_bwd_f32, d <= 64, not with GCR_INFONCE_ENGINE_F32): run the loops
 * on TWO f16 planes and three product terms instead of three bf16 planes and six.  The rows are not unit rows, so a pre-pass
 * takes every row's norm, the loops see unit rows (operands rounded at 2^-22 of the row's norm) and every score is
 * un-scaled by its two norms before the softplus / sigmoid; the second product carries norm x weight of the streamed row,
 * shifted into f16 range by a power of two, and sigmoid against a running power-of-two reference per output row (rescaled
 * like a flash forward), so that a row whose sigmoids are all tiny keeps its digits.  Half the matrix-core work; |score
 * error| a few 2^-24 |a_i| |b_j| — the f32 dot product's own class; terms below 2^-39 of (largest sigmoid of the row) x
 * (largest norm x weight) are flushed.  gcr_bce_bwd_f32 with the weights on the stationary rows (w_x) ignores the flag. */
#define GCR_BCE_TWO_PLANES 8u
/* 1 when gcr_bce_fwd_f32 can also return o (d <= 64 on the split-operand engine) */
int32_t gcr_bce_fwd_o_supported(int32_t d, uint32_t flags);
int64_t gcr_bce_fwd_workspace_bytes(int64_t m, int64_t n, int32_t d);
/*
 * row_softplus[i] = sum_j softplus(<a_i, b_j>)                                   [m]
 * o[i, :]         = sum_j sigmoid(<a_i, b_j>) * b_j      (optional, NULL to skip)  [m, d]
 * — with o the gradient w.r.t. a_i of sum_i c_i row_softplus[i] is c_i * o[i, :]: no backward launch for that side.
 */
int32_t gcr_bce_fwd_f32(const float* a, int64_t m, const float* b, int64_t n, int32_t d, float* row_softplus,
                        float* o, void* workspace, uint32_t flags, void* stream);
int64_t gcr_bce_bwd_workspace_bytes(int64_t mx, int64_t ny, int32_t d);
/*
 * g[i, :] = sum_j (w_x[i] + w_y[j]) * sigmoid(<x_i, y_j>) * y_j — the gradient of sum over pairs of w * softplus(s)
 * w.r.t. x_i, the weight sitting on the stationary rows (w_x, [mx]) or on the streamed rows (w_y, [ny]): exactly one
 * of the two is given, the other NULL.  Scores are recomputed, never stored.
 */
int32_t gcr_bce_bwd_f32(const float* x, int64_t mx, const float* y, int64_t ny, int32_t d, const float* w_x,
                        const float* w_y, float* g, void* workspace, uint32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------------
 * k-means E-step of NCL's prototype contrast (ncl.py:340-356: faiss.Kmeans(d, k).train(x) +
 * index.search(x, 1); faiss itself is an un-vendored dependency, not installed: what is restated is its published
 * Clustering::train loop — Lloyd iterations + split_clusters for empty clusters; PARITY WITH FAISS UNPINNED).
 * --------------------------------------------------------------------------------------------- */
/* assign[i] = argmin_c ||x_i - centroids_c||^2 (ties -> smaller c) on the MFMA tile engine;
 * half_sqnorm[c] = 0.5 ||centroids_c||^2 (kept up to date by gcr_kmeans_update_f32);
 * best_score (optional) = <x_i, c> - 0.5 ||c||^2 of the winner.  d in {32, 64, 128, 256}. */
int32_t gcr_kmeans_assign_f32(const float* x, int64_t n, const float* centroids, const float* half_sqnorm,
                              int64_t k, int32_t d, int64_t* assign, float* best_score, void* stream);
/* The search fused with the first half of the Lloyd update: besides assign[i] (optional, may be NULL) every row x_i is
 * added to sums[copy, assign[i], :] and counts[copy, assign[i]] += 1 (float-atomic 256-B rows; copy = workgroup %
 * n_copies; sums [n_copies, k, d] / counts [n_copies, k] zeroed by the caller or by the previous
 * gcr_kmeans_lloyd_update_f32) — one launch instead of search + accumulate.  d in {32, 64, 128}; GCR_EUNSUPPORTED
 * otherwise (use gcr_kmeans_assign_f32 + the `assign` form of gcr_kmeans_lloyd_update_f32). */
int32_t gcr_kmeans_assign_accumulate_f32(const float* x, int64_t n, const float* centroids, const float* half_sqnorm,
                                         int64_t k, int32_t d, int64_t* assign, float* sums, float* counts,
                                         int32_t n_copies, void* stream);
#define GCR_KMEANS_SEARCH_LOW_REGISTERS 1u
/* INCREMENTAL centroid update in 64-bit fixed point (exact, order-independent, run-to-run reproducible): the search of
 * gcr_kmeans_search_image_f32 followed, for every point whose nearest centroid CHANGED since prev_assign (int32 [n], -1 =
 * none yet; updated in place), by  sums_q[new] += q, sums_q[old] -= q, counts[new] += 1, counts[old] -= 1  with
 * q = round(x * qscale[0]) (qscale = {2^e, 2^-e}, device floats, |x| * 2^e < 2^30).  sums_q int64 [n_copies, k, d] / counts
 * int32 [n_copies, k] persist across the iterations of one k-means (zero before the first); a workgroup adds into private copy
 * (its index % n_copies) and the copies are summed — exactly — by the update.  gcr_kmeans_lloyd_update_q_f32 then sets
 * centroid = sums_q / (2^e * count) (an empty cluster keeps its centroid), 0.5 |c|^2, and runs the split step of
 * gcr_kmeans_lloyd_update_f32 on a float copy of the counts (counts_scratch [k], left zeroed); `image` (optional): the
 * operand image of the search (gcr_kmeans_centroid_image_f32 built it once) is kept up to date in the same two launches —
 * every row by the finalize kernel, the rows a split rewrote by the split kernel — so an iteration needs no image launch. */
int32_t gcr_kmeans_search_image_incr_f32(const float* x, int64_t n, const void* image, int64_t k, int32_t d,
                                         int32_t* prev_assign, const float* qscale, int64_t* sums_q, int32_t* counts,
                                         int32_t n_copies, uint32_t flags, void* stream);
int32_t gcr_kmeans_lloyd_update_q_f32(const int64_t* sums_q, const int32_t* counts, const float* qscale, int64_t k, int32_t d,
                                      float* centroids, float* half_sqnorm, float* counts_scratch, int64_t n_points,
                                      uint64_t seed, int32_t iter, int32_t* n_split, void* image, int32_t n_copies, void* stream);

/* The search of a Lloyd iteration for small problems (NCL's e_step: 76.8 K sampled points x a few hundred centroids, 25
 * iterations per table and training step — a latency problem, not a throughput one): the centroids are split into their
 * three bf16 planes ONCE per iteration into an image in MFMA-fragment order (gcr_kmeans_centroid_image_f32;
 * gcr_kmeans_image_bytes bytes), and gcr_kmeans_search_image_f32 runs every wave on its own — 32 points stationary in
 * registers, fragments as coalesced loads from L2, no LDS, no barrier.  assign (optional) as gcr_kmeans_assign_f32; sums /
 * counts (optional, n_copies private copies) as gcr_kmeans_assign_accumulate_f32.  d in {32, 64}; same arithmetic, same
 * tie rule (smaller centroid id) as the tiled kernels: identical assignments. */
int64_t gcr_kmeans_image_bytes(int64_t k, int32_t d);
int32_t gcr_kmeans_centroid_image_f32(const float* centroids, const float* half_sqnorm, int64_t k, int32_t d, void* image,
                                      void* stream);
/* GCR_KMEANS_SEARCH_LOW_REGISTERS: the <= 128-register form (four waves per SIMD, fragments fetched one k-chunk ahead): its
 * waves fit beside the InfoNCE loops of the same training step on a SIMD, so the e_step overlaps them instead of queueing. */
int32_t gcr_kmeans_search_image_f32(const float* x, int64_t n, const void* image, int64_t k, int32_t d, int64_t* assign,
                                    float* sums, float* counts, int32_t n_copies, uint32_t flags, void* stream);

/* centroids_c <- mean of the rows assigned to c (empty clusters keep their centroid), and
 * half_sqnorm refreshed.  n == 0 only refreshes half_sqnorm (initialisation).
 * sums [k, d] / counts [k] are fp32 scratch. */
int32_t gcr_kmeans_update_f32(const float* x, int64_t n, int32_t d, const int64_t* assign, int64_t k,
                              float* centroids, float* half_sqnorm, float* sums, float* counts, void* stream);
/* The same update from the points ordered by cluster — (keys_sorted, perm) = gcr_sort_index(assign, n, k): one
 * row atomic per run of equal cluster ids and 64-entry chunk instead of one per point (large n). */
int32_t gcr_kmeans_update_sorted_f32(const float* x, int64_t n, int32_t d, const uint32_t* keys_sorted,
                                     const int32_t* perm, int64_t k, float* centroids, float* half_sqnorm,
                                     float* sums, float* counts, void* stream);

/*
 * One Lloyd update without host involvement (a fixed launch sequence: hipGraph-capturable):
 *   1. centroids_c <- mean of the rows assigned to c (empty clusters keep their centroid) — from `assign` (one row
 *      atomic per point), or, when (keys_sorted, perm) = gcr_sort_index(assign, n, k) is given, from the ordered points,
 *      or — assign == keys_sorted == NULL — from sums / counts as gcr_kmeans_assign_accumulate_f32 left them;
 *   2. faiss Clustering.cpp `split_clusters` on the device: every empty cluster takes over a copy of the centroid of a
 *      cluster cj accepted with probability (size_cj - 1) / (n - k) while walking cj = 0, 1, ...; the two copies are
 *      perturbed by (1 +- 1/1024) alternating over the dimensions, the sizes split in half.  Trial q of the e-th empty
 *      cluster uses word x of Philox-4x32-10(counter (q, e, iter, 'KMSP'), key seed); after 64 k misses the largest
 *      cluster (>= 2 points) is split instead;
 *   3. half_sqnorm refreshed.
 * sums [n_copies, k, d] / counts [n_copies, k] must be ZERO on entry and are zero again on exit (no memsets between
 * iterations).  n_copies (1 .. 64; used by the `assign` form only): workgroup b adds into private copy b % n_copies and
 * the copies are summed in a fixed order — with a few hundred clusters and ~256 points each, one copy would serialise
 * the memory-side row atomics on a few hundred rows.
 * n_split (optional device int32) is incremented by the number of re-seeded clusters.  d in {32, 64, 128, 256}.
 */
int32_t gcr_kmeans_lloyd_update_f32(const float* x, int64_t n, int32_t d, const int64_t* assign,
                                    const uint32_t* keys_sorted, const int32_t* perm, int64_t k,
                                    float* centroids, float* half_sqnorm, float* sums, float* counts, int32_t n_copies,
                                    uint64_t seed, int32_t iter, int32_t* n_split, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Full-ranking evaluation: user x ALL-items scores, training positives masked, exact top-N.
 * --------------------------------------------------------------------------------------------- */
/* scores[q, j] = <user_emb[user_ids[q]], item_emb[j]>  (user_ids == NULL: q itself), fp32 [n_query,
 * n_items] row-major.  replaces torch.matmul(user_emb, item_emb.t()) lightgcn.py:50, gcl.py:88,
 * `torch.matmul(self.user_emb[u], self.item_emb.T)` ncl.py:394.  d in {32, 64, 128, 256}. */
int32_t gcr_score_rows_f32(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                           const float* item_emb, int64_t n_items, int32_t d, float* scores, void* stream);
/* Per query row: scores[q, training items of the user] = -inf (in place), then the k largest scores
 * in descending order, ties -> smaller item id; rows with fewer than k finite items are padded with
 * (-1, -inf).  replaces `scores_user[list(known_pos)] = -np.inf; np.argsort(-scores_user)[:k]`
 * lightgcn.py:55-57, gcl.py:93-96 and `candidates[...] = -1e8; torch.topk(candidates, max_N)`
 * ncl.py:257-261.  user_rowptr / user_items_sorted (CSR of training positives) may both be NULL.
 * k <= 256. */
int32_t gcr_topk_masked_f32(float* scores, int64_t n_query, int64_t n_items, const int64_t* user_ids,
                            int64_t n_users, const int64_t* user_rowptr, const int32_t* user_items_sorted,
                            int32_t k, int64_t* top_items, float* top_scores, void* stream);

/*
 * Fused full ranking: the same top-k as gcr_score_rows_f32 + gcr_topk_masked_f32 without ever writing the
 * [n_query, n_items] score matrix.  Pass 0 scores the first 4096 items densely and takes, per user, the k-th
 * largest eligible score as a threshold; pass 1 recomputes the score tiles over all items (split-operand
 * bf16 MFMA, f32-accurate) and keeps only (score >= threshold, item not a training positive) candidates —
 * about k * n_items / 4096 per user; a per-user bitonic sort finishes.  status[q] = 1 marks a user whose
 * candidate list overflowed or came up short (re-rank those through the two-call path); 0 = row q is final.
 * Supported (gcr_rank_fused_supported): d in {32, 64, 128}, n_items >= 16384, k <= 256.
 * replaces lightgcn.py:48-57, gcl.py:87-96, ncl.py:253-264,390-394 (as gcr_topk_masked_f32).
 */
int32_t gcr_rank_fused_supported(int64_t n_items, int32_t d, int32_t k);
int64_t gcr_rank_fused_workspace_bytes(int64_t n_query);
int32_t gcr_rank_fused_f32(const float* user_emb, const int64_t* user_ids, int64_t n_query, int64_t n_users,
                           const float* item_emb, int64_t n_items, int32_t d, const int64_t* user_rowptr,
                           const int32_t* user_items_sorted, int32_t k, int64_t* top_items, float* top_scores,
                           int32_t* status, void* workspace, void* stream);

/*
 * Per-user terms of the ranking metrics (ncl.py:133-177 `Metric.hits` / `Metric.NDCG`, the same quantities in
 * lightgcn.py:59-74, gcl.py:98-108) from the ranked lists: for query row q and every cut-off n = cutoffs[c]
 * (ascending), hits[q, c] = #{p < n : top_items[q, p] in the user's test items}, dcg[q, c] = sum over those p of
 * 1 / log2(p + 2), idcg[q, c] = sum_{p < min(|test_q|, n)} 1 / log2(p + 2).  test_rowptr [n_query + 1] /
 * test_items_sorted: CSR of the test items per QUERY row, ascending inside a row; top_items: int64 [n_query, k]
 * (-1 = padding).  The means over users (hit ratio, precision, recall, NDCG) are the caller's.
 */
int32_t gcr_rank_metrics(const int64_t* top_items, int64_t n_query, int32_t k, const int64_t* test_rowptr,
                         const int32_t* test_items_sorted, const int32_t* cutoffs, int32_t n_cut,
                         int32_t* hits, double* dcg, double* idcg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Graph ingest on the device (integer work, bit-exact with the reference's host construction).
 * --------------------------------------------------------------------------------------------- */
int64_t gcr_coo_to_csr_workspace_bytes(int64_t nnz);

/*
 * COO (row, col int64 [nnz], val fp32 or NULL = ones) -> CSR.
 *   coalesce == 0: stable sort by row, COO order kept inside a row, duplicates kept — the layout
 *     torch.sparse.mm sees for the uncoalesced COO of ncl.py:74-85,203-209; perm_out (optional)
 *     receives the source position of every output non-zero;
 *   coalesce != 0: sorted by (row, col) with duplicate pairs summed — scipy's `tmp + tmp.T`
 *     selfcf.py:297, `.coalesce()` sept.py:50.
 * Outputs are sized for nnz entries; *nnz_out (device) is the number actually produced.  Entries
 * with a row/col outside the matrix are dropped and counted in *n_errors (device).
 */
int32_t gcr_coo_to_csr(const int64_t* row, const int64_t* col, const float* val, int64_t nnz,
                       int64_t n_rows, int64_t n_cols, int32_t coalesce,
                       int64_t* rowptr, int32_t* col_out, float* val_out, int64_t* perm_out,
                       int64_t* nnz_out, int64_t* n_errors, void* workspace, void* stream);

/*
 * val_out[e] = dinv_row[row(e)] * val[e] * dinv_col[col[e]],  dinv = (row sum)^-1/2 with inf -> 0.
 * Square symmetric operator (rowptr_t == NULL): Graph.normalize_graph_mat selfcf.py:240-249,
 * ssl4rec.py:85-88; with val == NULL on a dst-major CSR it is gcn_norm(add_self_loops=False) of
 * lightgcn.py's LGConv (in-degree counts).  Rectangular: pass the transposed CSR's rowptr_t / val_t
 * for the column sums.  dinv_row [n_rows] / dinv_col [n_cols] are caller-provided scratch.
 */
int32_t gcr_csr_sym_norm_f32(const int64_t* rowptr, const int32_t* col, const float* val,
                             int64_t n_rows, int64_t n_cols, const int64_t* rowptr_t, const float* val_t,
                             float* dinv_row, float* dinv_col, float* val_out, void* stream);

/* val_out[e] = val[e] / (row sum), 1/0 -> 0: the non-square branch of Graph.normalize_graph_mat
 * (selfcf.py:250-254 = ncl.py:37-41), used for MHCN's row-normalised R / H operators
 * (univariate/mhcn.py:340-368,401-402).  val == NULL = ones.  rinv [n_rows] is scratch. */
int32_t gcr_csr_row_norm_f32(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                             float* rinv, float* val_out, void* stream);

/*
 * Keep bitmap with EXACTLY n_keep of nnz bits set, a uniformly random subset without replacement
 * (64-bit Philox key per edge, stable radix sort, first n_keep win):
 * GraphAugmentor.edge_dropout univariate/sept.py:55-61 with n_keep = int(nnz * (1 - drop_rate)).
 */
int64_t gcr_edge_mask_exact_workspace_bytes(int64_t nnz);
int32_t gcr_edge_mask_exact_bits(int64_t nnz, int64_t n_keep, uint64_t seed, uint32_t* bits,
                                 void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The stages either side of the hot path.
 * --------------------------------------------------------------------------------------------- */
/*
 * One fused dense Adam step on an embedding table (torch.optim.Adam as ncl.py:305, lightgcn.py:84, gcl.py:201 use
 * it: no amsgrad, L2 weight_decay added to the gradient): g = grad_scale * (grad + grad2 + grad3) + weight_decay * p;
 * m += (1 - beta1) (g - m); v = beta2 v + (1 - beta2) g^2; p -= lr / (1 - beta1^step) * m / (sqrt(v / (1 - beta2^step)) + eps).
 * grad2 / grad3 are optional further gradient pieces of the same parameter (summed on the way in instead of by
 * separate element-wise adds).  Pointers 16-B aligned; step counts from 1.
 */
int32_t gcr_adam_step_f32(float* param, const float* grad, const float* grad2, const float* grad3, float* exp_avg,
                          float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t step, float grad_scale, void* stream);
/* The same step with the step count read from device memory (*step_dev >= 1, incremented by the caller on the same
 * stream): a training step captured in a hipGraph replays with the current count instead of the captured one. */
int32_t gcr_adam_step_dev_f32(float* param, const float* grad, const float* grad2, const float* grad3, float* exp_avg,
                              float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, const int64_t* step_dev, float grad_scale, void* stream);

/* out[r, c] = keep bit c ? x[r, c] : 0 — PyGCL's FeatureMasking / drop_feature (univariate/grace.py:261-278): whole
 * feature columns zeroed.  keep_bits: bit c of a little-endian uint32 bitmap (gcr_edge_mask_bits(d, pf, seed) draws
 * it: keep = u_c >= pf, as `uniform_(0, 1) < drop_prob` drops).  The backward is the same call on the gradient. */
int32_t gcr_mask_columns_f32(const float* x, int64_t n, int32_t d, const uint32_t* keep_bits, float* out, void* stream);

/*
 * Sparse x sparse product, expand step (MHCN's motif adjacency, univariate/mhcn.py:340-368: U.dot(U), Y.dot(Y.T), ...):
 * for every non-zero e = (i, k, a) of A and every non-zero (k, j, b) of B's row k one COO entry (i, j, a * b) at
 * out[offset[e] + position in row k]; offset = exclusive prefix sum of B's row lengths over A's non-zeros (caller),
 * a_row_of[e] = row of e.  The sort + duplicate sum that completes the product is gcr_coo_to_csr(coalesce = 1).
 * a_val / b_val NULL = ones.
 */
int32_t gcr_spgemm_expand_f32(const int64_t* a_rowptr, const int32_t* a_col, const float* a_val, int64_t a_rows,
                              int64_t a_nnz, const int32_t* a_row_of, const int64_t* b_rowptr, const int32_t* b_col,
                              const float* b_val, const int64_t* offset, int64_t* out_row, int64_t* out_col,
                              float* out_val, void* stream);

/* out[e] = value of (row_of[e], col[e]) in the CSR m (columns ascending inside a row; m_val NULL = ones), 0 when it
 * is not stored: the sparse element-wise products `.multiply(U.T)` / `.multiply(B)` of mhcn.py:340-368. */
int32_t gcr_csr_lookup_f32(const int32_t* row_of, const int32_t* col, int64_t nnz, const int64_t* m_rowptr,
                           const int32_t* m_col, const float* m_val, float* out, void* stream);

/* out[i, :] = table[idx[i], :] (0 for an id outside [0, n_rows)) and its backward out[idx[i], :] += src[i, :] (256-B
 * float-atomic row segments, duplicate ids add up, bad ids skipped): the batch-row gathers of the loss functions
 * (`emb[user_idx]`, ncl.py:314-316,360-361,370-373) without the sort that a generic index_put(accumulate) runs. */
int32_t gcr_gather_rows_f32(const float* table, const int64_t* idx, int64_t n, int32_t d, int64_t n_rows, float* out,
                            void* stream);
int32_t gcr_scatter_add_rows_f32(const float* src, const int64_t* idx, int64_t n, int32_t d, int64_t n_rows, float* out,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * Raw id -> dense id maps of graph ingest, on the device.
 *   order 0  dense id = rank of the key among the DISTINCT keys in ascending unsigned order
 *            replaces  self.user = {u: idx for idx, u in enumerate(sorted(users))}   ncl.py:60-61, directau.py:116-117,
 *                      univariate/sept.py:122-123
 *   order 1  dense id = order of first appearance
 *            replaces  if user not in self.user: self.user[user] = len(self.user)    selfcf.py:281-288, ssl4rec.py:69-75
 * keys [n]: one 64-bit key per record whose unsigned order is the raw ids' order (the host packs id strings big-endian,
 * 8 bytes per word; wider ids are folded word by word through order 0).  Outputs: dense [n] (id of every record),
 * first_pos [n] (entry j < *n_unique: position of the first record carrying dense id j — the host reads the raw id there),
 * n_unique (device int64).  Stable LSD radix sort (rocPRIM) + head flags + scan; no host involvement.
 * --------------------------------------------------------------------------------------------- */
int64_t gcr_dense_ids_workspace_bytes(int64_t n);
int32_t gcr_dense_ids_u64(const uint64_t* keys, int64_t n, int32_t order, int64_t* dense, int64_t* first_pos,
                          int64_t* n_unique, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Memory-system probes (measurement aids for the roofline block of bench.py; SURVEY.md §8d asks for
 * a device-copy bandwidth measured on the box next to the vendor peak).  16 B per lane.
 * --------------------------------------------------------------------------------------------- */
/* dst[i] = src[i]; n_floats a multiple of 4, both pointers 16-B aligned (reads n*4 B, writes n*4 B). */
int32_t gcr_probe_copy_f32(const float* src, float* dst, int64_t n_floats, void* stream);
/* streaming read of n_floats (multiple of 4) with an in-register reduction; sink is never written in practice. */
int32_t gcr_probe_read_f32(const float* src, int64_t n_floats, float* sink, void* stream);
/* out[k, :] = sum of the 64 table rows idx[64k .. 64k+63] (256-B rows, d = 64, 16 loads in flight per wave):
 * the gather shape of gcr_spmm_csr_f32 without its CSR streams.  n_idx a multiple of 64; ids are clamped
 * to [0, n_rows). out: [n_idx / 64, 64]. */
int32_t gcr_probe_gather_rows_f32(const float* table, int64_t n_rows, const int32_t* idx, int64_t n_idx,
                                  float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GCR_H */
