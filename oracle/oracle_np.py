"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference hot path.

Every function cites the reference file:line whose arithmetic it restates
(paths relative to the reference checkout).  The restatement computes in
float64 unless ``dtype`` says otherwise, so that it can serve as the "truth"
both for the reference's own fp32 CPU results (pinned in tests/golden, see
oracle/gen_golden.py) and for the HIP kernels (<= 1e-5 relative, fp32).

Pinning status (see DESIGN.md "Oracle"):
  * graph build, LGCNEncoder / LGCN_Encoder / SEPT.encoder propagation,
    info_nce_loss, InfoNCE, bpr_loss variants, l2_reg_loss, ssl_layer_loss,
    ProtoNCE_loss, batch_softmax_loss: pinned against outputs of the
    reference functions run in the build container (tests/golden/*.npz).
  * LightGCN.forward over torch_geometric.nn.LGConv (lightgcn.py:21-27):
    PARITY UNPINNED at the LGConv boundary (torch_geometric is not installed
    and no version is pinned by the reference); restated from PyG's documented
    gcn_norm(add_self_loops=False) semantics and cross-checked through the
    derived identity  LightGCN.forward == (K+1) * selfcf.LGCN_Encoder.forward
    on a de-duplicated symmetric graph.
  * negative sampler / edge masks: the reference RNG streams are unseeded
    python/numpy/torch generators, so only the distributional contract is
    pinned; bit-exactness is defined against the Philox restatement below.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64

# --------------------------------------------------------------------------
# graph construction (integer work: bit-exact)
# --------------------------------------------------------------------------


def build_edge_index(users, items, num_users):
    """lightgcn.py:36-39, gcl.py:72-77: [[u ; i+U], [i+U ; u]] int64 [2, 2E]."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    src = np.concatenate([users, items + num_users])
    dst = np.concatenate([items + num_users, users])
    return np.stack([src, dst])


def id_maps_sorted(train):
    """ncl.py:55-61 (= directau.py:111-117, sept.py:117-123): dense ids by sorted raw id."""
    users = sorted({t[0] for t in train})
    items = sorted({t[1] for t in train})
    return ({u: k for k, u in enumerate(users)}, {i: k for k, i in enumerate(items)})


def id_maps_first_seen(train):
    """selfcf.py:279-288, ssl4rec.py:69-75: dense ids by first appearance."""
    umap, imap = {}, {}
    for t in train:
        if t[0] not in umap:
            umap[t[0]] = len(umap)
        if t[1] not in imap:
            imap[t[1]] = len(imap)
    return umap, imap


def raw_adj_coo(uid, iid, num_users, num_items):
    """ncl.py:74-85: per interaction append (u, i+U) then (i+U, u), data = 1, duplicates kept (Q1)."""
    uid = np.asarray(uid, dtype=np.int64)
    iid = np.asarray(iid, dtype=np.int64) + num_users
    row = np.empty(2 * uid.size, dtype=np.int64)
    col = np.empty_like(row)
    row[0::2], row[1::2] = uid, iid
    col[0::2], col[1::2] = iid, uid
    return row, col, np.ones(row.size, dtype=np.float32)


def coo_to_csr_stable(row, col, val, n_rows):
    """Canonical device layout: stable counting sort by row, COO order kept inside a
    row, duplicates kept (torch.sparse.mm on an uncoalesced COO sums them, ncl.py:203-209,419)."""
    row = np.asarray(row, dtype=np.int64)
    order = np.argsort(row, kind="stable")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, row + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr, np.asarray(col)[order].astype(np.int32), np.asarray(val)[order].astype(np.float32), order


def coalesce_csr(row, col, val, n_rows):
    """selfcf.py:296-301 / ssl4rec.py:79-84 (`tmp + tmp.T` on scipy CSR): entries sorted by
    (row, col), duplicate (row, col) pairs summed in fp32."""
    row = np.asarray(row, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    key = row * (int(np.max(col, initial=0)) + 1) + col
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    first = np.ones(key_s.size, dtype=bool)
    first[1:] = key_s[1:] != key_s[:-1]
    seg = np.cumsum(first) - 1
    out_val = np.zeros(int(first.sum()), dtype=np.float32)
    np.add.at(out_val, seg, np.asarray(val, dtype=np.float32)[order])
    out_row = row[order][first]
    out_col = col[order][first]
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.add.at(rowptr, out_row + 1, 1)
    return np.cumsum(rowptr), out_col.astype(np.int32), out_val


def sym_norm_values(rowptr, col, val):
    """selfcf.py:240-249 (Graph.normalize_graph_mat, square case), ssl4rec.py:85-88:
    d = rowsum^-1/2 with inf -> 0 ; value <- d[row] * value * d[col] (float32 like scipy)."""
    n = rowptr.size - 1
    rowsum = np.zeros(n, dtype=np.float32)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    np.add.at(rowsum, rows, val.astype(np.float32))
    with np.errstate(divide="ignore"):
        d = np.power(rowsum, np.float32(-0.5), dtype=np.float32)
    d[np.isinf(d)] = 0.0
    return (d[rows] * val.astype(np.float32) * d[col]).astype(np.float32)


def norm_adj_csr(uid, iid, num_users, num_items):
    """selfcf.py:291-306 + 240-255: A = tmp + tmp^T (duplicates summed), then D^-1/2 A D^-1/2."""
    uid = np.asarray(uid, dtype=np.int64)
    iid = np.asarray(iid, dtype=np.int64) + num_users
    row = np.concatenate([uid, iid])
    col = np.concatenate([iid, uid])
    n = num_users + num_items
    rowptr, c, v = coalesce_csr(row, col, np.ones(row.size, np.float32), n)
    return rowptr, c, sym_norm_values(rowptr, c, v)


def gcn_norm_weights(edge_index, num_nodes):
    """lightgcn.py:17,25 -> torch_geometric LGConv -> gcn_norm(add_self_loops=False)
    (documented semantics; PARITY UNPINNED, see module docstring): deg[v] = number of
    edges whose target is v (duplicates counted); w_e = deg^-1/2[src] * deg^-1/2[dst], inf -> 0;
    out[dst] += w_e * x[src]."""
    src, dst = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    deg = np.zeros(num_nodes, dtype=np.float32)
    np.add.at(deg, dst, np.float32(1.0))
    with np.errstate(divide="ignore"):
        dis = np.power(deg, np.float32(-0.5), dtype=np.float32)
    dis[np.isinf(dis)] = 0.0
    return (dis[src] * dis[dst]).astype(np.float32)


# --------------------------------------------------------------------------
# propagation (S1-S3)
# --------------------------------------------------------------------------


def spmm_coo(row, col, val, x, n_rows=None, dtype=F64):
    """ncl.py:419 `torch.sparse.mm(A_coo, emb)`: y[row] += val * x[col] (duplicates sum)."""
    x = np.asarray(x, dtype=dtype)
    y = np.zeros((x.shape[0] if n_rows is None else n_rows, x.shape[1]), dtype=dtype)
    np.add.at(y, np.asarray(row), np.asarray(val, dtype=dtype)[:, None] * x[np.asarray(col)])
    return y


def spmm_csr(rowptr, col, val, x, n_rows=None, dtype=F64, keep=None, scale=1.0):
    """Same operator on the canonical CSR; ``keep`` is a per-nnz bool predicate
    (edge dropout consumed as a mask), ``scale`` the buir.py:300-309 1/(1-rate) rescale."""
    n = (rowptr.size - 1) if n_rows is None else n_rows
    rows = np.repeat(np.arange(rowptr.size - 1), np.diff(rowptr))
    v = np.asarray(val, dtype=dtype)
    if keep is not None:
        v = v * np.asarray(keep, dtype=dtype)
    x = np.asarray(x, dtype=dtype)
    y = np.zeros((n, x.shape[1]), dtype=dtype)
    np.add.at(y, rows, (v * scale)[:, None] * x[np.asarray(col)])
    return y


def row_l2_normalize(x, eps=1e-12):
    """F.normalize(x, dim=1): x / max(||x||_2, eps) (sept.py:224, ncl.py:127)."""
    x = np.asarray(x, dtype=F64)
    nrm = np.sqrt((x * x).sum(1, keepdims=True))
    return x / np.maximum(nrm, eps)


def lgcn_encoder_forward(rowptr, col, val, x0, n_layers, combine="mean", layer_norm=False, dtype=F64):
    """ncl.py:415-422 (= directau.py:286-293, selfcf.py:475-485): emb_k = A emb_{k-1};
    final = mean over the K+1 layer outputs.  combine='sum' restates lightgcn.py:22-27 (Q3);
    layer_norm=True restates sept.py:220-226 (row L2-normalise every propagated layer)."""
    emb = np.asarray(x0, dtype=dtype)
    all_emb = [emb]
    for _ in range(n_layers):
        emb = spmm_csr(rowptr, col, val, emb, dtype=dtype)
        if layer_norm:
            emb = row_l2_normalize(emb).astype(dtype)
        all_emb.append(emb)
    acc = all_emb[0].copy()
    for e in all_emb[1:]:
        acc = acc + e
    if combine == "mean":
        acc = acc / (n_layers + 1)
    return acc, all_emb


def lightgcn_forward(edge_index, user_w, item_w, n_layers):
    """lightgcn.py:21-27: x = cat(U, I); out = x; for conv: out = LGConv(out); x += out.
    Returns (x[:U], x[U:]) = sum over layers 0..K (no division, Q3)."""
    x0 = np.concatenate([user_w, item_w]).astype(F64)
    n = x0.shape[0]
    w = gcn_norm_weights(edge_index, n)
    # out[dst] += w * x[src]  <=> CSR over rows = dst, cols = src
    rowptr, col, val, _ = coo_to_csr_stable(edge_index[1], edge_index[0], w, n)
    acc, _ = lgcn_encoder_forward(rowptr, col, val, x0, n_layers, combine="sum")
    return acc[: user_w.shape[0]], acc[user_w.shape[0]:]


def spmm_backward(rowptr, col, val, dy, n_cols, keep=None, scale=1.0):
    """autograd of torch.sparse.mm w.r.t. the dense operand: dX = A^T dY."""
    rows = np.repeat(np.arange(rowptr.size - 1), np.diff(rowptr))
    v = np.asarray(val, dtype=F64) * scale
    if keep is not None:
        v = v * np.asarray(keep, dtype=F64)
    dy = np.asarray(dy, dtype=F64)
    dx = np.zeros((n_cols, dy.shape[1]), dtype=F64)
    np.add.at(dx, np.asarray(col), v[:, None] * dy[rows])
    return dx


# --------------------------------------------------------------------------
# contrastive losses (C1-C4)
# --------------------------------------------------------------------------


def _logsumexp(s, axis):
    m = s.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(s - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def row_lse_scores(a, b, inv_tau, normalize=True):
    """Row-wise logsumexp_j(<a_i, b_j> * inv_tau) without keeping more than the score matrix."""
    a = np.asarray(a, dtype=F64)
    b = np.asarray(b, dtype=F64)
    if normalize:
        a, b = row_l2_normalize(a), row_l2_normalize(b)
    s = (a @ b.T) * inv_tau
    return _logsumexp(s, 1), s


def infonce(view1, view2, temperature, b_cos=True):
    """ncl.py:125-130 (= ssl4rec.py:19-23): -mean_i( S_ii - logsumexp_j S_ij )."""
    lse, s = row_lse_scores(view1, view2, 1.0 / temperature, normalize=b_cos)
    return float(-(np.diag(s) - lse).mean())


def info_nce_loss(z1, z2, temp=0.2):
    """gcl.py:28-35: 0.5 * (CE(S, arange) + CE(S^T, arange)), S = z1n z2n^T / temp."""
    lse_r, s = row_lse_scores(z1, z2, 1.0 / temp, normalize=True)
    lse_c = _logsumexp(s, 0)
    d = np.diag(s)
    return float(0.5 * ((lse_r - d).mean() + (lse_c - d).mean()))


def ssl_layer_loss(context, initial, user_idx, item_idx, num_users, ssl_temp, ssl_reg, alpha):
    """ncl.py:358-367: anchors = normalised context rows at the batch ids, positives = own
    normalised layer-0 row, denominator = sum over ALL normalised layer-0 rows (naive exp
    in the reference; restated through logsumexp, identical in exact arithmetic); summed."""
    context = np.asarray(context, dtype=F64)
    initial = np.asarray(initial, dtype=F64)
    cu, ci = context[:num_users], context[num_users:]
    iu, ii = initial[:num_users], initial[num_users:]

    def side(c, i0, idx):
        a = row_l2_normalize(c[idx])
        p = row_l2_normalize(i0[idx])
        pos = (a * p).sum(1) / ssl_temp
        lse, _ = row_lse_scores(a, i0, 1.0 / ssl_temp, normalize=True)  # a is already unit norm
        return float(-(pos - lse).sum())

    return ssl_reg * (side(cu, iu, np.asarray(user_idx)) + alpha * side(ci, ii, np.asarray(item_idx)))


def proto_nce_loss(initial, user_idx, item_idx, num_users, user_centroids, user_2cluster,
                   item_centroids, item_2cluster, ssl_temp, proto_reg, batch_size):
    """ncl.py:369-375: InfoNCE(e0[idx], centroids[assign[idx]], tau) * batch_size, users + items."""
    initial = np.asarray(initial, dtype=F64)
    ue, ie = initial[:num_users], initial[num_users:]
    user_idx, item_idx = np.asarray(user_idx), np.asarray(item_idx)
    lu = infonce(ue[user_idx], np.asarray(user_centroids)[np.asarray(user_2cluster)[user_idx]], ssl_temp) * batch_size
    li = infonce(ie[item_idx], np.asarray(item_centroids)[np.asarray(item_2cluster)[item_idx]], ssl_temp) * batch_size
    return proto_reg * (lu + li)


def grace_infonce(h1, h2, tau, intraview_negs=True, exclude_self=False):
    """univariate/grace.py:218-224 + 396-419 + 448-502 (DualBranchContrast, L2L): dense float64.  sample =
    [other view; own view]; pos_mask = [eye | 0]; the denominator runs over pos + neg masks.
    exclude_self=False restates what add_extra_mask leaves when extra_neg_mask is None (neg = 1 - pos: the
    anchor's own row is a negative), True the sampler's mask (`1 - eye` on the intra-view block)."""
    a, b = row_l2_normalize(h1), row_l2_normalize(h2)
    m = a.shape[0]

    def one(x, y):
        s_inter = x @ y.T / tau
        mask = [np.ones((m, m), bool)]
        blocks = [s_inter]
        if intraview_negs:
            blocks.append(x @ x.T / tau)
            mask.append(~np.eye(m, dtype=bool) if exclude_self else np.ones((m, m), bool))
        s, k = np.concatenate(blocks, 1), np.concatenate(mask, 1)
        s_m = np.where(k, s, -np.inf)
        return float(np.mean(_logsumexp(s_m, 1) - np.diag(s_inter)))

    return 0.5 * (one(a, b) + one(b, a))


def batch_softmax_loss(user_emb, item_emb, temperature):
    """ssl4rec.py:25-30: mean(-log(exp(pos/t) / sum_j exp(<u,i_j>/t) + 1e-6)) on normalised rows."""
    lse, s = row_lse_scores(user_emb, item_emb, 1.0 / temperature, normalize=True)
    p = np.exp(np.diag(s) - lse)
    return float((-np.log(p + 1e-6)).mean())


def infonce_grads(a, b, pos_idx, inv_tau, normalize, row_w, col_w=None):
    """Gradient of  L = sum_i row_w[i] * (lse_i - s_{i,pos_i}) [+ sum_j col_w[j] * (clse_j - s_{jj})]
    w.r.t. the raw (un-normalised) inputs a [M,d], b [N,d]."""
    a = np.asarray(a, dtype=F64)
    b = np.asarray(b, dtype=F64)
    an, bn = (row_l2_normalize(a), row_l2_normalize(b)) if normalize else (a, b)
    s = (an @ bn.T) * inv_tau
    p = np.exp(s - _logsumexp(s, 1)[:, None]) * np.asarray(row_w, dtype=F64)[:, None]
    m = a.shape[0]
    p[np.arange(m), pos_idx] -= np.asarray(row_w, dtype=F64)
    if col_w is not None:
        q = np.exp(s - _logsumexp(s, 0)[None, :]) * np.asarray(col_w, dtype=F64)[None, :]
        q[np.arange(m), np.arange(m)] -= np.asarray(col_w, dtype=F64)
        p = p + q
    d_an = (p @ bn) * inv_tau
    d_bn = (p.T @ an) * inv_tau
    if not normalize:
        return d_an, d_bn

    def through_norm(x, xn, g):
        nrm = np.maximum(np.sqrt((x * x).sum(1, keepdims=True)), 1e-12)
        return (g - xn * (xn * g).sum(1, keepdims=True)) / nrm

    return through_norm(a, an, d_an), through_norm(b, bn, d_bn)


# --------------------------------------------------------------------------
# BPR + regularisers (P1, P2)
# --------------------------------------------------------------------------

BPR_NCL = 0         # -log(1e-5 + sigmoid(x))      ncl.py:116-120 (10e-6 literal), mhcn.py:35-39
BPR_LOGSIGMOID = 1  # -logsigmoid(x)               gcl.py:221, sept.py:34-38
BPR_LOG_SIGMOID = 2  # -log(sigmoid(x))            lightgcn.py:108


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def bpr_scores(user_tab, item_tab, u_idx, i_idx, j_idx):
    """Gathers + dots: ncl.py:314-318, lightgcn.py:95-105 (n_neg > 1: mean of the negatives' scores)."""
    u = np.asarray(user_tab, dtype=F64)[np.asarray(u_idx)]
    p = np.asarray(item_tab, dtype=F64)[np.asarray(i_idx)]
    j_idx = np.asarray(j_idx)
    n = np.asarray(item_tab, dtype=F64)[j_idx]
    pos = (u * p).sum(-1)
    if j_idx.ndim == 2:
        neg = (u[:, None, :] * n).sum(-1).mean(1)
    else:
        neg = (u * n).sum(-1)
    return pos, neg


def bpr_loss(user_tab, item_tab, u_idx, i_idx, j_idx, variant=BPR_NCL):
    pos, neg = bpr_scores(user_tab, item_tab, u_idx, i_idx, j_idx)
    x = pos - neg
    if variant == BPR_NCL:
        l = -np.log(10e-6 + _sigmoid(x))
    elif variant == BPR_LOGSIGMOID:
        l = np.logaddexp(0.0, -x)
    else:
        l = -np.log(_sigmoid(x))
    return float(l.mean())


def l2_reg_loss(reg, *args):
    """ncl.py:122-123: reg * sum_x ||x||_F / rows(x)."""
    return float(reg * sum(np.sqrt((np.asarray(x, dtype=F64) ** 2).sum()) / x.shape[0] for x in args))


def sq_norm_reg(*args):
    """lightgcn.py:118 / gcl.py:222 / sept.py:241: sum of squared Frobenius norms (caller scales)."""
    return float(sum((np.asarray(x, dtype=F64) ** 2).sum() for x in args))


def bpr_grads(user_tab, item_tab, u_idx, i_idx, j_idx, variant=BPR_NCL):
    """d(mean loss)/d(user_tab), d/d(item_tab) as dense tables (duplicates accumulate)."""
    U = np.asarray(user_tab, dtype=F64)
    I = np.asarray(item_tab, dtype=F64)
    u_idx, i_idx, j_idx = map(np.asarray, (u_idx, i_idx, j_idx))
    pos, neg = bpr_scores(U, I, u_idx, i_idx, j_idx)
    x = pos - neg
    s = _sigmoid(x)
    if variant == BPR_NCL:
        dl = -(s * (1 - s)) / (10e-6 + s)
    else:
        dl = -(1 - s)
    dl = dl / x.size
    gu, gi = np.zeros_like(U), np.zeros_like(I)
    u, p = U[u_idx], I[i_idx]
    if j_idx.ndim == 2:
        n = I[j_idx]
        k = j_idx.shape[1]
        np.add.at(gu, u_idx, dl[:, None] * (p - n.mean(1)))
        np.add.at(gi, i_idx, dl[:, None] * u)
        np.add.at(gi, j_idx.reshape(-1), np.repeat(-dl / k, k)[:, None] * np.repeat(u, k, axis=0))
    else:
        n = I[j_idx]
        np.add.at(gu, u_idx, dl[:, None] * (p - n))
        np.add.at(gi, i_idx, dl[:, None] * u)
        np.add.at(gi, j_idx, -dl[:, None] * u)
    return gu, gi


def bce_rows(a, b):
    """(row sums of softplus(a b^T) [M], sigmoid(a b^T) [M, N]) in float64 — the all-pairs part of lightgcn.py:110-113."""
    s = np.asarray(a, F64) @ np.asarray(b, F64).T
    return np.logaddexp(0.0, s).sum(1), 0.5 * (1.0 + np.tanh(0.5 * s))


def lightgcn_bce_loss(user_tab, item_tab, u_idx, i_idx, reg_weight=0.0):
    """lightgcn.py:95-96,109-118 with loss_type == "bce": scores = user_emb[pos_u] @ item_emb.T ([B, I]), labels one-hot at
    pos_i, F.binary_cross_entropy_with_logits (mean over B * I: softplus(s) - y s) + reg_weight (|u|^2 + |p|^2).
    Returns (loss, d loss / d user_tab, d loss / d item_tab)."""
    ut, it = np.asarray(user_tab, F64), np.asarray(item_tab, F64)
    u, i = np.asarray(u_idx), np.asarray(i_idx)
    bsz, n_items = u.size, it.shape[0]
    loss = 0.0
    gu, gi = np.zeros_like(ut), np.zeros_like(it)
    for lo in range(0, bsz, 8192):                      # the [B, I] matrix of the reference, a slab of rows at a time
        uc, ic = u[lo:lo + 8192], i[lo:lo + 8192]
        uv = ut[uc]
        rows, sig = bce_rows(uv, it)
        loss += rows.sum() - (uv * it[ic]).sum() + reg_weight * bsz * n_items * ((uv ** 2).sum() + (it[ic] ** 2).sum())
        sig[np.arange(uc.size), ic] -= 1.0
        gi += sig.T @ uv / (bsz * n_items)
        np.add.at(gu, uc, sig @ it / (bsz * n_items) + 2.0 * reg_weight * uv)
        np.add.at(gi, ic, 2.0 * reg_weight * it[ic])
    return loss / (bsz * n_items), gu, gi


# --------------------------------------------------------------------------
# counter-based RNG: negative sampler (N1) and edge masks (A1)
# --------------------------------------------------------------------------

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11), vectorised over numpy uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


STREAM_NEG = 0x4E454753   # 'NEGS'
STREAM_EDGE = 0x45444745  # 'EDGE'


def neg_sample_uniform(user_rowptr, user_items_sorted, u_idx, n_negs, num_items, seed, offset, max_trials):
    """Restates the *contract* of ncl.py:91-114 / gcl.py:111-125 / ssl4rec.py:33-50 /
    lightgcn.py:91-94 with a counter RNG: for slot s = offset + b*n_negs + k draw trial t as
    mulhi(philox(ctr=(s_lo, s_hi, t/4, STREAM_NEG), key=seed)[t%4], num_items); the first draw
    not in the user's sorted training row wins; max_trials == 0 -> no rejection (lightgcn.py);
    no winner within max_trials draws -> -1 (ncl.py:110-112 bail-out)."""
    u_idx = np.asarray(u_idx, dtype=np.int64)
    b = u_idx.size
    out = np.full(b * n_negs, -1, dtype=np.int64)
    slots = np.uint64(offset) + np.arange(b * n_negs, dtype=np.uint64)
    users = np.repeat(u_idx, n_negs)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    pending = np.ones(b * n_negs, dtype=bool)
    n_draws = max(1, max_trials)
    for t in range(n_draws):
        if not pending.any():
            break
        r = philox4x32_10((slots & _MASK32).astype(np.uint32), (slots >> np.uint64(32)).astype(np.uint32),
                          np.full(slots.size, t // 4, np.uint32), np.full(slots.size, STREAM_NEG, np.uint32), k0, k1)[t % 4]
        cand = ((r.astype(np.uint64) * np.uint64(num_items)) >> np.uint64(32)).astype(np.int64)
        if max_trials == 0:
            out[:] = cand
            break
        for s in np.nonzero(pending)[0]:
            lo, hi = user_rowptr[users[s]], user_rowptr[users[s] + 1]
            row = user_items_sorted[lo:hi]
            k = np.searchsorted(row, cand[s])
            if not (k < row.size and row[k] == cand[s]):
                out[s] = cand[s]
                pending[s] = False
    return out


def edge_keep_mask(nnz, pe, seed, first_edge=0):
    """gcl.py:22-25 `rand(num_edges) >= pe` with a counter RNG: edge e uses word e%4 of
    philox(ctr=(e/4 lo, e/4 hi, 0, STREAM_EDGE), key=seed); u = (r >> 8) * 2^-24 (torch.rand's
    24-bit float32 grid); keep = u >= float32(pe)."""
    e = np.uint64(first_edge) + np.arange(nnz, dtype=np.uint64)
    blk = e >> np.uint64(2)
    k0, k1 = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    r = philox4x32_10((blk & _MASK32).astype(np.uint32), (blk >> np.uint64(32)).astype(np.uint32),
                      np.zeros(nnz, np.uint32), np.full(nnz, STREAM_EDGE, np.uint32), k0, k1)
    r = np.stack(r, 1)[np.arange(nnz), (e & np.uint64(3)).astype(np.int64)]
    u = (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return u >= np.float32(pe)


# --------------------------------------------------------------------------
# synthetic bipartite graph (BASELINE.md §3 / SURVEY.md §8d)
# --------------------------------------------------------------------------


def synthetic_interactions(num_users, num_items, num_edges, seed=20250919, zipf_alpha=1.0, cap_frac=0.005):
    """Users uniform, items Zipf(alpha) with per-item probability capped at cap_frac,
    duplicate pairs removed, every user >= 1 interaction.  Returns int64 (users, items)."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, num_items + 1, dtype=np.float64) ** zipf_alpha
    p /= p.sum()
    for _ in range(50):
        over = p > cap_frac
        if not over.any():
            break
        excess = (p[over] - cap_frac).sum()
        p[over] = cap_frac
        p[~over] += excess * p[~over] / p[~over].sum()
    cdf = np.cumsum(p)
    cdf[-1] = 1.0
    perm = rng.permutation(num_items)  # popularity rank -> item id

    def draw_items(n):
        return perm[np.minimum(np.searchsorted(cdf, rng.random(n), side="right"), num_items - 1)]

    base_u = np.arange(num_users, dtype=np.int64)
    base_i = draw_items(num_users)
    keys = base_u * num_items + base_i
    need = num_edges - num_users
    while need > 0:
        n = int(need * 1.15) + 16
        k = rng.integers(0, num_users, n) * num_items + draw_items(n)
        k = np.setdiff1d(np.unique(k), keys, assume_unique=False)
        if k.size > need:
            k = rng.choice(k, need, replace=False)
        keys = np.concatenate([keys, k])
        need = num_edges - keys.size
    keys = keys[rng.permutation(keys.size)]
    return keys // num_items, keys % num_items


# --------------------------------------------------------------------------
# k-means E-step (K1) — ncl.py:340-356 delegates to faiss (absent: PARITY UNPINNED); this is the
# plain Lloyd iteration the HIP path implements, for an implementation-vs-restatement check.
# --------------------------------------------------------------------------


STREAM_KMSPLIT = 0x4B4D5350  # 'KMSP'


def kmeans_split_clusters(cent, counts, n_points, seed, it):
    """faiss Clustering.cpp `split_clusters` (the rule ncl.py:352 gets from faiss.Kmeans.train; faiss itself is absent:
    PARITY UNPINNED), with the draws of the HIP path: every empty cluster ci, ascending, walks cj = 0, 1, ..., k-1, 0, ...
    and accepts cj when u < (counts[cj] - 1) / (n_points - k), u = (x >> 8) * 2^-24 of
    Philox-4x32-10(ctr = (q lo, e, it, 'KMSP'), key = (seed lo + q hi, seed hi)) for trial q of the e-th empty cluster;
    64 k misses -> the largest cluster (smallest id among ties) if it has >= 2 points.  centroid[ci] = centroid[cj] *
    (1 +- 1/1024), centroid[cj] *= (1 -+ 1/1024), alternating over the dimensions; counts split in half.
    Works in place on float64 / float32 `cent` and float `counts`; returns the number of splits."""
    k, d = cent.shape
    empties = np.nonzero(np.asarray(counts) == 0)[0]
    denom = float(n_points - k)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    eps = 1.0 / 1024.0
    sign = np.where(np.arange(d) % 2 == 0, 1.0, -1.0)
    n_split = 0
    for e, ci in enumerate(empties):
        hit = -1
        if denom > 0:
            max_trials = 64 * k
            for q0 in range(0, max_trials, 4096):
                q = np.arange(q0, min(q0 + 4096, max_trials), dtype=np.uint64)
                cj = (q % np.uint64(k)).astype(np.int64)
                p = ((np.asarray(counts, dtype=np.float64)[cj] - 1.0) / denom).astype(np.float32)
                r = philox4x32_10((q & _MASK32).astype(np.uint32), np.full(q.size, e, np.uint32), np.full(q.size, it, np.uint32),
                                  np.full(q.size, STREAM_KMSPLIT, np.uint32),
                                  np.uint32((k0 + int(q0 >> 32)) & 0xFFFFFFFF), np.uint32(k1))[0]
                u = (r >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
                ok = np.nonzero(u < p)[0]
                if ok.size:
                    hit = int(cj[ok[0]])
                    break
        if hit < 0:
            best = int(np.argmax(counts))
            if counts[best] >= 2:
                hit = best
        if hit >= 0:
            c0 = cent[hit].copy()
            cent[ci] = c0 * (1.0 + eps * sign).astype(cent.dtype)
            cent[hit] = c0 * (1.0 - eps * sign).astype(cent.dtype)
            half = np.floor(counts[hit] * 0.5)
            counts[ci] = half
            counts[hit] -= half
            n_split += 1
    return n_split


def kmeans_lloyd(x, init_centroids, niter=20, split_seed=None, n_split_out=None):
    """Assignment = argmin_c ||x - c||^2 (ties -> smaller c); update = mean of members, an empty
    cluster keeps its centroid — or, with split_seed (the HIP path's behaviour), is re-seeded by
    `kmeans_split_clusters`; final assignment against the final centroids (index.search(x, 1))."""
    x = np.asarray(x, dtype=F64)
    cent = np.asarray(init_centroids, dtype=F64).copy()

    def assign(c):
        score = x @ c.T - 0.5 * (c * c).sum(1)[None, :]
        return np.argmax(score, axis=1)

    total = 0
    for it in range(niter):
        a = assign(cent)
        sums = np.zeros_like(cent)
        np.add.at(sums, a, x)
        cnt = np.bincount(a, minlength=cent.shape[0]).astype(F64)
        nz = cnt > 0
        cent[nz] = sums[nz] / cnt[nz, None]
        if split_seed is not None:
            total += kmeans_split_clusters(cent, cnt, x.shape[0], split_seed, it)
    if n_split_out is not None:
        n_split_out.append(total)
    return cent, assign(cent)


# --------------------------------------------------------------------------
# MHCN / sept_social / BUIR (univariate/): multi-channel layer loop, motif adjacency, dropout rescale
# pinned by tests/golden/{mhcn,sept_social,buir}.npz (oracle/gen_golden.py --mhcn / --sept-social / --buir)
# --------------------------------------------------------------------------


def csr_to_dense(indptr, indices, data, shape):
    out = np.zeros(tuple(int(s) for s in shape), dtype=F64)
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    np.add.at(out, (rows, np.asarray(indices)), np.asarray(data, dtype=F64))
    return out


def mhcn_motif_adjacency(S, Y):
    """univariate/mhcn.py:340-368 `build_hyper_adj_mats` on dense 0/1 matrices S [U,U] (social, directed) and
    Y [U,I] (interactions): the ten triangle-motif adjacencies A1..A10, then H_s = rownorm(A1+..+A7),
    H_j = rownorm(A8+A9), H_p = rownorm(A10 * (A10 > 3)).  `*` is the element-wise product, `@` the matrix one."""
    S, Y = np.asarray(S, dtype=F64), np.asarray(Y, dtype=F64)
    B = S * S.T
    U = S - B
    C1 = (U @ U) * U.T
    A1 = C1 + C1.T
    C2 = (B @ U) * U.T + (U @ B) * U.T + (U @ U) * B
    A2 = C2 + C2.T
    C3 = (B @ B) * U + (B @ U) * B + (U @ B) * B
    A3 = C3 + C3.T
    A4 = (B @ B) * B
    C5 = (U @ U) * U + (U @ U.T) * U + (U.T @ U) * U
    A5 = C5 + C5.T
    A6 = (U @ B) * U + (B @ U.T) * U.T + (U.T @ U) * B
    A7 = (U.T @ B) * U.T + (B @ U) * U + (U @ U.T) * B
    YY = Y @ Y.T
    A8 = YY * B
    A9 = YY * U
    A9 = A9 + A9.T
    A10 = YY - A8 - A9

    def rownorm(h):
        rs = h.sum(1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = h / rs
        return np.where(h != 0, out, 0.0)        # scipy only touches stored non-zeros: empty rows stay empty

    H_s = rownorm(A1 + A2 + A3 + A4 + A5 + A6 + A7)
    H_j = rownorm(A8 + A9)
    H_p = rownorm(A10 * (A10 > 3))
    return H_s, H_j, H_p


def _sigmoid64(x):
    return 1.0 / (1.0 + np.exp(-x))


def mhcn_self_gating(em, w, b):
    """mhcn.py:404-410: em * sigmoid(em @ W + b)."""
    return em * _sigmoid64(em @ w + b)


def mhcn_channel_attention(attention, attention_mat, *embs):
    """mhcn.py:412-420: softmax over channels of sum(attention * (emb @ attention_mat), 1); mixed = sum_c score_c emb_c."""
    w = np.stack([(attention * (e @ attention_mat)).sum(1) for e in embs])
    w = w - w.max(0, keepdims=True)
    score = np.exp(w) / np.exp(w).sum(0, keepdims=True)
    return sum(score[c][:, None] * embs[c] for c in range(len(embs))), score


def mhcn_layer_loop(H_s, H_j, H_p, R, user_emb, item_emb, gw, gb, attention, attention_mat, n_layers):
    """mhcn.py:422-466 on dense operators: per layer the RAW products A x feed the next layer while their
    row-normalised copies are appended and finally summed; items use R^T on the attention mix of the three
    channels + simple/2, the simple channel uses R on the (raw) item embeddings."""
    c = [mhcn_self_gating(user_emb, gw[k], gb[k]) for k in range(3)]
    simple = mhcn_self_gating(user_emb, gw[3], gb[3])
    items = np.asarray(item_emb, dtype=F64)
    all_c = [[c[0]], [c[1]], [c[2]]]
    all_simple, all_i = [simple], [items]
    for _ in range(n_layers):
        mixed, _ = mhcn_channel_attention(attention, attention_mat, *c)
        mixed = mixed + simple / 2
        for k, h in enumerate((H_s, H_j, H_p)):
            c[k] = h @ c[k]
            all_c[k].append(row_l2_normalize(c[k]))
        new_items = R.T @ mixed
        all_i.append(row_l2_normalize(new_items))
        simple = R @ items
        all_simple.append(row_l2_normalize(simple))
        items = new_items
    cs = [sum(a) for a in all_c]
    final_user, _ = mhcn_channel_attention(attention, attention_mat, *cs)
    final_user = final_user + sum(all_simple) / 2
    return final_user, sum(all_i)


def mhcn_hierarchical_self_supervision(em, adj, perms):
    """mhcn.py:480-506 with the three torch.randperm draws passed in (row_shuffle, row_column_shuffle twice)."""
    edge = adj @ em
    score = lambda a, b: (a * b).sum(1)      # noqa: E731
    pos = score(em, edge)
    neg1 = score(em[perms[0]], edge)
    neg2 = score(edge[perms[1]], em)
    local = (-np.log(_sigmoid64(pos - neg1)) - np.log(_sigmoid64(neg1 - neg2))).sum()
    graph = edge.mean(0, keepdims=True)
    pos = score(edge, graph)
    neg1 = score(edge[perms[2]], graph)
    return (-np.log(_sigmoid64(pos - neg1))).sum() + local


def neighbor_discrimination(positive, emb, aug_emb, temperature=0.1):
    """univariate/sept_social.py:408-420 on the already selected rows: emb = emb[unique_u],
    aug_emb = aug_user_embeddings[unique_u], positive int [B', k]."""
    e, a = row_l2_normalize(emb), row_l2_normalize(aug_emb)
    pos = (e[:, None, :] * a[np.asarray(positive)]).sum(2)
    ttl = e @ a.T
    return -np.log(np.exp(pos / temperature).sum(1) / np.exp(ttl / temperature).sum(1)).sum()


def sparse_dropout_values(val, keep, rate):
    """univariate/buir.py:300-309: kept non-zeros (dropout_mask = floor(1 - rate + rand)) times 1 / (1 - rate)."""
    return np.asarray(val, dtype=F64) * np.asarray(keep, dtype=F64) * (1.0 / (1.0 - rate))


# --------------------------------------------------------------------------
# ranking metrics (ncl.py:133-177), pinned by tests/golden/eval.json (the reference's own report lines)
# --------------------------------------------------------------------------


def ranking_report(origin, res, cutoffs):
    """ncl.py:133-177 `ranking_evaluation` + `Metric`: origin {user: {item: 1}}, res {user: [(item, score), ...]}.
    Per cut-off n over the users of `origin` that have a list: hit ratio = all hits / all test items, precision =
    hits / (users * n), recall = mean of hits_u / |test_u|, NDCG = mean of DCG_u / IDCG_u with gains
    1 / log2(rank + 2); every value rounded to 5 decimals, formatted as the reference prints them."""
    import math
    users = [u for u in origin if u in res]
    lines = []
    for n in cutoffs:
        hits, rec_sum, ndcg_sum = {}, 0.0, 0.0
        for u in users:
            top = [it for it, _ in res[u][:n]]
            hits[u] = len(set(origin[u]) & set(top))
        total_test = sum(len(origin[u]) for u in origin)
        for u in hits:
            rec_sum += hits[u] / len(origin[u])
        for u in res:
            dcg = sum(1.0 / math.log2(p + 2) for p, (it, _) in enumerate(res[u][:n]) if it in origin[u])
            idcg = sum(1.0 / math.log2(p + 2) for p in range(min(len(origin[u]), n)))
            ndcg_sum += dcg / idcg if idcg else 0
        lines.append(f"Top {n}\n")
        lines.append(f"Hit Ratio:{round(sum(hits.values()) / total_test, 5)}\n")
        lines.append(f"Precision:{round(sum(hits.values()) / (len(hits) * n), 5)}\n")
        lines.append(f"Recall:{round(float(np.mean([hits[u] / len(origin[u]) for u in hits])), 5)}\n")
        lines.append(f"NDCG:{round(ndcg_sum / len(res), 5)}\n")
    return lines
