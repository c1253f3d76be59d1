"""univariate/ encoders on the HIP path against outputs of the reference's own methods:
  mhcn.npz         MHCN.forward (mhcn.py:422-506): five-operator layer loop with the dual-output SpMM
  sept_social.npz  SEPT.encoder / social_encoder (sept_social.py:370-385), neighbor_discrimination (:408-420)
  buir.npz         LGCN_Encoder.sparse_dropout (buir.py:300-309) consumed as an edge mask + rescale,
                   gradient through the masked SYMMETRIC operator
Fixtures: oracle/gen_golden.py --mhcn / --sept-social / --buir (reference code lifted and run unchanged)."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(z, name):
    import recommendation_amd as ra
    shape = z[f"{name}_shape"]
    return ra.CsrGraph(z[f"{name}_indptr"], z[f"{name}_indices"], z[f"{name}_data"], int(shape[0]), int(shape[1]), DEV)


def _close(got, ref, rel=1e-5):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= rel * scale, (np.abs(got - ref).max(), scale)


@pytest.mark.parametrize("d", [64, 48, 128])
def test_dual_output_spmm_matches_oracle(d):
    """gcr_spmm_csr_dual_f32: raw and row-normalised product from one launch, and its backward through both."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    rng = np.random.default_rng(d)
    n_r, n_c, nnz = 700, 500, 9000
    row, col = rng.integers(0, n_r, nnz), rng.integers(0, n_c, nnz)
    row[row == 3] = 4                                    # an empty row: raw 0, normalised 0
    row[:1500] = 11                                      # a split (long) row
    val = rng.standard_normal(nnz).astype(np.float32)
    g = ra.CsrGraph.from_coo(row, col, val, n_r, n_c, DEV)
    x = rng.standard_normal((n_c, d)).astype(np.float32)
    xt = torch.from_numpy(x).to(DEV).requires_grad_(True)
    raw, nrm = Fn.spmm_l2norm_dual(g, xt)
    rp, ci, va = g.rowptr_host, g.col.cpu().numpy(), g.val.cpu().numpy()
    ref_raw = O.spmm_csr(rp, ci, va, x, n_r)
    ref_nrm = O.row_l2_normalize(ref_raw)
    _close(raw, ref_raw)
    _close(nrm, ref_nrm)
    assert float(raw[3].abs().max()) == 0.0 and float(nrm[3].abs().max()) == 0.0
    w1, w2 = rng.standard_normal((n_r, d)), rng.standard_normal((n_r, d))
    ((raw * torch.from_numpy(w1).to(DEV)).sum() + (nrm * torch.from_numpy(w2).to(DEV)).sum()).backward()
    inv = 1.0 / np.maximum(np.sqrt((ref_raw ** 2).sum(1, keepdims=True)), 1e-12)
    dz = w1 + (w2 - ref_nrm * (ref_nrm * w2).sum(1, keepdims=True)) * inv
    dz[3] = w1[3]                                        # clamped row: F.normalize's gradient there is w2 * 1e12 * 0-row
    ref_dx = O.spmm_backward(rp, ci, va, dz, n_c)
    _close(xt.grad, ref_dx, 2e-5)


@pytest.mark.parametrize("d", [64, 48, 128])
def test_dual_output_spmm_with_the_layer_sum_folded_in(d):
    """gcr_spmm_csr_dual_acc_f32 (`Fn.spmm_l2norm_dual_acc`): (A x, acc + normalize(A x)) = the dual launch followed by the add,
    bit for bit; its backward (the normalised rows rebuilt from the raw ones, gcr_normalize_bwd_raw_f32) = the composition's,
    for x and for the running sum; empty and split rows."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    rng = np.random.default_rng(100 + d)
    n_r, n_c, nnz = 700, 500, 9000
    row, col = rng.integers(0, n_r, nnz), rng.integers(0, n_c, nnz)
    row[row == 3] = 4
    row[:1500] = 11
    g = ra.CsrGraph.from_coo(row, col, rng.standard_normal(nnz).astype(np.float32), n_r, n_c, DEV)
    x1 = torch.from_numpy(rng.standard_normal((n_c, d)).astype(np.float32)).to(DEV).requires_grad_(True)
    a1 = torch.from_numpy(rng.standard_normal((n_r, d)).astype(np.float32)).to(DEV).requires_grad_(True)
    x2, a2 = x1.detach().clone().requires_grad_(True), a1.detach().clone().requires_grad_(True)
    w1 = torch.from_numpy(rng.standard_normal((n_r, d)).astype(np.float32)).to(DEV)
    w2 = torch.from_numpy(rng.standard_normal((n_r, d)).astype(np.float32)).to(DEV)
    raw1, sum1 = Fn.spmm_l2norm_dual_acc(g, x1, a1)
    raw2, nrm2 = Fn.spmm_l2norm_dual(g, x2)
    sum2 = a2 + nrm2
    assert torch.equal(raw1, raw2) and torch.equal(sum1, sum2)
    ((raw1 * w1).sum() + (sum1 * w2).sum()).backward()
    ((raw2 * w1).sum() + (sum2 * w2).sum()).backward()
    assert torch.equal(a1.grad, a2.grad)
    assert float((x1.grad - x2.grad).abs().max()) <= 2e-6 * float(x2.grad.abs().max())


def test_mhcn_forward_matches_reference(golden):
    from recommendation_amd.mhcn import MHCNEncoder
    z = golden("mhcn.npz")
    for concurrent in (True, False):
        enc = MHCNEncoder(_graph(z, "H_s"), _graph(z, "H_j"), _graph(z, "H_p"), _graph(z, "R"), 64, int(z["n_layers"]),
                          float(z["ss_rate"]), concurrent=concurrent)
        with torch.no_grad():
            enc.user_embeddings.copy_(torch.from_numpy(z["user_emb"]))
            enc.item_embeddings.copy_(torch.from_numpy(z["item_emb"]))
            enc.attention.copy_(torch.from_numpy(z["attention"]))
            enc.attention_mat.copy_(torch.from_numpy(z["attention_mat"]))
            for c in (1, 2, 3, 4):
                enc.gating_weights[str(c)].copy_(torch.from_numpy(z[f"gw{c}"]))
                enc.gating_bias[str(c)].copy_(torch.from_numpy(z[f"gb{c}"]))
                enc.sgating_weights[str(c)].copy_(torch.from_numpy(z[f"sgw{c}"]))
                enc.sgating_bias[str(c)].copy_(torch.from_numpy(z[f"sgb{c}"]))
        perms = [torch.from_numpy(p).to(DEV) for p in z["perms"]]
        idx = [torch.from_numpy(z[k]).to(DEV) for k in ("u_idx", "v_idx", "j_idx")]
        bu, bp, bn, ss, fu, fi = enc(*idx, perms=perms)
        _close(fu, z["final_user"])
        _close(fi, z["final_item"])
        _close(bu, z["batch_user"])
        _close(bn, z["batch_neg"])
        assert float(ss) == pytest.approx(float(z["ss_loss"]), rel=2e-5)
        ((fu * torch.from_numpy(z["wu"]).to(DEV)).sum() + (fi * torch.from_numpy(z["wi"]).to(DEV)).sum() + ss).backward()
        _close(enc.user_embeddings.grad, z["grad_user"], 2e-5)
        _close(enc.item_embeddings.grad, z["grad_item"], 2e-5)


def test_mhcn_raw_product_feeds_the_next_layer(golden):
    """The semantics VERDICT r1 flagged: feeding the NORMALISED rows forward (SEPT-style) gives a different
    result than the reference's loop; the golden distinguishes the two."""
    z = golden("mhcn.npz")
    H = O.csr_to_dense(z["H_s_indptr"], z["H_s_indices"], z["H_s_data"], z["H_s_shape"])
    x = O.mhcn_self_gating(z["user_emb"].astype(np.float64), z["gw1"].astype(np.float64), z["gb1"].astype(np.float64))
    raw2 = O.row_l2_normalize(H @ (H @ x))
    sept2 = O.row_l2_normalize(H @ O.row_l2_normalize(H @ x))
    assert np.abs(raw2 - sept2).max() > 1e-3


def test_sept_social_encoder_and_neighbor_discrimination(golden):
    from recommendation_amd import losses
    from recommendation_amd.encoders import sept_encoder
    z = golden("sept_social.npz")
    n_u, k = int(z["n_users"]), int(z["n_layers"])
    ego = torch.from_numpy(z["ego"]).to(DEV).requires_grad_(True)
    final = sept_encoder(ego, _graph(z, "norm_adj"), k, combine="sum")
    _close(final[:n_u], z["rec_user"])
    _close(final[n_u:], z["rec_item"])
    (final * torch.from_numpy(z["w"]).to(DEV)).sum().backward()
    _close(ego.grad, z["enc_grad"], 2e-5)
    users = ego.detach()[:n_u]
    _close(sept_encoder(users, _graph(z, "social"), k, combine="sum"), z["friend_view"])
    _close(sept_encoder(users, _graph(z, "sharing"), k, combine="sum"), z["sharing_view"])
    uniq = torch.unique(torch.from_numpy(z["u_idx"])).to(DEV)
    emb = torch.from_numpy(z["friend_view"]).to(DEV).requires_grad_(True)
    aug = torch.from_numpy(z["aug_user"]).to(DEV)
    loss = losses.neighbor_discrimination(torch.from_numpy(z["positive"]).to(DEV), emb[uniq], aug[uniq], 0.1)
    assert float(loss) == pytest.approx(float(z["nd_loss"]), rel=1e-5)
    loss.backward()
    _close(emb.grad, z["nd_grad"], 2e-5)


def test_buir_sparse_dropout_masked_symmetric_backward(golden):
    """buir.py:300-326: the reference's own Bernoulli draw handed over as a keep bitmap + 1/(1-rate) rescale.
    The operator is symmetric, the mask is not: the backward needs the mask in A^T's order (mirror_perm)."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    from recommendation_amd.graph import coo_to_csr_device
    z = golden("buir.npz")
    n = int(z["n_users"]) + int(z["n_items"])
    rate, k = float(z["rate"]), int(z["n_layers"])
    rp, c, v, perm = coo_to_csr_device(z["adj_row"], z["adj_col"], z["adj_val"], n, n, DEV, want_perm=True)
    g = ra.CsrGraph(rp, c, v, n, n, DEV, symmetric=True)
    keep = torch.from_numpy(z["keep"]).to(DEV)[perm]          # the draw of every stored non-zero, in CSR order
    bits = Fn.pack_bits(keep)
    bits_t = Fn.mirror_bits(bits, g.mirror_perm(), g.nnz)
    x = torch.from_numpy(z["x"]).to(DEV).requires_grad_(True)
    with pytest.raises(ValueError):                           # an asymmetric mask must bring its transpose
        Fn.spmm(g, x, keep_bits=bits, val_scale=1 / (1 - rate))
    acc, e = x, x
    for _ in range(k):
        e = Fn.spmm(g, e, keep_bits=bits, keep_bits_t=bits_t, val_scale=1 / (1 - rate))
        acc = acc + e
    final = acc / (k + 1)
    _close(final, z["final"])
    (final * torch.from_numpy(z["w"]).to(DEV)).sum().backward()
    _close(x.grad, z["grad"], 2e-5)
    # a counter-based mask: edge_id = mirror gives the transposed mask without a gather
    m = g.mirror_perm()
    b1 = Fn.edge_mask_bits(g.nnz, 0.3, 5, DEV)
    b1t = Fn.edge_mask_bits(g.nnz, 0.3, 5, DEV, edge_id=m)
    assert torch.equal(b1t, Fn.mirror_bits(b1, m, g.nnz))
