"""Pins the CPU oracle (oracle/oracle_np.py) against outputs of the reference's own
functions (tests/golden/*.npz, produced by oracle/gen_golden.py in the build container).
Integer work is bit-exact; fp work is compared at fp32 resolution (the reference ran in
CPU PyTorch fp32, the oracle runs in float64)."""
import numpy as np
import pytest

from oracle import oracle_np as O

RT = 2e-5  # fp32 reference vs fp64 oracle


def _maps(train_user, train_item, fn):
    train = [[u, i, 1.0] for u, i in zip(train_user.tolist(), train_item.tolist())]
    umap, imap = fn(train)
    uid = np.array([umap[u] for u in train_user.tolist()])
    iid = np.array([imap[i] for i in train_item.tolist()])
    return umap, imap, uid, iid


def test_graph_build_sorted_ids_and_raw_coo(golden):
    g = golden("graph_build.npz")
    umap, imap, uid, iid = _maps(g["train_user"], g["train_item"], O.id_maps_sorted)
    assert [k for k, _ in sorted(umap.items(), key=lambda kv: kv[1])] == g["sorted_user_ids"].tolist()
    assert [k for k, _ in sorted(imap.items(), key=lambda kv: kv[1])] == g["sorted_item_ids"].tolist()
    row, col, data = O.raw_adj_coo(uid, iid, len(umap), len(imap))
    assert np.array_equal(row, g["coo_row"]) and np.array_equal(col, g["coo_col"])
    assert np.array_equal(data, g["coo_data"])


def test_graph_build_first_seen_ids_and_norm_csr(golden):
    g = golden("graph_build.npz")
    umap, imap, uid, iid = _maps(g["train_user"], g["train_item"], O.id_maps_first_seen)
    assert [k for k, _ in sorted(umap.items(), key=lambda kv: kv[1])] == g["seen_user_ids"].tolist()
    assert [k for k, _ in sorted(imap.items(), key=lambda kv: kv[1])] == g["seen_item_ids"].tolist()
    rowptr, col, val = O.norm_adj_csr(uid, iid, len(umap), len(imap))
    assert np.array_equal(rowptr, g["norm_indptr"])
    assert np.array_equal(col.astype(np.int64), g["norm_indices"])
    np.testing.assert_allclose(val, g["norm_data"], rtol=3e-7, atol=0)


def test_gcl_edge_index(golden):
    g = golden("graph_build.npz")
    nu = max(g["gcl_user"].max(), g["gcl_test_user"].max()) + 1
    ni = max(g["gcl_item"].max(), g["gcl_test_item"].max()) + 1
    assert (nu, ni) == (int(g["gcl_num_users"]), int(g["gcl_num_items"]))
    assert np.array_equal(O.build_edge_index(g["gcl_user"], g["gcl_item"], nu), g["gcl_edge_index"])


def _raw_csr(golden):
    g = golden("graph_build.npz")
    n = len(g["sorted_user_ids"]) + len(g["sorted_item_ids"])
    rowptr, col, val, _ = O.coo_to_csr_stable(g["coo_row"], g["coo_col"], g["coo_data"], n)
    return rowptr, col, val, n


@pytest.mark.parametrize("k", [1, 2, 3])
def test_lgcn_encoder_raw_adjacency(golden, k):
    p = golden("propagation.npz")
    rowptr, col, val, n = _raw_csr(golden)
    final, all_emb = O.lgcn_encoder_forward(rowptr, col, val, p["x0"], k, combine="mean")
    ref = p[f"raw_mean_K{k}"]
    np.testing.assert_allclose(final, ref, rtol=RT, atol=RT * np.abs(ref).max())
    np.testing.assert_allclose(all_emb[-1], p[f"raw_last_K{k}"], rtol=RT, atol=RT * np.abs(p[f"raw_last_K{k}"]).max())
    # backward of sum(final * w): d/dx0 = mean_k (A^T)^k w ; A symmetric here
    gacc, g = p["w"].astype(np.float64), p["w"].astype(np.float64)
    for _ in range(k):
        g = O.spmm_backward(rowptr, col, val, g, n)
        gacc = gacc + g
    gref = p[f"raw_grad_K{k}"]
    np.testing.assert_allclose(gacc / (k + 1), gref, rtol=RT, atol=RT * np.abs(gref).max())


@pytest.mark.parametrize("k", [2, 3])
def test_lgcn_encoder_normalised_adjacency(golden, k):
    g, p = golden("graph_build.npz"), golden("propagation.npz")
    final, _ = O.lgcn_encoder_forward(g["norm_indptr"], g["norm_indices"], g["norm_data"], p["xs"], k, combine="mean")
    np.testing.assert_allclose(final, p[f"norm_mean_K{k}"], rtol=RT, atol=RT * np.abs(p[f"norm_mean_K{k}"]).max())


def test_lightgcn_forward_derived_pin(golden):
    """PARITY UNPINNED at LGConv; derived pin: on the same symmetric multigraph
    (gcn_norm counts duplicate edges in the degree and the scatter sums them, exactly like
    selfcf's `tmp + tmp.T` with summed duplicates) LightGCN.forward (sum of layers,
    lightgcn.py:26) == (K+1) * selfcf.LGCN_Encoder.forward (mean of layers)."""
    g, p = golden("graph_build.npz"), golden("propagation.npz")
    umap, imap, uid, iid = _maps(g["train_user"], g["train_item"], O.id_maps_first_seen)
    nu = len(umap)
    edge_index = O.build_edge_index(uid, iid, nu)
    for k in (2, 3):
        u, i = O.lightgcn_forward(edge_index, p["xs"][:nu], p["xs"][nu:], k)
        ref = (k + 1) * p[f"norm_mean_K{k}"]
        np.testing.assert_allclose(np.concatenate([u, i]), ref, rtol=RT, atol=RT * np.abs(ref).max())


def test_sept_encoder(golden):
    g, p = golden("graph_build.npz"), golden("propagation.npz")
    n = len(g["sorted_user_ids"]) + len(g["sorted_item_ids"])
    # sept.py:42-50 coalesces the raw COO: duplicate interactions become value 2
    rowptr, col, val = O.coalesce_csr(g["coo_row"], g["coo_col"], g["coo_data"], n)
    final, _ = O.lgcn_encoder_forward(rowptr, col, val, p["x0"], 2, combine="mean", layer_norm=True)
    np.testing.assert_allclose(final, p["sept_mean_K2"], rtol=RT, atol=RT * np.abs(p["sept_mean_K2"]).max())


@pytest.mark.parametrize("m", [1, 7, 257, 1000])
def test_contrast_losses(golden, m):
    c = golden("contrast.npz")
    z1, z2 = c[f"z1_{m}"], c[f"z2_{m}"]
    for temp in (0.1, 0.2, 0.5):
        assert O.info_nce_loss(z1, z2, temp) == pytest.approx(float(c[f"gcl_loss_{m}_{temp}"]), rel=RT, abs=2e-6)
    for b_cos in (True, False):
        ref = float(c[f"ncl_infonce_{m}_{int(b_cos)}"])
        assert O.infonce(0.3 * z1, 0.3 * z2, 0.2, b_cos) == pytest.approx(ref, rel=RT, abs=2e-6)
        assert float(c[f"s4r_infonce_{m}_{int(b_cos)}"]) == pytest.approx(ref, rel=1e-6, abs=1e-6)
    assert O.batch_softmax_loss(z1, z2, 0.2) == pytest.approx(float(c[f"s4r_bsl_{m}"]), rel=RT, abs=2e-6)


def test_contrast_grads(golden):
    c = golden("contrast.npz")
    for m in (7, 257):
        z1, z2 = c[f"z1_{m}"], c[f"z2_{m}"]
        w = np.full(m, 0.5 / m)
        g1, g2 = O.infonce_grads(z1, z2, np.arange(m), 1 / 0.2, True, w, w)
        np.testing.assert_allclose(g1, c[f"gcl_g1_{m}"], rtol=1e-4, atol=1e-5 * np.abs(c[f"gcl_g1_{m}"]).max())
        np.testing.assert_allclose(g2, c[f"gcl_g2_{m}"], rtol=1e-4, atol=1e-5 * np.abs(c[f"gcl_g2_{m}"]).max())
    m = 257
    z1, z2 = 0.3 * c[f"z1_{m}"], 0.3 * c[f"z2_{m}"]
    for b_cos in (1, 0):
        g1, g2 = O.infonce_grads(z1, z2, np.arange(m), 5.0, bool(b_cos), np.full(m, 1.0 / m))
        r1, r2 = c[f"ncl_infonce_g1_{m}_{b_cos}"], c[f"ncl_infonce_g2_{m}_{b_cos}"]
        # b_cos=False saturates the softmax (logit gap ~25): the fp32 reference gradient is
        # cancellation noise around 1e-9, hence the absolute floor
        np.testing.assert_allclose(g1, r1, rtol=1e-4, atol=max(1e-5 * np.abs(r1).max(), 1e-8))
        np.testing.assert_allclose(g2, r2, rtol=1e-4, atol=max(1e-5 * np.abs(r2).max(), 1e-8))


def test_ncl_structure_and_prototype_losses(golden):
    c = golden("contrast.npz")
    nu = int(c["ncl_num_users"])
    ssl = O.ssl_layer_loss(c["ncl_ctx"], c["ncl_x0"], c["ncl_uidx"], c["ncl_iidx"], nu,
                           float(c["ncl_ssl_temp"]), float(c["ncl_ssl_reg"]), float(c["ncl_alpha"]))
    assert ssl == pytest.approx(float(c["ncl_ssl"]), rel=RT)
    proto = O.proto_nce_loss(c["ncl_x0"], c["ncl_uidx"], c["ncl_iidx"], nu, c["ncl_ucent"], c["ncl_u2c"],
                             c["ncl_icent"], c["ncl_i2c"], float(c["ncl_ssl_temp"]), float(c["ncl_proto_reg"]),
                             int(c["ncl_bsz"]))
    assert proto == pytest.approx(float(c["ncl_proto"]), rel=RT)


def test_bpr_and_regularisers(golden):
    b = golden("bpr.npz")
    ut, it, u, i, j, j3 = (b[k] for k in ("user_tab", "item_tab", "u_idx", "i_idx", "j_idx", "j_idx3"))
    assert O.bpr_loss(ut, it, u, i, j, O.BPR_NCL) == pytest.approx(float(b["ncl_bpr_loss"]), rel=RT)
    assert O.bpr_loss(ut, it, u, i, j, O.BPR_LOGSIGMOID) == pytest.approx(float(b["sept_bpr_loss"]), rel=RT)
    for var, name in ((O.BPR_NCL, "ncl_bpr"), (O.BPR_LOGSIGMOID, "sept_bpr")):
        gu, gi = O.bpr_grads(ut, it, u, i, j, var)
        np.testing.assert_allclose(gu, b[f"{name}_gu"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gu"]).max())
        np.testing.assert_allclose(gi, b[f"{name}_gi"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gi"]).max())
    l2 = O.l2_reg_loss(1e-4, ut[u], it[i], it[j])
    assert l2 == pytest.approx(float(b["ncl_l2reg_loss"]), rel=RT)
    assert l2 == pytest.approx(float(b["directau_l2reg_loss"]), rel=RT)
    # lightgcn.py:95-118 block = -log(sigmoid) BPR + reg_weight * (|u|^2 + |p|^2)
    for jj, name in ((j, "lgcn_block_n1"), (j3, "lgcn_block_n3")):
        val = O.bpr_loss(ut, it, u, i, jj, O.BPR_LOG_SIGMOID) + 1e-4 * O.sq_norm_reg(ut[u], it[i])
        assert val == pytest.approx(float(b[f"{name}_loss"]), rel=RT)
        gu, gi = O.bpr_grads(ut, it, u, i, jj, O.BPR_LOG_SIGMOID)
        np.add.at(gu, u, 2e-4 * ut[u])
        np.add.at(gi, i, 2e-4 * it[i])
        np.testing.assert_allclose(gu, b[f"{name}_gu"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gu"]).max())
        np.testing.assert_allclose(gi, b[f"{name}_gi"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gi"]).max())
    val = O.bpr_loss(ut, it, u, i, j, O.BPR_LOGSIGMOID) + 1e-4 * O.sq_norm_reg(ut[u], it[i], it[j]) / u.size
    assert val == pytest.approx(float(b["gcl_block_loss"]), rel=RT)


def test_lightgcn_bce_block(golden):
    # lightgcn.py:95-118 with loss_type == "bce", lifted from the epoch loop's statements by the generator
    b = golden("bpr.npz")
    u, i = b["u_idx"], b["i_idx"]
    for name, ut, it in (("lgcn_bce", b["user_tab"], b["item_tab"]), ("lgcn_bce_big", b["user_tab_big"], b["item_tab_big"])):
        loss, gu, gi = O.lightgcn_bce_loss(ut, it, u, i, reg_weight=1e-4)
        assert loss == pytest.approx(float(b[f"{name}_loss"]), rel=RT)
        np.testing.assert_allclose(gu, b[f"{name}_gu"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gu"]).max())
        np.testing.assert_allclose(gi, b[f"{name}_gi"], rtol=1e-4, atol=1e-5 * np.abs(b[f"{name}_gi"]).max())


def test_augmentation_contract(golden):
    a = golden("augment.npz")
    # gcl.py:22-25 Bernoulli keep: rate ~ 1 - pe ; sept.py:55-61 keeps exactly floor(nnz*(1-p)) with value 1
    assert abs(int(a["gcl_kept"]) / int(a["gcl_nnz"]) - (1 - float(a["gcl_pe"]))) < 0.1
    assert int(a["sept_kept"]) == int(int(a["sept_nnz"]) * (1 - float(a["sept_rate"])))
    assert bool(a["sept_vals_all_one"])
    keep = O.edge_keep_mask(200000, 0.3, seed=7)
    assert abs(keep.mean() - 0.7) < 5e-3
    assert np.array_equal(keep[1000:2000], O.edge_keep_mask(1000, 0.3, seed=7, first_edge=1000))


def test_philox_known_answer():
    """Random123 known-answer vectors for philox4x32-10."""
    out = O.philox4x32_10([0], [0], [0], [0], 0, 0)
    assert [int(x[0]) for x in out] == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    out = O.philox4x32_10([0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], [0xFFFFFFFF], 0xFFFFFFFF, 0xFFFFFFFF)
    assert [int(x[0]) for x in out] == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    out = O.philox4x32_10([0x243F6A88], [0x85A308D3], [0x13198A2E], [0x03707344], 0xA4093822, 0x299F31D0)
    assert [int(x[0]) for x in out] == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_negative_sampler_contract():
    u, i = O.synthetic_interactions(60, 40, 600, seed=3)
    order = np.lexsort((i, u))
    rowptr = np.zeros(61, dtype=np.int64)
    np.add.at(rowptr, u + 1, 1)
    rowptr = np.cumsum(rowptr)
    items_sorted = i[order]
    ub = u[:128]
    neg = O.neg_sample_uniform(rowptr, items_sorted, ub, 2, 40, seed=11, offset=5, max_trials=101)
    assert neg.shape == (256,) and neg.min() >= 0 and neg.max() < 40
    pos = {(a, b) for a, b in zip(u.tolist(), i.tolist())}
    assert all((int(a), int(b)) not in pos for a, b in zip(np.repeat(ub, 2), neg))
    free = O.neg_sample_uniform(rowptr, items_sorted, ub, 1, 40, seed=11, offset=5, max_trials=0)
    assert free.min() >= 0 and free.max() < 40  # lightgcn.py:92: plain randint, positives allowed


def test_synthetic_graph_contract():
    u, i = O.synthetic_interactions(943, 1682, 80000, seed=20250919)
    assert u.size == 80000 and np.unique(u * 1682 + i).size == 80000
    assert np.unique(u).size == 943 and i.max() < 1682
    assert np.bincount(i).max() <= 0.006 * 80000 + 50


def test_grace_dual_branch_infonce(golden):
    """univariate/grace.py DualBranchContrast(InfoNCE, 'L2L') run as-is (oracle/gen_golden.py --grace):
    without intra-view negatives, with them as the model calls it (the anchor's own row stays a negative,
    grace.py:448-455) and with the sampler's mask kept (diagonal excluded, grace.py:399-404)."""
    g = golden("grace.npz")
    for m in (7, 257):
        for tau in (0.2, 0.5):
            for intra, keep in ((0, 0), (1, 0), (1, 1)):
                got = O.grace_infonce(g[f"h1_{m}"], g[f"h2_{m}"], tau, bool(intra), bool(keep))
                assert got == pytest.approx(float(g[f"loss_{m}_{tau}_{intra}_{keep}"]), rel=RT, abs=2e-6)


# --------------------------------------------------------------------------- MHCN / sept_social / BUIR
def _dense(z, name):
    return O.csr_to_dense(z[f"{name}_indptr"], z[f"{name}_indices"], z[f"{name}_data"], z[f"{name}_shape"])


def test_mhcn_motif_adjacency_and_layer_loop(golden):
    """univariate/mhcn.py:340-368 and :422-506, outputs of the reference's own methods (tests/golden/mhcn.npz)."""
    z = golden("mhcn.npz")
    n_u, n_i = int(z["n_users"]), int(z["n_items"])
    S = np.zeros((n_u, n_u))
    S[z["S_row"], z["S_col"]] = 1.0
    Y = np.zeros((n_u, n_i))
    Y[z["Y_row"], z["Y_col"]] = 1.0
    H = O.mhcn_motif_adjacency(S, Y)
    for name, h in zip(("H_s", "H_j", "H_p"), H):
        ref = _dense(z, name)
        assert np.array_equal(h != 0, ref != 0), name          # integer structure: bit-exact
        np.testing.assert_allclose(h, ref, rtol=2e-6, atol=1e-7)
    R = _dense(z, "R")
    gw = [z[f"gw{c}"].astype(np.float64) for c in (1, 2, 3, 4)]
    gb = [z[f"gb{c}"].astype(np.float64) for c in (1, 2, 3, 4)]
    fu, fi = O.mhcn_layer_loop(_dense(z, "H_s"), _dense(z, "H_j"), _dense(z, "H_p"), R, z["user_emb"].astype(np.float64),
                               z["item_emb"], gw, gb, z["attention"].astype(np.float64),
                               z["attention_mat"].astype(np.float64), int(z["n_layers"]))
    np.testing.assert_allclose(fu, z["final_user"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(fi, z["final_item"], rtol=2e-5, atol=2e-6)
    ss = 0.0
    for c, name in enumerate(("H_s", "H_j", "H_p")):
        em = O.mhcn_self_gating(fu, z[f"sgw{c + 1}"].astype(np.float64), z[f"sgb{c + 1}"].astype(np.float64))
        ss += O.mhcn_hierarchical_self_supervision(em, _dense(z, name), z["perms"][3 * c:3 * c + 3])
    assert float(z["ss_rate"]) * ss == pytest.approx(float(z["ss_loss"]), rel=2e-5)


def test_sept_social_encoder_and_neighbor_discrimination(golden):
    """univariate/sept_social.py:370-385 (sum of row-normalised layers) and :408-420."""
    z = golden("sept_social.npz")
    n_u = int(z["n_users"])
    rp, ci, va = z["norm_adj_indptr"], z["norm_adj_indices"], z["norm_adj_data"]
    final, _ = O.lgcn_encoder_forward(rp, ci, va, z["ego"], int(z["n_layers"]), combine="sum", layer_norm=True)
    np.testing.assert_allclose(final[:n_u], z["rec_user"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(final[n_u:], z["rec_item"], rtol=2e-5, atol=2e-6)
    for name, key in (("social", "friend_view"), ("sharing", "sharing_view")):
        v, _ = O.lgcn_encoder_forward(z[f"{name}_indptr"], z[f"{name}_indices"], z[f"{name}_data"], z["ego"][:n_u],
                                      int(z["n_layers"]), combine="sum", layer_norm=True)
        np.testing.assert_allclose(v, z[key], rtol=2e-5, atol=2e-6)
    uniq = np.unique(z["u_idx"])
    loss = O.neighbor_discrimination(z["positive"], z["friend_view"][uniq], z["aug_user"][uniq])
    assert loss == pytest.approx(float(z["nd_loss"]), rel=2e-5)


def test_buir_sparse_dropout(golden):
    """univariate/buir.py:300-309 + the forward that consumes it (:311-326)."""
    z = golden("buir.npz")
    rate, n = float(z["rate"]), int(z["n_users"]) + int(z["n_items"])
    assert np.array_equal(z["keep"], np.floor(1 - rate + z["rand"]).astype(bool))
    vals = O.sparse_dropout_values(z["adj_val"], z["keep"], rate)
    # the reference's dropped operator, coalesced, equals the kept entries with rescaled values
    got = np.zeros((n, n))
    np.add.at(got, (z["adj_row"], z["adj_col"]), vals)
    ref = np.zeros((n, n))
    np.add.at(ref, (z["dropped_row"], z["dropped_col"]), z["dropped_val"].astype(np.float64))
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-9)
    rowptr, col, val, order = O.coo_to_csr_stable(z["adj_row"], z["adj_col"], z["adj_val"], n)
    final, _ = O.lgcn_encoder_forward(rowptr, col, val * z["keep"][order] / (1 - rate), z["x"], int(z["n_layers"]), "mean")
    np.testing.assert_allclose(final, z["final"], rtol=2e-5, atol=2e-6)
