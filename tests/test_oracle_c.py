"""Pins the C restatement (oracle/oracle.c) to the numpy oracle, which is itself pinned to the
reference-generated golden fixtures (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from oracle import oracle_c as C
from oracle import oracle_np as O


def _graph(seed=0, n_u=300, n_i=120, e=4000):
    u, i = O.synthetic_interactions(n_u, n_i, e, seed=seed)
    rowptr, col, val = O.norm_adj_csr(u, i, n_u, n_i)
    return u, i, rowptr, col, val, n_u + n_i


def test_spmm_and_propagate_match_numpy():
    _, _, rowptr, col, val, n = _graph()
    x = np.random.default_rng(0).standard_normal((n, 64)).astype(np.float32)
    ref = O.spmm_csr(rowptr, col, val, x)
    np.testing.assert_allclose(C.spmm_csr(rowptr, col, val, x), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    got = C.spmm_coo(rows, col, val, n, x)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    for comb in ("mean", "sum"):
        ref, _ = O.lgcn_encoder_forward(rowptr, col, val, x, 3, combine=comb)
        got = C.lightgcn_propagate(rowptr, col, val, x, 3, comb)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())


def test_golden_propagation_through_c(golden):
    g, p = golden("graph_build.npz"), golden("propagation.npz")
    got = C.lightgcn_propagate(g["norm_indptr"], g["norm_indices"], g["norm_data"], p["xs"], 3, "mean")
    np.testing.assert_allclose(got, p["norm_mean_K3"], rtol=2e-5, atol=2e-5 * np.abs(p["norm_mean_K3"]).max())


@pytest.mark.parametrize("normalize", [True, False])
def test_row_lse_matches_numpy(normalize):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((70, 64)).astype(np.float32) * 0.4
    b = rng.standard_normal((333, 64)).astype(np.float32) * 0.4
    pos = rng.integers(0, 333, 70)
    lse, s = O.row_lse_scores(a, b, 5.0, normalize)
    got_lse, got_pos = C.row_lse(a, b, pos, 5.0, normalize)
    np.testing.assert_allclose(got_lse, lse, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(got_pos, s[np.arange(70), pos], rtol=1e-4, atol=1e-5)


def test_bpr_matches_numpy(golden):
    b = golden("bpr.npz")
    ut, it, u, i, j, j3 = (b[k] for k in ("user_tab", "item_tab", "u_idx", "i_idx", "j_idx", "j_idx3"))
    for var in (O.BPR_NCL, O.BPR_LOGSIGMOID, O.BPR_LOG_SIGMOID):
        assert C.bpr_loss(ut, it, u, i, j, var) == pytest.approx(O.bpr_loss(ut, it, u, i, j, var), rel=2e-6)
    assert C.bpr_loss(ut, it, u, i, j3, 2) == pytest.approx(O.bpr_loss(ut, it, u, i, j3, 2), rel=2e-6)
    assert C.bpr_loss(ut, it, u, i, j, 0) == pytest.approx(float(b["ncl_bpr_loss"]), rel=2e-5)


def test_sampler_and_mask_bit_exact_with_numpy():
    u, i = O.synthetic_interactions(80, 50, 900, seed=4)
    order = np.lexsort((i, u))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=80))])
    items_sorted = i[order].astype(np.int32)
    ub = u[:200]
    for n_negs, trials in ((1, 101), (3, 101), (1, 0), (2, 2)):
        ref = O.neg_sample_uniform(rowptr, items_sorted, ub, n_negs, 50, seed=0x1234567890AB, offset=2**33 + 7, max_trials=trials)
        got = C.neg_sample(rowptr, items_sorted, ub, n_negs, 50, 0x1234567890AB, 2**33 + 7, trials)
        assert np.array_equal(ref, got)
    assert np.array_equal(C.edge_keep_mask(10001, 0.25, 77), O.edge_keep_mask(10001, 0.25, 77))
