"""Empty, ragged and degenerate inputs through the entry points added in round 2 (the round-1 ops have theirs in
their own test modules): nothing may fault, sizes of zero are no-ops, unsupported shapes are refused loudly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_flash_forward_empty_anchor_set_and_single_row_table():
    from recommendation_amd import functional as Fn
    b = torch.randn(5, 64, device=DEV)
    lse, o = Fn.infonce_fwd_o_raw(torch.empty(0, 64, device=DEV), None, b, None, 5.0)
    assert lse.shape == (0,) and o.shape == (0, 64)
    a = torch.randn(3, 64, device=DEV)
    lse, o = Fn.infonce_fwd_o_raw(a, None, b[:1].contiguous(), None, 5.0)        # one candidate: softmax = 1, o = that row
    assert torch.allclose(o, b[:1].expand(3, 64), rtol=1e-6, atol=1e-6)
    assert torch.allclose(lse, 5.0 * (a @ b[0]), rtol=1e-5, atol=1e-5)
    assert not Fn.infonce_fwd_o_supported(256)                                    # d = 256 stays on the two-launch path
    with pytest.raises(Exception):
        Fn.infonce_fwd_o_raw(torch.randn(4, 256, device=DEV), None, torch.randn(4, 256, device=DEV), None, 5.0)


def test_infonce_stats_grad_modes_pick_the_right_forward(monkeypatch):
    """no_grad / non-differentiable anchors must not pay for the weighted row sum."""
    from recommendation_amd import functional as Fn
    calls = []
    real = Fn.infonce_fwd_o_raw
    monkeypatch.setattr(Fn, "infonce_fwd_o_raw", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    a, b = torch.randn(40, 64, device=DEV, requires_grad=True), torch.randn(90, 64, device=DEV, requires_grad=True)
    with torch.no_grad():
        Fn.infonce_stats(a, b, None, 0.2)
    Fn.infonce_stats(a.detach(), b, None, 0.2)
    assert not calls
    lse, pl = Fn.infonce_stats(a, b, None, 0.2)
    assert len(calls) == 1
    (lse - pl).sum().backward()
    assert torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all()


def test_dual_spmm_empty_graph_and_odd_width():
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    g = ra.CsrGraph(np.zeros(6, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), 5, 7, DEV)
    raw, nrm = Fn.spmm_l2norm_dual(g, torch.randn(7, 33, device=DEV))
    assert float(raw.abs().max()) == 0.0 and float(nrm.abs().max()) == 0.0
    rng = np.random.default_rng(0)
    g2 = ra.CsrGraph.from_coo(rng.integers(0, 9, 40), rng.integers(0, 11, 40), rng.standard_normal(40).astype(np.float32), 9, 11, DEV)
    x = torch.randn(11, 1, device=DEV)                                            # d = 1
    raw, nrm = Fn.spmm_l2norm_dual(g2, x)
    assert torch.allclose(nrm, torch.nn.functional.normalize(raw, dim=1), atol=1e-6)


def test_fused_rank_zero_queries_and_unsupported_shapes():
    from recommendation_amd import _lib
    from recommendation_amd.evaluate import rank_topk
    L = _lib.lib()
    assert not L.gcr_rank_fused_supported(1000, 64, 50)          # small catalogue -> two-call path
    assert not L.gcr_rank_fused_supported(100000, 256, 50)
    assert not L.gcr_rank_fused_supported(100000, 64, 300)
    ue, ie = torch.randn(10, 64, device=DEV), torch.randn(20000, 64, device=DEV)
    items, scores = rank_topk(ue, ie, torch.zeros(0, dtype=torch.int64), None, None, 10)
    assert items.shape == (0, 10)
    # an out-of-range user id is never dereferenced: its row comes back through the exact fallback, all padding
    items, scores = rank_topk(ue, ie, torch.tensor([3, 99, -1]), None, None, 10)
    assert int(items[0].min()) >= 0
    ref = torch.topk(ue[3] @ ie.T, 10).indices
    assert torch.equal(items[0].cpu(), ref.cpu())


def test_rank_metrics_users_without_test_items_and_padding():
    from recommendation_amd.evaluate import ranking_metrics
    top = torch.tensor([[5, 3, -1, -1], [1, 2, 3, 4], [7, 7, 7, 7]], device=DEV)
    rowptr = torch.tensor([0, 2, 2, 3])                                           # user 1 has no test item: not evaluated
    test_items = np.array([3, 9, 7], dtype=np.int32)
    m = ranking_metrics(top, rowptr, test_items, [2, 4])
    assert m[2]["Hit Ratio"] == round(2 / 3, 5)                                   # user 0 hits item 3, user 2 hits 7 once (set)
    assert m[4]["Precision"] == round(2 / (2 * 4), 5)
    assert m[2]["Recall"] == round((1 / 2 + 1 / 1) / 2, 5)


def test_adam_mask_spgemm_degenerate_sizes():
    from recommendation_amd import functional as Fn, graph_ops as G
    from recommendation_amd.optim import FusedAdam
    for n in (1, 2, 3, 5, 7):                                                     # tails of the float4 loop
        p = torch.nn.Parameter(torch.randn(n, device=DEV))
        q = torch.nn.Parameter(p.detach().clone())
        p.grad = q.grad = torch.randn(n, device=DEV)
        FusedAdam([p], lr=0.1).step()
        torch.optim.Adam([q], lr=0.1).step()
        assert torch.allclose(p, q, rtol=1e-6, atol=1e-7)
    y, _ = Fn.feature_masking(torch.empty(0, 64, device=DEV), 0.5, 1)
    assert y.shape == (0, 64)
    y, bits = Fn.feature_masking(torch.ones(3, 5, device=DEV), 1.0, 1)            # pf = 1: every column dropped
    assert float(y.abs().max()) == 0.0
    e = G.Sp.from_coo(torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64), torch.zeros(0), 4, 4, DEV)
    a = G.Sp.from_coo(torch.tensor([0, 1]), torch.tensor([1, 2]), torch.ones(2), 4, 4, DEV)
    assert (e @ a).nnz == 0 and (a @ e).nnz == 0 and (a * e).nnz == 0
    assert (a - a).nnz == 0                                                       # exact zeros are dropped, as scipy does
    assert torch.equal((a + e).to_dense(), a.to_dense())
