"""Rows widened beyond the first slice: MHCN-style row-normalised rectangular operators
(Graph.normalize_graph_mat, non-square branch) and sept_social's neighbour-discrimination loss."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def test_row_normalised_operator_matches_reference(golden):
    """selfcf.Graph.normalize_graph_mat on a rectangular matrix (selfcf.py:250-254 = ncl.py:37-41;
    univariate/mhcn.py:401-402 builds R with it): structure bit-exact, values at fp32 resolution, and
    the SpMM over it (+ its transpose for the backward) against the oracle."""
    import recommendation_amd as ra
    g = golden("rownorm.npz")
    n_rows, n_cols = int(g["n_rows"]), int(g["n_cols"])
    graph = ra.CsrGraph.row_normalised(g["row"], g["col"], g["val"], n_rows, n_cols, "cuda")
    assert np.array_equal(graph.rowptr.cpu().numpy(), g["indptr"])
    assert np.array_equal(graph.col.cpu().numpy().astype(np.int64), g["indices"])
    np.testing.assert_allclose(graph.val.cpu().numpy(), g["data"], rtol=3e-7)
    rowsum = np.add.reduceat(np.append(g["data"], 0.0), g["indptr"][:-1])[np.diff(g["indptr"]) > 0]
    np.testing.assert_allclose(rowsum, 1.0, rtol=1e-6)                      # D^-1 A: rows sum to one
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n_cols, 64)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    y = ra.functional.spmm(graph, xt)
    ref = O.spmm_csr(g["indptr"], g["indices"], g["data"], x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref, rtol=1e-5, atol=1e-6)
    w = rng.standard_normal((n_rows, 64)).astype(np.float32)
    (y * torch.from_numpy(w).cuda()).sum().backward()
    gref = O.spmm_backward(g["indptr"], g["indices"], g["data"], w, n_cols)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("b,k,d", [(300, 10, 64), (64, 5, 32), (1000, 10, 64)])
def test_neighbor_discrimination_matches_dense(b, k, d):
    """sept_social.py:408-420 restated densely in float64 (the module needs tensorflow to import)."""
    from recommendation_amd.losses import neighbor_discrimination
    rng = np.random.default_rng(b)
    emb = rng.standard_normal((b, d)).astype(np.float32)
    aug = (emb + 0.5 * rng.standard_normal((b, d))).astype(np.float32)
    positive = rng.integers(0, b, (b, k))
    e, a = O.row_l2_normalize(emb), O.row_l2_normalize(aug)
    pos = (e[:, None, :] * a[positive]).sum(2)
    ttl = e @ a.T
    ref = -np.log(np.exp(pos / 0.1).sum(1) / np.exp(ttl / 0.1).sum(1)).sum()
    et, at = torch.from_numpy(emb).cuda().requires_grad_(True), torch.from_numpy(aug).cuda().requires_grad_(True)
    loss = neighbor_discrimination(torch.from_numpy(positive).cuda(), et, at, 0.1)
    assert float(loss) == pytest.approx(ref, rel=1e-5)
    loss.backward()
    # gradient against torch autograd of the dense float64 formulation
    e64 = torch.from_numpy(emb).double().requires_grad_(True)
    a64 = torch.from_numpy(aug).double().requires_grad_(True)
    en, an = torch.nn.functional.normalize(e64, dim=1), torch.nn.functional.normalize(a64, dim=1)
    p64 = (en.unsqueeze(1) * an[torch.from_numpy(positive)]).sum(2)
    dense = -torch.log(torch.exp(p64 / 0.1).sum(1) / torch.exp(en @ an.T / 0.1).sum(1)).sum()
    dense.backward()
    for got, want in ((et.grad, e64.grad), (at.grad, a64.grad)):
        want = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * np.abs(want).max())


@pytest.mark.parametrize("engine", ["auto", "b3", "f32"])
@pytest.mark.parametrize("m", [7, 257])
def test_grace_dual_branch_infonce_matches_reference(golden, monkeypatch, engine, m):
    """losses.grace_infonce_loss against the reference's own DualBranchContrast outputs (values and both
    gradients), all three mask variants, both MFMA engines."""
    from recommendation_amd.losses import grace_infonce_loss
    from recommendation_amd import functional as _Fn
    monkeypatch.setattr(_Fn, "INFONCE_ENGINE", engine)
    g = golden("grace.npz")
    for tau in (0.2, 0.5):
        for intra, keep in ((0, 0), (1, 0), (1, 1)):
            h1 = torch.from_numpy(g[f"h1_{m}"]).cuda().requires_grad_(True)
            h2 = torch.from_numpy(g[f"h2_{m}"]).cuda().requires_grad_(True)
            loss = grace_infonce_loss(h1, h2, tau, intraview_negs=bool(intra), exclude_self=bool(keep))
            key = f"{m}_{tau}_{intra}_{keep}"
            assert float(loss) == pytest.approx(float(g[f"loss_{key}"]), rel=1e-5, abs=2e-6)
            loss.backward()
            for got, want in ((h1.grad, g[f"g1_{key}"]), (h2.grad, g[f"g2_{key}"])):
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-5 * np.abs(want).max() + 1e-9)


@pytest.mark.parametrize("engine", ["auto", "b3", "f32"])
@pytest.mark.parametrize("m,d", [(1, 64), (31, 64), (33, 64), (64, 64), (65, 32), (129, 64), (300, 128), (1000, 64), (2100, 64)])
def test_exclude_diagonal_lse_and_grads(monkeypatch, engine, m, d):
    """GCR_INFONCE_EXCLUDE_DIAGONAL on square self-similarity problems across tile boundaries: row LSE
    over j != i and its gradient against dense float64 torch."""
    from recommendation_amd import functional as Fn
    from recommendation_amd import functional as _Fn
    monkeypatch.setattr(_Fn, "INFONCE_ENGINE", engine)
    rng = np.random.default_rng(m + d)
    x = (rng.standard_normal((m, d)) * 0.5).astype(np.float32)
    w = rng.standard_normal(m)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    lse, _ = Fn.infonce_stats(xt, xt, None, 0.25, normalize=True, exclude_diagonal=True)
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    xn = torch.nn.functional.normalize(x64, dim=1)
    s = (xn @ xn.T) * 4.0
    s = s.masked_fill(torch.eye(m, dtype=torch.bool), float("-inf"))
    ref = torch.logsumexp(s, 1) if m > 1 else torch.full((1,), float("-inf"), dtype=torch.float64)
    if m == 1:
        assert float(lse[0]) == float("-inf")
        return
    np.testing.assert_allclose(lse.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-5)
    (lse * torch.from_numpy(w.astype(np.float32)).cuda()).sum().backward()
    (ref * torch.from_numpy(w)).sum().backward()
    want = x64.grad.numpy()
    np.testing.assert_allclose(xt.grad.cpu().numpy(), want, rtol=2e-4, atol=1e-5 * np.abs(want).max())
