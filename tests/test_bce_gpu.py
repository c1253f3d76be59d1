"""GPU parity of the BCE-with-logits branch of LightGCN's training step (lightgcn.py:109-113, `loss_type == "bce"`):
the fused softplus row sums and their sigmoid-weighted gradients against the reference-generated goldens
(tests/golden/bpr.npz `lgcn_bce*`, lifted from the reference's own statements) and the float64 oracle, on the three engines
(two f16 planes on rows scaled inside the launch / three bf16 planes / f32 MFMA), every supported width, ragged shapes, and the full-batch edge-list form."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

ENGINES = ("auto", "b3", "f32")       # two f16 planes (rows scaled inside) / three bf16 planes / f32 MFMA


@pytest.fixture(scope="module")
def Fn():
    from recommendation_amd import functional
    return functional


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(grad)


def _close(t, ref, rel=1e-5):
    got = t.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=rel * 10, atol=rel * max(np.abs(ref).max(), 1e-30))



@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("name,ukey,ikey", [("lgcn_bce", "user_tab", "item_tab"), ("lgcn_bce_big", "user_tab_big", "item_tab_big")])
def test_lightgcn_bce_golden(golden, engine, name, ukey, ikey):
    """The reference's own loss block (lightgcn.py:95-118, loss_type "bce", reg_weight 1e-4): value and both gradients."""
    from recommendation_amd import losses
    b = golden("bpr.npz")
    ut, it = _t(b[ukey], True), _t(b[ikey], True)
    u, i = _t(b["u_idx"]), _t(b["i_idx"])
    loss = losses.lightgcn_bce_loss(ut, it, u, i, engine=engine) + 1e-4 * (ut[u].norm(2).pow(2) + it[i].norm(2).pow(2))
    assert float(loss.detach()) == pytest.approx(float(b[f"{name}_loss"]), rel=1e-5)
    loss.backward()
    _close(ut.grad, b[f"{name}_gu"])
    _close(it.grad, b[f"{name}_gi"])


@pytest.mark.parametrize("engine", ENGINES)
def test_lightgcn_bce_small_batch_gathers_rows(golden, engine):
    """fewer samples than users: the softplus part runs over the gathered batch rows (duplicates included)."""
    from recommendation_amd import losses
    b = golden("bpr.npz")
    u, i = b["u_idx"][:23], b["i_idx"][:23]
    ut, it = _t(b["user_tab_big"], True), _t(b["item_tab_big"], True)
    loss = losses.lightgcn_bce_loss(ut, it, _t(u), _t(i), engine=engine)
    ref, gu, gi = O.lightgcn_bce_loss(b["user_tab_big"], b["item_tab_big"], u, i)
    assert float(loss.detach()) == pytest.approx(ref, rel=1e-5)
    loss.backward()
    _close(ut.grad, gu)
    _close(it.grad, gi)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("d", [32, 64, 128, 256, 48])
@pytest.mark.parametrize("m,n", [(1, 1), (7, 33), (257, 1682), (1000, 95), (130, 4100)])
def test_softplus_rowsum_and_grads(Fn, engine, d, m, n):
    """rowsum_i = sum_j softplus(<a_i, b_j>) and the gradients of sum_i w_i rowsum_i (weights of both signs) vs float64;
    scores from -25 to 25: the softplus of a strongly negative score keeps its e^s."""
    rng = np.random.default_rng(m * 1000 + n + d)
    a = (rng.standard_normal((m, d)) * (2.5 / np.sqrt(d))).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 2.0).astype(np.float32)
    a[0] *= 4.0
    w = rng.standard_normal(m)
    at, bt = _t(a, True), _t(b, True)
    rows = Fn.bce_softplus_rowsum(at, bt, engine=engine)
    ref_rows, sig = O.bce_rows(a, b)
    _close(rows, ref_rows)
    with torch.no_grad():                                  # without a gradient the forward skips its second product
        rows_ng = Fn.bce_softplus_rowsum(at, bt, engine=engine)
    assert torch.equal(rows_ng, rows.detach())
    (rows * _t(w.astype(np.float32))).sum().backward()
    w32 = w.astype(np.float32).astype(np.float64)
    _close(at.grad, (sig * w32[:, None]) @ b.astype(np.float64))
    _close(bt.grad, (sig * w32[:, None]).T @ a.astype(np.float64))


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("regime,shift", [("moderate", 0.0), ("converged", 11.5), ("very_negative", 20.0)])
def test_second_products_in_three_score_regimes(Fn, engine, regime, shift):
    """o_i = sum_j sigmoid(s_ij) b_j and g_j = sum_i w_i sigmoid(s_ij) a_i against float64 where the scores sit around 0,
    around -11.5 (a converged one-hot BCE: sigmoid ~ 1e-5) and around -20 (sigmoid ~ 2e-9 everywhere), weights spread over two
    orders of magnitude like degree / (E I): 1e-5 of the largest entry on every engine.  (The two-plane engine keeps a running
    power-of-two reference per output row for its f16 probability planes: without it the last regime read 2.5-4e-5.)"""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    m, n, d = 1 << 15, 4096, 64
    a = torch.randn(m, d, device=dev, generator=g) * 0.12
    b = torch.randn(n, d, device=dev, generator=g) * 0.12
    a[:, 0] = shift ** 0.5
    b[:, 0] = -(shift ** 0.5)
    w = torch.exp(torch.rand(m, device=dev, generator=g) * 4.6) * 1e-12
    sg = torch.sigmoid(a.double() @ b.double().T)
    o_ref = sg @ b.double()
    g_ref = (sg * w.double().unsqueeze(1)).T @ a.double()
    fl = Fn._bce_flags(engine)
    if engine == "f32":                                   # (no fused o on the f32 MFMA: the same sum as a backward launch)
        o = Fn.bce_bwd_raw(a, b, w_x=torch.ones(m, device=dev), engine_flag=fl)
    else:
        _, o = Fn.bce_fwd_raw(a, b, want_o=True, engine_flag=fl)
    gj = Fn.bce_bwd_raw(b, a, w_y=w, engine_flag=fl)
    assert float((o.double() - o_ref).abs().max()) <= 1e-5 * float(o_ref.abs().max())
    assert float((gj.double() - g_ref).abs().max()) <= 1e-5 * float(g_ref.abs().max())
    # every output row against its own largest entry (a row whose sigmoids are all tiny must not be flushed)
    assert float(((o.double() - o_ref).abs().amax(1) / o_ref.abs().amax(1)).max()) <= 2e-5
    assert float(((gj.double() - g_ref).abs().amax(1) / g_ref.abs().amax(1)).max()) <= 2e-5


@pytest.mark.parametrize("engine", ENGINES)
def test_softplus_rows_far_below_zero(Fn, engine):
    """every score around -12 .. -30: softplus(s) ~ e^s must not collapse to 0 (1 + e^s rounds to 1 in f32)."""
    rng = np.random.default_rng(5)
    a = np.abs(rng.standard_normal((70, 64))).astype(np.float32) * 0.5
    b = -np.abs(rng.standard_normal((300, 64))).astype(np.float32) * 0.9
    rows = Fn.bce_softplus_rowsum(_t(a), _t(b), engine=engine)
    ref, _ = O.bce_rows(a, b)
    assert ref.max() < 1e-2
    np.testing.assert_allclose(rows.cpu().numpy(), ref, rtol=2e-5)


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("d", [64, 32, 128])
def test_bce_edge_loss_full_batch(Fn, engine, d):
    """LightGCN.loss(loss_type="bce") on the graph's own edge list (duplicates, an isolated user) = the reference's
    [E, I] formulation in float64, value and the gradient w.r.t. the stacked table."""
    from recommendation_amd import CsrGraph
    rng = np.random.default_rng(11)
    n_u, n_i, e = 90, 57, 1200
    u = rng.integers(0, n_u - 1, e)                       # user n_u - 1 has no edge
    i = rng.integers(0, n_i, e)
    u[-40:], i[-40:] = u[:40], i[:40]                     # repeated pairs
    ei = O.build_edge_index(u, i, n_u)
    graph = CsrGraph.from_edge_index_gcn_norm(ei, n_u + n_i, "cuda", symmetric=True)
    tab = (rng.standard_normal((n_u + n_i, d)) * (1.5 / np.sqrt(d)) * 2).astype(np.float32)
    t = _t(tab, True)
    old = Fn.INFONCE_ENGINE
    Fn.INFONCE_ENGINE = engine
    try:
        loss = Fn.bce_edge_loss(graph, t, n_u)
        loss.backward()
    finally:
        Fn.INFONCE_ENGINE = old
    ue, ie = graph.user_major_edges(n_u)
    ref, gu, gi = O.lightgcn_bce_loss(tab[:n_u], tab[n_u:], ue.cpu().numpy(), ie.cpu().numpy())
    assert float(loss.detach()) == pytest.approx(ref, rel=1e-5)
    _close(t.grad, np.concatenate([gu, gi]))


def test_lightgcn_model_bce_step_cfg1():
    """BASELINE configs[0] sizes (943 x 1682, 80 000 edges, K = 2, d = 64): `LightGCN.loss(edge_index, loss_type="bce")` =
    lightgcn.py:85-118 restated in float64 (propagation by the oracle, the [E, I] BCE in row slabs), loss and the
    gradients of both embedding tables."""
    from recommendation_amd.encoders import LightGCN
    n_u, n_i, e, k = 943, 1682, 80000, 2
    u, i = O.synthetic_interactions(n_u, n_i, e, seed=20250919)
    ei = O.build_edge_index(u, i, n_u)
    torch.manual_seed(0)
    model = LightGCN(n_u, n_i, 64, k).cuda()
    with torch.no_grad():                                 # trained-scale rows: scores of a few units
        model.user_embedding.weight.mul_(12.0)
        model.item_embedding.weight.mul_(12.0)
    edge_index = torch.from_numpy(ei).cuda()
    loss = model.loss(edge_index, loss_type="bce", reg_weight=1e-4)
    loss.backward()
    uw = model.user_embedding.weight.detach().cpu().numpy()
    iw = model.item_embedding.weight.detach().cpu().numpy()
    fu, fi = O.lightgcn_forward(ei, uw, iw, k)
    graph = model.prepare(edge_index)
    ue, ie = (x.cpu().numpy() for x in graph.user_major_edges(n_u))
    ref, gfu, gfi = O.lightgcn_bce_loss(fu, fi, ue, ie, reg_weight=1e-4)
    assert float(loss.detach()) == pytest.approx(ref, rel=1e-5)
    # back through the linear propagation: d x0 = sum_k (A^T)^k g (sum combine, lightgcn.py:26)
    rowptr, col, val = (graph.rowptr.cpu().numpy(), graph.col.cpu().numpy().astype(np.int64), graph.val.cpu().numpy().astype(np.float64))
    g = np.concatenate([gfu, gfi])
    acc, cur = g.copy(), g.copy()
    for _ in range(k):
        cur = O.spmm_backward(rowptr, col, val, cur, n_u + n_i)
        acc += cur
    _close(model.user_embedding.weight.grad, acc[:n_u])
    _close(model.item_embedding.weight.grad, acc[n_u:])


def test_unsupported_loss_type_raises():
    from recommendation_amd.encoders import LightGCN
    model = LightGCN(8, 8, 32, 1).cuda()
    ei = torch.tensor(O.build_edge_index(np.arange(8), np.arange(8), 8)).cuda()
    with pytest.raises(ValueError, match="Unsupported loss_type"):
        model.loss(ei, loss_type="hinge")
