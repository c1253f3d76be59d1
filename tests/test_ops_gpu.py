"""The stages either side of the hot path on the GPU:
  FusedAdam (gcr_adam_step_f32)             vs torch.optim.Adam, the optimiser the reference uses (ncl.py:305, gcl.py:201)
  feature masking (gcr_mask_columns_f32)    vs the reference's own drop_feature output (tests/golden/featmask.npz)
  motif adjacency (gcr_spgemm_expand_f32 ..) vs the reference's own build_hyper_adj_mats output (tests/golden/mhcn.npz)"""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wd", [0.0, 1e-4])
def test_fused_adam_matches_torch_adam(wd):
    from recommendation_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(1)
    p0 = torch.randn(1000, 64, device="cuda", generator=g)
    pa, pb = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    ref = torch.optim.Adam([pa], lr=1e-2, weight_decay=wd)
    opt = FusedAdam([pb], lr=1e-2, weight_decay=wd)
    for _ in range(6):
        gr = torch.randn(1000, 64, device="cuda", generator=g)
        pa.grad, pb.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
    assert float((pa - pb).abs().max()) <= 2e-6 * float(pa.abs().max())
    assert float((ref.state[pa]["exp_avg_sq"] - opt.state[pb]["exp_avg_sq"]).abs().max()) <= 1e-6
    # gradient pieces summed inside the kernel == accumulated beforehand
    pc, pd = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone())
    o1, o2 = FusedAdam([pc], lr=1e-2), FusedAdam([pd], lr=1e-2)
    g1, g2, g3 = (torch.randn(1000, 64, device="cuda", generator=g) for _ in range(3))
    pc.grad = g1 + g2 + g3
    pd.grad = g1
    o1.step()
    o2.step(extra_grads={pd: [g2, g3]})
    assert float((pc - pd).abs().max()) <= 1e-6


def test_feature_masking_matches_reference(golden):
    """univariate/grace.py:261-278: whole columns zeroed, the rest untouched; the recorded mask replayed through the
    kernel reproduces the reference output bit for bit; the device draw is the oracle's Philox stream."""
    from recommendation_amd import functional as Fn
    z = golden("featmask.npz")
    x = torch.from_numpy(z["x"]).cuda()
    for k in range(3):
        y = z[f"y{k}"]
        dropped = (y == 0).all(0) & ~(z["x"] == 0).all(0)
        bits = Fn.pack_bits(torch.from_numpy(~dropped).cuda())
        xt = x.clone().requires_grad_(True)
        got, _ = Fn.feature_masking(xt, float(z[f"pf{k}"]), 0, keep_bits=bits)
        assert np.array_equal(got.detach().cpu().numpy(), y)
        got.sum().backward()
        assert np.array_equal(xt.grad.cpu().numpy(), np.broadcast_to((~dropped).astype(np.float32), y.shape))
    d, pf, seed = 256, 0.3, 77
    xx = torch.ones(5, d, device="cuda")
    got, bits = Fn.feature_masking(xx, pf, seed)
    keep = O.edge_keep_mask(d, pf, seed)
    assert np.array_equal(got[0].cpu().numpy() != 0, keep)
    assert abs(keep.mean() - (1 - pf)) < 0.1


def _dense(z, name):
    return O.csr_to_dense(z[f"{name}_indptr"], z[f"{name}_indices"], z[f"{name}_data"], z[f"{name}_shape"])


def test_motif_adjacency_matches_reference(golden):
    """univariate/mhcn.py:340-368 on the device: sparse x sparse products, masked products, sums, `> 3` filter and
    the row normalisation — the reference's own H_s / H_j / H_p (structure bit-exact, values to f32 rounding)."""
    from recommendation_amd import graph_ops as G
    z = golden("mhcn.npz")
    n_u, n_i = int(z["n_users"]), int(z["n_items"])
    hs, hj, hp, r = G.build_hyper_graphs(z["S_row"], z["S_col"], z["Y_row"], z["Y_col"], n_u, n_i, "cuda")
    for name, g in (("H_s", hs), ("H_j", hj), ("H_p", hp), ("R", r)):
        ref = _dense(z, name)
        got = np.zeros_like(ref)
        rows = np.repeat(np.arange(g.n_rows), np.diff(g.rowptr_host))
        got[rows, g.col.cpu().numpy()] = g.val.cpu().numpy()
        assert np.array_equal(got != 0, ref != 0), name
        np.testing.assert_allclose(got, ref, rtol=2e-6, atol=1e-7, err_msg=name)
    # algebra spot checks against dense numpy
    rng = np.random.default_rng(0)
    a = (rng.random((40, 30)) < 0.2) * rng.integers(1, 4, (40, 30))
    b = (rng.random((30, 50)) < 0.2) * rng.integers(1, 4, (30, 50))
    m = (rng.random((40, 50)) < 0.3).astype(np.float64)

    def sp(x):
        r_, c_ = np.nonzero(x)
        return G.Sp.from_coo(torch.from_numpy(r_), torch.from_numpy(c_), torch.from_numpy(x[r_, c_].astype(np.float32)),
                             x.shape[0], x.shape[1], "cuda")

    A, B, M = sp(a), sp(b), sp(m)
    assert np.array_equal((A @ B).to_dense().cpu().numpy(), a @ b)
    assert np.array_equal(((A @ B) * M).to_dense().cpu().numpy(), (a @ b) * m)
    assert np.array_equal((A @ B - M).to_dense().cpu().numpy(), a @ b - m)
    assert np.array_equal(A.T.to_dense().cpu().numpy(), a.T)
    assert ((A @ B - M).val != 0).all()


@pytest.mark.parametrize("n,d,b", [(1000, 64, 4096), (37, 32, 5), (500, 128, 300), (10, 48, 0)])
def test_gather_rows_forward_and_scatter_backward(n, d, b):
    """Fn.gather_rows == table[idx]; its backward == the dense index_put(accumulate) gradient (duplicates add up)."""
    from recommendation_amd import functional as Fn
    g = torch.Generator(device="cuda").manual_seed(n + b)
    t0 = torch.randn(n, d, device="cuda", generator=g)
    idx = torch.randint(0, n, (b,), device="cuda", generator=g)
    w = torch.randn(b, d, device="cuda", generator=g)
    ta, tb = t0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    ya, yb = ta[idx], Fn.gather_rows(tb, idx)
    assert torch.equal(ya, yb)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    tol = 1e-6 * max(float(ta.grad.abs().max()), 1.0) if b else 0.0
    assert float((ta.grad - tb.grad).abs().max()) <= tol


@pytest.mark.parametrize("d", [64, 32, 128, 100])
@pytest.mark.parametrize("with_raw", [False, True])
def test_normalize_bwd_n_one_pass(d, with_raw):
    """gcr_normalize_bwd_n_f32 = (g - n <n, g>) * inv (+ g_raw) against float64, incl. an eps-clamped (all-zero) row and
    in-place output."""
    from recommendation_amd import functional as Fn
    rng = np.random.default_rng(d)
    z = rng.standard_normal((1003, d)).astype(np.float32)
    z[17] = 0.0
    norm = np.maximum(np.linalg.norm(z.astype(np.float64), axis=1, keepdims=True), 1e-12)
    n, inv = (z / norm).astype(np.float32), (1.0 / norm[:, 0]).astype(np.float32)
    g = rng.standard_normal((1003, d)).astype(np.float32)
    graw = rng.standard_normal((1003, d)).astype(np.float32) if with_raw else None
    n64, g64 = n.astype(np.float64), g.astype(np.float64)
    ref = (g64 - n64 * (n64 * g64).sum(1, keepdims=True)) * inv.astype(np.float64)[:, None]
    if with_raw:
        ref = ref + graw
    tn, ti, tg = (torch.from_numpy(a).cuda() for a in (n, inv, g))
    tr = torch.from_numpy(graw).cuda() if with_raw else None
    out = Fn.normalize_bwd_n(tn, ti, tg, tr)
    live = np.arange(1003) != 17                              # (row 17: inv = 1e12 times a rounding residue: not compared)
    np.testing.assert_allclose(out.cpu().numpy()[live], ref[live], rtol=1e-5, atol=1e-5 * np.abs(ref[live]).max())
    assert torch.equal(Fn.normalize_bwd_n(tn, ti, tg.clone(), tr, out=tg), out)


@pytest.mark.parametrize("n", [1, 255, 1000, 70001])
@pytest.mark.parametrize("dx,dg", [(64, 64), (32, 128), (96, 32), (128, 128)])
def test_gram_tn(n, dx, dg):
    """x^T g with the rows split over the chip (gcr_gram_tn_f32: the weight gradient of MHCN's `em @ W`) vs float64."""
    from recommendation_amd import functional as Fn
    rng = np.random.default_rng(n + dx)
    x = rng.standard_normal((n, dx)).astype(np.float32)
    g = rng.standard_normal((n, dg)).astype(np.float32)
    out = Fn.gram_tn(torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda())
    ref = x.astype(np.float64).T @ g.astype(np.float64)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * np.sqrt(n) * 4)
    again = Fn.gram_tn(torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda())
    assert torch.equal(out, again)                            # fixed-order partial sums: bitwise reproducible


def test_dense_proj_gradients():
    from recommendation_amd import functional as Fn
    torch.manual_seed(0)
    em = torch.randn(5000, 64, device="cuda", requires_grad=True)
    w = torch.randn(64, 64, device="cuda", requires_grad=True)
    up = torch.randn(5000, 64, device="cuda")
    (Fn.dense_proj(em, w) * up).sum().backward()
    ge, gw = em.grad.clone(), w.grad.clone()
    em.grad = w.grad = None
    ((em.double() @ w.double()) * up.double()).sum().backward()
    np.testing.assert_allclose(ge.cpu().numpy(), em.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(em.grad.abs().max()))
    np.testing.assert_allclose(gw.cpu().numpy(), w.grad.cpu().numpy(), rtol=1e-4, atol=1e-5 * float(w.grad.abs().max()))


@pytest.mark.parametrize("n,d", [(1, 64), (1000, 64), (70001, 64), (5000, 48), (3000, 128), (777, 3)])
def test_rows_dot_vec_gradients(n, d):
    """`em @ v` with gcr_weighted_colsum_f32 as the gradient of v, vs float64."""
    from recommendation_amd import functional as Fn
    torch.manual_seed(n)
    em = torch.randn(n, d, device="cuda", requires_grad=True)
    v = torch.randn(d, device="cuda", requires_grad=True)
    up = torch.randn(n, device="cuda")
    out = Fn.rows_dot_vec(em, v)
    (out * up).sum().backward()
    ge, gv = em.grad.clone(), v.grad.clone()
    em.grad = v.grad = None
    ref = em.double() @ v.double()
    (ref * up.double()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ge.cpu().numpy(), em.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gv.cpu().numpy(), v.grad.cpu().numpy(), rtol=1e-4, atol=2e-6 * np.sqrt(n) * 4)


@pytest.mark.parametrize("n,d,bias", [(1, 64, True), (1000, 64, True), (70001, 64, True), (5000, 48, True), (3000, 128, False),
                                       (777, 3, True), (513, 256, True)])
def test_gate_matches_the_torch_expression(n, d, bias):
    """Fn.gate = em * sigmoid(z + bias) (mhcn.py:404-411) forward and all three gradients vs the float64 expression."""
    from recommendation_amd import functional as Fn
    torch.manual_seed(n + d)
    em = torch.randn(n, d, device="cuda", requires_grad=True)
    z = (3 * torch.randn(n, d, device="cuda")).requires_grad_(True)
    b = torch.randn(1, d, device="cuda", requires_grad=True) if bias else None
    up = torch.randn(n, d, device="cuda")
    out = Fn.gate(em, z, b)
    (out * up).sum().backward()
    got = [out.detach(), em.grad.clone(), z.grad.clone()] + ([b.grad.clone()] if bias else [])
    em.grad = z.grad = None
    if bias:
        b.grad = None
    ref = em.double() * torch.sigmoid(z.double() + (b.double() if bias else 0.0))
    (ref * up.double()).sum().backward()
    want = [ref.detach(), em.grad, z.grad] + ([b.grad] if bias else [])
    for k, (a, w) in enumerate(zip(got, want)):
        assert a.shape == w.shape
        tol = 2e-6 * float(w.abs().max()) * (np.sqrt(n) if k == 3 else 1.0) + 1e-30
        assert float((a.double() - w.double()).abs().max()) <= tol, k


@pytest.mark.parametrize("n,d,extra", [(1, 64, True), (1000, 64, True), (70001, 64, False), (4097, 32, True), (3000, 128, True),
                                        (513, 256, False), (2000, 48, True)])
def test_channel_mix_matches_the_torch_expression(n, d, extra):
    """Fn.channel_mix (mhcn.py:413-420 + the `+ simple / 2` of :443): mixed, score and the gradients of the three channel
    tables, of v and of the extra table vs the float64 expression.  d = 48 takes the torch composition."""
    from recommendation_amd import functional as Fn
    torch.manual_seed(n + d)
    es = [torch.randn(n, d, device="cuda", requires_grad=True) for _ in range(3)]
    v = (0.5 * torch.randn(d, device="cuda")).requires_grad_(True)
    ex = torch.randn(n, d, device="cuda", requires_grad=True) if extra else None
    up = torch.randn(n, d, device="cuda")
    mixed, score = Fn.channel_mix(*es, v, extra=ex, extra_scale=0.5 if extra else 0.0)
    (mixed * up).sum().backward()
    leaves = es + [v] + ([ex] if extra else [])
    got = [mixed.detach(), score.detach()] + [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    logits = torch.stack([e.double() @ v.double() for e in es])
    sc = torch.softmax(logits, dim=0)
    ref = sum(sc[k].unsqueeze(1) * es[k].double() for k in range(3))
    if extra:
        ref = ref + 0.5 * ex.double()
    (ref * up.double()).sum().backward()
    want = [ref.detach(), sc.detach()] + [t.grad for t in leaves]
    for k, (a, w) in enumerate(zip(got, want)):
        assert a.shape == w.shape
        # (d v sums q_k e_k with q_k = score_k (<g, e_k> - mean): its error scales with |<g, e_k>| |e| ~ O(10), not with the
        # result, which cancels to ~0 on a peaked softmax — hence the floor of 1 on its scale)
        tol = 4e-6 * (max(float(w.abs().max()), 1.0) * np.sqrt(n) if k == 5 else float(w.abs().max())) + 1e-30
        assert float((a.double() - w.double()).abs().max()) <= tol, k
