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
