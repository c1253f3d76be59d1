"""GPU parity of the MFMA InfoNCE kernels (through the C ABI) against the CPU oracle and the
reference-generated goldens.  Tolerance 1e-5 relative fp32 on the loss values (north_star)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_c as C
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Fn():
    from recommendation_amd import functional
    return functional


@pytest.fixture(autouse=True, params=["auto", "b3", "f32"])
def engine(request, monkeypatch, Fn):
    """Every test of this module runs three ways (csrc/gcr_infonce.hip), with the same tolerances: "auto", the
    library default (split-operand engine for d <= 128; its two-product launches on two f16 planes when the op
    normalises the rows itself and d <= 64 — the GCR_INFONCE_UNIT_ROWS promise); "b3", the promise withheld (three
    bf16 planes everywhere); "f32", the f32 MFMA engine (GCR_INFONCE_ENGINE_F32).  The host resolves the flags once
    per forward and hands them to the backward."""
    monkeypatch.setattr(Fn, "INFONCE_ENGINE", request.param)
    return request.param


def _lse(Fn, a, b, inv_tau, normalize):
    at, bt = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    sa = Fn.row_inv_norm(at) if normalize else None
    sb = Fn.row_inv_norm(bt) if normalize else None
    return Fn.infonce_lse_raw(Fn._pad_dim(at).contiguous(), sa, Fn._pad_dim(bt).contiguous(), sb, inv_tau,
                              engine_flag=Fn._resolve_engine(unit_rows=normalize)).cpu().numpy()


@pytest.mark.parametrize("m,n,d", [(1, 1, 64), (7, 7, 64), (257, 257, 64), (64, 1000, 64), (300, 33, 64),
                                   (130, 4097, 32), (100, 777, 128), (70, 500, 256), (50, 300, 48), (40, 90, 100)])
@pytest.mark.parametrize("normalize", [True, False])
def test_row_lse_matches_oracle(Fn, m, n, d, normalize):
    rng = np.random.default_rng(m * 1000 + n + d)
    a = (rng.standard_normal((m, d)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.4).astype(np.float32)
    ref, _ = O.row_lse_scores(a, b, 5.0, normalize)
    got = _lse(Fn, a, b, 5.0, normalize)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


def test_lse_extreme_logits_and_inv_norm(Fn):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((33, 64)).astype(np.float32) * 3
    b = rng.standard_normal((2000, 64)).astype(np.float32) * 3
    b[17] = a[5] * 4             # a huge positive logit (~ +2000) in the middle of the stream
    a[9] = 0                     # an all-zero anchor: every logit 0 -> lse = log n
    ref, _ = O.row_lse_scores(a, b, 1.0, False)
    got = _lse(Fn, a, b, 1.0, False)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
    assert got[9] == pytest.approx(np.log(2000), rel=1e-6)
    inv = Fn.row_inv_norm(torch.from_numpy(a).cuda()).cpu().numpy()
    nrm = np.sqrt((a.astype(np.float64) ** 2).sum(1))
    np.testing.assert_allclose(inv, 1 / np.maximum(nrm, 1e-12), rtol=1e-6)


def test_pos_logit(Fn):
    rng = np.random.default_rng(1)
    a = rng.standard_normal((100, 64)).astype(np.float32)
    b = rng.standard_normal((50, 64)).astype(np.float32)
    pos = rng.integers(0, 50, 100)
    at, bt = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
    got = Fn.pos_logit_raw(at, sa, bt, sb, torch.from_numpy(pos).cuda(), 5.0).cpu().numpy()
    _, s = O.row_lse_scores(a, b, 5.0, True)
    np.testing.assert_allclose(got, s[np.arange(100), pos], rtol=1e-5, atol=1e-5)


def test_large_rectangular_against_c_oracle(Fn):
    """ncl.py:363-366 shape: B anchors against ALL rows of a table (2048 x 100K, d = 64)."""
    rng = np.random.default_rng(2)
    a = (rng.standard_normal((2048, 64)) * 0.3).astype(np.float32)
    b = (rng.standard_normal((100_000, 64)) * 0.3).astype(np.float32)
    ref, _ = C.row_lse(a, b, None, 10.0, True)
    got = _lse(Fn, a, b, 10.0, True)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ reference-named losses
@pytest.fixture(scope="module")
def Ls():
    from recommendation_amd import losses
    return losses


def _t(x, grad=False):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda().requires_grad_(grad)


def _gclose(t, ref, rel=2e-5, floor=1e-8):
    got = t.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=rel, atol=max(1e-5 * np.abs(ref).max(), floor))


@pytest.mark.parametrize("m", [1, 7, 257, 1000])
def test_golden_loss_values(Ls, golden, m):
    """gcl.info_nce_loss, ncl.InfoNCE (= ssl4rec.InfoNCE), ssl4rec.batch_softmax_loss outputs."""
    c = golden("contrast.npz")
    z1, z2 = _t(c[f"z1_{m}"]), _t(c[f"z2_{m}"])
    for temp in (0.1, 0.2, 0.5):
        assert float(Ls.info_nce_loss(z1, z2, temp)) == pytest.approx(float(c[f"gcl_loss_{m}_{temp}"]), rel=1e-5, abs=2e-6)
    for b_cos in (True, False):
        got = float(Ls.InfoNCE(0.3 * z1, 0.3 * z2, 0.2, b_cos))
        assert got == pytest.approx(float(c[f"ncl_infonce_{m}_{int(b_cos)}"]), rel=1e-5, abs=2e-6)
    assert float(Ls.batch_softmax_loss(z1, z2, 0.2)) == pytest.approx(float(c[f"s4r_bsl_{m}"]), rel=1e-5, abs=2e-6)


@pytest.mark.parametrize("m", [7, 257])
def test_golden_info_nce_loss_grads(Ls, golden, m):
    c = golden("contrast.npz")
    z1, z2 = _t(c[f"z1_{m}"], True), _t(c[f"z2_{m}"], True)
    Ls.info_nce_loss(z1, z2, 0.2).backward()
    _gclose(z1.grad, c[f"gcl_g1_{m}"])
    _gclose(z2.grad, c[f"gcl_g2_{m}"])


@pytest.mark.parametrize("b_cos", [1, 0])
def test_golden_ncl_infonce_grads(Ls, golden, b_cos):
    c = golden("contrast.npz")
    z1, z2 = _t(0.3 * c["z1_257"], True), _t(0.3 * c["z2_257"], True)
    Ls.InfoNCE(z1, z2, 0.2, bool(b_cos)).backward()
    # b_cos=False saturates the softmax (logits ~29, p_ii = 1 - 1e-11): the true gradient (~1e-13) is
    # below fp32 cancellation noise in the reference itself (it is (softmax - onehot) @ z / (m t)),
    # so the bound is 1e-5 of the un-cancelled term scale |z| / (m * temperature)
    floor = 1e-8 if b_cos else 1e-5 * float(np.abs(0.3 * c["z2_257"]).max()) / (257 * 0.2)
    _gclose(z1.grad, c[f"ncl_infonce_g1_257_{b_cos}"], floor=floor)
    _gclose(z2.grad, c[f"ncl_infonce_g2_257_{b_cos}"], floor=floor)


def test_golden_ncl_structure_and_prototype(Ls, golden, engine):
    """NCLModel.ssl_layer_loss / ProtoNCE_loss (ncl.py:358-375) values and gradients."""
    c = golden("contrast.npz")
    rel = 1e-5
    nu = int(c["ncl_num_users"])
    ctx, x0 = _t(c["ncl_ctx"], True), _t(c["ncl_x0"], True)
    ssl = Ls.ssl_layer_loss(ctx, x0, c["ncl_uidx"], c["ncl_iidx"], nu, float(c["ncl_ssl_temp"]),
                            float(c["ncl_ssl_reg"]), float(c["ncl_alpha"]))
    assert float(ssl) == pytest.approx(float(c["ncl_ssl"]), rel=rel)
    ssl.backward()
    _gclose(ctx.grad, c["ncl_ssl_gctx"], floor=1e-12)
    # d / d x0 at the default 2e-5 does NOT hold against this golden on any engine, the exact f32 MFMA included (2-8 of
    # 5120 elements off by up to 2.3e-4 of their value, 4e-5 of the largest element): the golden is the reference's own
    # fp32 autograd result, whose softmax-minus-one-hot cancellation carries that much noise.  So: the golden at 2.5e-4,
    # and the same expression (ncl.py:358-367) in float64 at the default.
    _gclose(x0.grad, c["ncl_ssl_gx0"], rel=2.5e-4, floor=1e-12)
    c64, x64 = _t(c["ncl_ctx"]).double().requires_grad_(True), _t(c["ncl_x0"]).double().requires_grad_(True)
    ui, ii = _t(c["ncl_uidx"]), _t(c["ncl_iidx"])
    tau = float(c["ncl_ssl_temp"])

    def side(cur, init, idx):
        nc, ni_all = F.normalize(cur[idx], dim=1), F.normalize(init, dim=1)
        pos = (nc * ni_all[idx]).sum(1) / tau
        return -(pos - torch.logsumexp(nc @ ni_all.T / tau, dim=1)).sum()

    ref64 = float(c["ncl_ssl_reg"]) * (side(c64[:nu], x64[:nu], ui) + float(c["ncl_alpha"]) * side(c64[nu:], x64[nu:], ii))
    ref64.backward()
    assert float(ssl) == pytest.approx(float(ref64), rel=rel)
    _gclose(ctx.grad, c64.grad.cpu().numpy(), floor=1e-12)
    # ... where it holds to 1e-5 of the largest element for the context rows and to 5e-5 for d / d x0: the layer-0 rows'
    # gradient goes through F.normalize's projection g - xhat <xhat, g>, and in this fixture the context rows are the
    # layer-0 rows plus 10 % noise, so g is nearly parallel to xhat and the projection cancels an order of magnitude — f32
    # rounding of g (1e-6) shows up at ~1.6e-5 of the result on every engine (three bf16 planes: 7 of 5120 elements)
    gx = x64.grad.cpu().numpy()
    _gclose(x0.grad, gx, floor=5e-5 * np.abs(gx).max())
    x0p = _t(c["ncl_x0"], True)
    proto = Ls.ProtoNCE_loss(x0p, c["ncl_uidx"], c["ncl_iidx"], nu, _t(c["ncl_ucent"]), _t(c["ncl_u2c"]),
                             _t(c["ncl_icent"]), _t(c["ncl_i2c"]), float(c["ncl_ssl_temp"]), float(c["ncl_proto_reg"]),
                             int(c["ncl_bsz"]))
    assert float(proto) == pytest.approx(float(c["ncl_proto"]), rel=rel)
    proto.backward()
    _gclose(x0p.grad, c["ncl_proto_gx0"], floor=1e-13)


@pytest.mark.parametrize("m,n,d,normalize,sym", [(64, 333, 64, True, False), (300, 300, 64, True, True),
                                                 (130, 1000, 128, False, False), (90, 90, 32, True, True),
                                                 (33, 70, 256, True, False), (50, 120, 48, True, False)])
def test_stats_grads_match_oracle(Fn, m, n, d, normalize, sym):
    """Random upstream weights on lse / pos / col: exercises every term of the backward kernels."""
    rng = np.random.default_rng(m + n + d)
    sc = 0.4 if normalize else 0.15
    a = (rng.standard_normal((m, d)) * sc).astype(np.float32)
    b = (rng.standard_normal((n, d)) * sc).astype(np.float32)
    pos = np.arange(m) if sym else rng.integers(0, n, m)
    w_l, w_p = rng.standard_normal(m), rng.standard_normal(m)
    at, bt = _t(a, True), _t(b, True)
    out = Fn.infonce_stats(at, bt, None if sym else pos, 0.2, normalize, want_col=sym)
    loss = (out[0] * _t(w_l.astype(np.float32))).sum() + (out[1] * _t(w_p.astype(np.float32))).sum()
    w_c = None
    if sym:
        w_c = rng.standard_normal(n)
        loss = loss + (out[2] * _t(w_c.astype(np.float32))).sum()
    loss.backward()
    # oracle: L = sum w_l lse + sum w_p pos (+ sum w_c col)  ==  infonce_grads with row_w = w_l and the
    # positive coefficient -w_l replaced by +w_p: add the difference analytically
    g1, g2 = O.infonce_grads(a, b, pos, 5.0, normalize, w_l, w_c)
    an, bn = (O.row_l2_normalize(a), O.row_l2_normalize(b)) if normalize else (a.astype(np.float64), b.astype(np.float64))
    extra = w_l + w_p + (w_c[:m] if sym else 0.0)       # infonce_grads subtracted w_l (+ w_c) at the positive
    d_an = extra[:, None] * bn[pos] * 5.0
    d_bn = np.zeros_like(bn)
    np.add.at(d_bn, pos, extra[:, None] * an * 5.0)
    if normalize:
        def thr(x, xn, g):
            nrm = np.maximum(np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True)), 1e-12)
            return (g - xn * (xn * g).sum(1, keepdims=True)) / nrm
        d_an, d_bn = thr(a, an, d_an), thr(b, bn, d_bn)
    _gclose(at.grad, g1 + d_an)
    _gclose(bt.grad, g2 + d_bn)


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 129, 161, 1023, 1025])
def test_every_tile_count_parity_forward_and_backward(Fn, n):
    """The software-pipelined kernels treat the first, the odd / even and the (ragged) last 32-row tile
    of a column split differently: sweep the table length across those boundaries, values and both
    gradients against the float64 oracle."""
    rng = np.random.default_rng(n)
    m, d = 45, 64
    a = (rng.standard_normal((m, d)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.4).astype(np.float32)
    pos = rng.integers(0, n, m)
    w = rng.standard_normal(m)
    at, bt = _t(a, True), _t(b, True)
    lse, pl = Fn.infonce_stats(at, bt, pos, 0.2, True)
    ref_lse, s = O.row_lse_scores(a, b, 5.0, True)
    np.testing.assert_allclose(lse.detach().cpu().numpy(), ref_lse, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pl.detach().cpu().numpy(), s[np.arange(m), pos], rtol=1e-5, atol=1e-5)
    ((lse - pl) * _t(w.astype(np.float32))).sum().backward()
    g1, g2 = O.infonce_grads(a, b, pos, 5.0, True, w)
    # n = 1: the exact gradient is 0 (softmax over one row), what is left is f32 cancellation noise of
    # terms of size |w| / tau * |xhat| ~ 1 summed over the 45 anchors
    floor = 1e-5 if n == 1 else 1e-8
    _gclose(at.grad, g1, floor=floor)
    _gclose(bt.grad, g2, floor=floor)


@pytest.mark.parametrize("m,n", [(70, 20), (70, 33), (70, 64), (70, 96), (70, 2049), (70, 40000), (600, 5000)])
def test_short_and_many_column_splits(Fn, m, n):
    """Column splits of one, two and three tiles (tables of <= 96 rows: the pipeline's prologue / tail paths)
    and the merge of many partials (40000 rows: 78 splits of 17 tiles)."""
    rng = np.random.default_rng(n)
    d = 64
    a = (rng.standard_normal((m, d)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.4).astype(np.float32)
    w = rng.standard_normal(m)
    pos = rng.integers(0, n, m)
    at, bt = _t(a, True), _t(b, True)
    lse, pl = Fn.infonce_stats(at, bt, torch.from_numpy(pos).cuda(), 0.1, True)
    ref_lse, s = O.row_lse_scores(a, b, 10.0, True)
    np.testing.assert_allclose(lse.detach().cpu().numpy(), ref_lse, rtol=1e-5, atol=1e-5)
    ((lse - pl) * _t(w.astype(np.float32))).sum().backward()
    g1, g2 = O.infonce_grads(a, b, pos, 10.0, True, w)
    _gclose(at.grad, g1)
    _gclose(bt.grad, g2)


@pytest.mark.parametrize("m,n,d,temp", [(300, 300, 64, 0.2), (1000, 257, 64, 0.05), (97, 4100, 128, 0.5), (64, 64, 32, 0.1)])
def test_one_pass_column_lse_matches_two_pass_and_oracle(Fn, m, n, d, temp):
    """gcl.py:34 `cross_entropy(sim.T)`: column LSE from the same pass (atomics) vs the swapped-role
    second pass (deterministic) vs the float64 oracle."""
    rng = np.random.default_rng(m + n)
    a = rng.standard_normal((m, d)).astype(np.float32)
    b = rng.standard_normal((n, d)).astype(np.float32)
    at, bt = Fn._pad_dim(torch.from_numpy(a).cuda()).contiguous(), Fn._pad_dim(torch.from_numpy(b).cuda()).contiguous()
    sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
    ef = Fn._resolve_engine(unit_rows=True)
    lse1, col1 = Fn.infonce_lse_raw(at, sa, bt, sb, 1 / temp, col_bound=1.0001 / temp, engine_flag=ef)
    col2 = Fn.infonce_lse_raw(bt, sb, at, sa, 1 / temp, engine_flag=ef)
    _, s = O.row_lse_scores(a, b, 1 / temp, True)
    ref_col = np.log(np.exp(s - s.max(0)).sum(0)) + s.max(0)
    np.testing.assert_allclose(col1.cpu().numpy(), ref_col, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(col2.cpu().numpy(), ref_col, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lse1.cpu().numpy(), O.row_lse_scores(a, b, 1 / temp, True)[0], rtol=1e-5, atol=1e-5)
    try:
        Fn.COL_DETERMINISTIC = True
        x, y = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        if m <= n:
            o1 = Fn.infonce_stats(x, y, None, temp, True, want_col=True)
            o2 = Fn.infonce_stats(x, y, None, temp, True, want_col=True)
            assert torch.equal(o1[2], o2[2])
    finally:
        Fn.COL_DETERMINISTIC = False


def test_info_nce_loss_round_trip_properties(Ls):
    """Size-independent checks at a size the dense formulation cannot allocate comfortably
    (100K x 100K logits = 40 GB): identical views give lse_i >= pos_i = 1/temp, the loss is
    symmetric in its arguments, and permuting both views together leaves it unchanged."""
    g = torch.Generator(device="cuda").manual_seed(3)
    z1 = torch.randn(100_000, 64, device="cuda", generator=g)
    z2 = z1 + 0.3 * torch.randn(100_000, 64, device="cuda", generator=g)
    l12 = float(Ls.info_nce_loss(z1, z2, 0.2))
    l21 = float(Ls.info_nce_loss(z2, z1, 0.2))
    assert l12 == pytest.approx(l21, rel=1e-5)
    perm = torch.randperm(100_000, device="cuda", generator=g)
    assert float(Ls.info_nce_loss(z1[perm], z2[perm], 0.2)) == pytest.approx(l12, rel=1e-5)
    from recommendation_amd import functional as F2
    lse, pos = F2.infonce_stats(z1, z1, None, 0.2)
    assert float((pos - 5.0).abs().max()) < 1e-4 and bool((lse >= pos - 1e-4).all())


@pytest.mark.parametrize("sa,sb", [(1e15, 1e-15), (1e-12, 1e12), (3e4, 3e-4), (1e-20, 1e20), (1.0, 1.0)])
def test_exponent_range_of_unnormalised_operands(Fn, sa, sb):
    """The split-operand engine claims f32's exponent range (bf16 planes): un-normalised operands whose
    magnitudes differ by up to 40 orders, products O(1) — row LSE and both gradients against float64."""
    rng = np.random.default_rng(11)
    m, n, d = 70, 900, 64
    a = (rng.standard_normal((m, d)) * 0.3 * sa).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.3 * sb).astype(np.float32)
    pos = rng.integers(0, n, m)
    w = rng.standard_normal(m)
    at, bt = _t(a, True), _t(b, True)
    lse, pl = Fn.infonce_stats(at, bt, pos, 0.5, normalize=False)
    a64, b64 = torch.from_numpy(a).double().requires_grad_(True), torch.from_numpy(b).double().requires_grad_(True)
    s = a64 @ b64.T * 2.0
    ref_lse = torch.logsumexp(s, 1)
    np.testing.assert_allclose(lse.detach().cpu().numpy(), ref_lse.detach().numpy(), rtol=1e-5, atol=1e-5)
    ((lse - pl) * _t(w.astype(np.float32))).sum().backward()
    ((ref_lse - s[torch.arange(m), torch.from_numpy(pos)]) * torch.from_numpy(w)).sum().backward()
    for got, want in ((at.grad, a64.grad.numpy()), (bt.grad, b64.grad.numpy())):
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=1e-5 * np.abs(want).max())


@pytest.mark.parametrize("m,n,d,exd", [(1, 1, 64, False), (70, 20, 64, False), (70, 33, 64, False), (129, 96, 64, False),
                                       (300, 2049, 64, False), (70, 40000, 64, False), (257, 257, 64, True),
                                       (64, 1000, 32, False), (97, 4100, 128, False), (200, 200, 128, True),
                                       (2048, 50000, 64, False)])
def test_flash_forward_lse_and_weighted_row_sum(Fn, engine, m, n, d, exd):
    """gcr_infonce_fwd_o_f32: lse and o[i] = sum_j softmax_ij bhat_j from one pass (online max with deferred
    rescale, per-split partials merged) against float64; the anchor-side gradient of a row-softmax loss is
    dL/dlse * inv_tau * o."""
    if engine == "f32":
        assert not Fn.infonce_fwd_o_supported(d, Fn.INFONCE_ENGINE_F32)
        pytest.skip("split-operand engine only; the f32 engine keeps the two-launch backward")
    ef = Fn._resolve_engine(unit_rows=True)       # rows normalised by the scales: "auto" may use the f16 planes
    rng = np.random.default_rng(m * 7 + n)
    a = (rng.standard_normal((m, d)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.4).astype(np.float32)
    if n > 300:
        b[n // 2] = 3.0 * a[0]              # a late, dominant logit: forces a rescale deep into the stream
    at, bt = _t(a), _t(b)
    sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
    inv_tau = 10.0
    lse, o = Fn.infonce_fwd_o_raw(at, sa, bt, sb, inv_tau, exclude_diagonal=exd, engine_flag=ef)
    an = a.astype(np.float64) / np.linalg.norm(a.astype(np.float64), axis=1, keepdims=True)
    bn = b.astype(np.float64) / np.linalg.norm(b.astype(np.float64), axis=1, keepdims=True)
    s = inv_tau * an @ bn.T
    if exd:
        s[np.arange(min(m, n)), np.arange(min(m, n))] = -np.inf
    mx = s.max(1, keepdims=True)
    p = np.exp(s - mx)
    ref_lse = (mx + np.log(p.sum(1, keepdims=True)))[:, 0]
    ref_o = (p / p.sum(1, keepdims=True)) @ bn
    if exd and n == 1:
        return
    np.testing.assert_allclose(lse.cpu().numpy(), ref_lse, rtol=1e-5, atol=1e-5)
    assert np.abs(o.cpu().numpy() - ref_o).max() <= 1e-5 * max(np.abs(ref_o).max(), 1e-3)
    # the same lse as the plain forward (same MFMA sequence for the scores)
    plain = Fn.infonce_lse_raw(at, sa, bt, sb, inv_tau, exclude_diagonal=exd, engine_flag=ef)
    assert float((plain - lse).abs().max()) <= 2e-6 * max(1.0, float(lse.abs().max()))


@pytest.mark.parametrize("m,n,d", [(300, 2049, 64), (64, 500, 128), (40, 40, 32)])
def test_flash_forward_path_gives_the_same_gradients(Fn, engine, monkeypatch, m, n, d):
    """Row-softmax losses take the flash-style forward when the anchors need a gradient (FWD_O): both input
    gradients equal the two-launch backward's and the float64 oracle's."""
    rng = np.random.default_rng(n)
    a = (rng.standard_normal((m, d)) * 0.4).astype(np.float32)
    b = (rng.standard_normal((n, d)) * 0.4).astype(np.float32)
    w = rng.standard_normal(m).astype(np.float32)
    pos = rng.integers(0, n, m)
    grads = {}
    for flag in (True, False):
        monkeypatch.setattr(Fn, "FWD_O", flag)
        at, bt = _t(a, True), _t(b, True)
        lse, pl = Fn.infonce_stats(at, bt, torch.from_numpy(pos).cuda(), 0.1, True)
        ((lse - pl) * _t(w)).sum().backward()
        grads[flag] = (at.grad.clone(), bt.grad.clone())
    g1, g2 = O.infonce_grads(a, b, pos, 10.0, True, w.astype(np.float64))
    for flag in (True, False):
        _gclose(grads[flag][0], g1)
        _gclose(grads[flag][1], g2)
    # table side: the same launch, fed an lse that differs in the last bits (online merge vs two-stage merge)
    assert float((grads[True][1] - grads[False][1]).abs().max()) <= 1e-5 * float(grads[False][1].abs().max())


def test_two_product_loop_random_shapes(Fn, engine):
    """The cross-tile pipelined loop (prologue of two tiles, steps in pairs, a peeled last step, a ring of three LDS
    images) over 48 random (M, N, d, 1/tau) incl. tile counts 1..6 per split and ragged last tiles: flash forward
    (lse, o) and the table-side / anchor-side backward against float64.  d = 128 runs the same loop on two f16 planes
    (three bf16 planes keep their own un-pipelined kernels there)."""
    if engine == "f32":
        pytest.skip("the pipelined loop is the split-operand engine's")
    ef = Fn._resolve_engine(unit_rows=True)
    rng = np.random.default_rng(2024)
    for case in range(48):
        d = int(rng.choice([32, 64, 128]))
        m = int(rng.integers(1, 400))
        n = int(rng.choice([rng.integers(1, 200), rng.integers(1, 3000), 32 * rng.integers(1, 7)]))
        inv_tau = float(rng.choice([1.0, 5.0, 10.0, 20.0]))
        a = (rng.standard_normal((m, d)) * rng.uniform(0.1, 2.0)).astype(np.float32)
        b = (rng.standard_normal((n, d)) * rng.uniform(0.1, 2.0)).astype(np.float32)
        at, bt = _t(a), _t(b)
        sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
        lse, o = Fn.infonce_fwd_o_raw(at, sa, bt, sb, inv_tau, engine_flag=ef)
        an = a.astype(np.float64) / np.linalg.norm(a.astype(np.float64), axis=1, keepdims=True)
        bn = b.astype(np.float64) / np.linalg.norm(b.astype(np.float64), axis=1, keepdims=True)
        s = inv_tau * an @ bn.T
        mx = s.max(1, keepdims=True)
        p = np.exp(s - mx)
        ref_lse = (mx + np.log(p.sum(1, keepdims=True)))[:, 0]
        sm = p / p.sum(1, keepdims=True)
        tag = (case, m, n, d, inv_tau)
        np.testing.assert_allclose(lse.cpu().numpy(), ref_lse, rtol=1e-5, atol=1e-5, err_msg=str(tag))
        ref_o = sm @ bn
        assert np.abs(o.cpu().numpy() - ref_o).max() <= 1e-5 * max(np.abs(ref_o).max(), 1e-3), tag
        # backward, both roles of the loop: rows of b stationary with the anchors' statistics on the streamed side,
        # and rows of a stationary with their own statistics
        w = rng.standard_normal(m).astype(np.float32)
        wt = _t(w)
        gb = Fn._infonce_bwd_raw(bt, sb, at, sa, inv_tau, None, None, lse, wt, engine_flag=ef).cpu().numpy()
        ref_gb = inv_tau * (sm * w[:, None].astype(np.float64)).T @ an
        assert np.abs(gb - ref_gb).max() <= 2e-5 * max(np.abs(ref_gb).max(), 1e-6), tag
        ga = Fn._infonce_bwd_raw(at, sa, bt, sb, inv_tau, lse, wt, None, None, engine_flag=ef).cpu().numpy()
        ref_ga = inv_tau * (sm * w[:, None].astype(np.float64)) @ bn
        assert np.abs(ga - ref_ga).max() <= 2e-5 * max(np.abs(ref_ga).max(), 1e-6), tag


@pytest.mark.parametrize("m,n,d,inv_tau", [(1280, 2, 32, 10.0), (300, 33, 64, 20.0), (130, 70, 128, 20.0)])
def test_backward_ragged_tile_when_every_real_logit_is_far_below_zero(Fn, engine, m, n, d, inv_tau):
    """The rows behind the end of a ragged streamed tile are staged as zero rows: score 0.  When every REAL logit of an
    anchor sits near -1/tau, e^{0 - lse_i} of such a row is ~e^{1/tau} — beyond the f16 planes of the two-plane format once
    the weight scale is in (inf times the zero row = NaN) unless the end-of-array mask removes it first.  Statistics on the
    stationary rows (weights of both signs), on both sides, and on the streamed rows, against float64.
    (Found by scripts/stress_infonce_formats.py, case (1280, 2, 32, 10).)"""
    ef = Fn._resolve_engine(unit_rows=True)
    rng = np.random.default_rng(m * 7 + n)
    a = rng.standard_normal((m, d)).astype(np.float32)
    a += 4.0 * rng.standard_normal((1, d)).astype(np.float32)             # every anchor close to one direction ...
    b = (-a[:n] + 0.05 * rng.standard_normal((n, d))).astype(np.float32)  # ... and every streamed row opposite to it
    at, bt = _t(a), _t(b)
    sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
    an = a.astype(np.float64) / np.linalg.norm(a.astype(np.float64), axis=1, keepdims=True)
    bn = b.astype(np.float64) / np.linalg.norm(b.astype(np.float64), axis=1, keepdims=True)
    sc = inv_tau * an @ bn.T
    assert sc.max() < -0.5 * inv_tau
    lse = np.log(np.exp(sc).sum(1))
    col = np.log(np.exp(sc).sum(0))
    w = rng.standard_normal(m).astype(np.float32)
    v = rng.standard_normal(n).astype(np.float32)
    p_row = np.exp(sc - lse[:, None]) * w[:, None].astype(np.float64)
    p_col = np.exp(sc - col[None, :]) * v[None, :].astype(np.float64)
    lt, ct = _t(lse.astype(np.float32)), _t(col.astype(np.float32))
    for args, ref in (((lt, _t(w), None, None), inv_tau * p_row @ bn),
                      ((lt, _t(w), ct, _t(v)), inv_tau * (p_row + p_col) @ bn),
                      ((None, None, ct, _t(v)), inv_tau * p_col @ bn)):
        g = Fn._infonce_bwd_raw(at, sa, bt, sb, inv_tau, *args, engine_flag=ef).cpu().numpy()
        assert np.isfinite(g).all()
        assert np.abs(g - ref).max() <= 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("m,n,d", [(70, 500, 64), (2048, 3000, 64), (333, 1000, 32), (333, 1000, 128)])
def test_table_side_backward_weights_in_the_exponent(Fn, engine, m, n, d):
    """Statistics on the streamed rows only: the two-f16-plane loop carries each weight as 2^(log2|w| - lse log2 e) inside
    the exponent and its sign in the staged row (h2_fold_kernel).  Weights of both signs, exact zeros, 25 orders of
    magnitude and a ragged last tile against float64; the rows whose weight is zero must contribute exactly nothing."""
    ef = Fn._resolve_engine(unit_rows=True)
    rng = np.random.default_rng(m + n + d)
    a = rng.standard_normal((m, d)).astype(np.float32)
    b = rng.standard_normal((n, d)).astype(np.float32)
    at, bt = _t(a), _t(b)
    sa, sb = Fn.row_inv_norm(at), Fn.row_inv_norm(bt)
    inv_tau = 10.0
    lse = Fn.infonce_lse_raw(at, sa, bt, sb, inv_tau, engine_flag=ef)
    w = (rng.standard_normal(m) * 10.0 ** rng.uniform(-20, 0, m)).astype(np.float32)
    w[rng.random(m) < 0.2] = 0.0
    w[0] = -3.0
    an = a.astype(np.float64) / np.linalg.norm(a.astype(np.float64), axis=1, keepdims=True)
    bn = b.astype(np.float64) / np.linalg.norm(b.astype(np.float64), axis=1, keepdims=True)
    sm = np.exp(inv_tau * an @ bn.T - lse.cpu().numpy().astype(np.float64)[:, None])
    gb = Fn._infonce_bwd_raw(bt, sb, at, sa, inv_tau, None, None, lse, _t(w), engine_flag=ef).cpu().numpy()
    ref = inv_tau * (sm * w[:, None].astype(np.float64)).T @ an
    assert np.isfinite(gb).all()
    assert np.abs(gb - ref).max() <= 1e-5 * np.abs(ref).max()
    # only the zero-weight anchors differ: same result with their rows replaced by garbage of unit norm
    a2 = a.copy()
    a2[w == 0] = rng.standard_normal((int((w == 0).sum()), d)).astype(np.float32)
    a2t = _t(a2)
    gb2 = Fn._infonce_bwd_raw(bt, sb, a2t, Fn.row_inv_norm(a2t), inv_tau, None, None, lse, _t(w), engine_flag=ef).cpu().numpy()
    np.testing.assert_array_equal(gb, gb2)
    # all weights zero: an exactly zero gradient (the scale pre-pass sees max |w| = 0)
    gz = Fn._infonce_bwd_raw(bt, sb, at, sa, inv_tau, None, None, lse, _t(np.zeros(m, np.float32)), engine_flag=ef)
    assert float(gz.abs().max()) == 0.0


@pytest.mark.parametrize("m,d", [(33, 64), (65, 64), (129, 64), (257, 64), (300, 32), (1000, 64), (129, 128), (700, 128)])
def test_exclude_diagonal_backward_raw_multi_tile(Fn, engine, m, d):
    """gcr_infonce_bwd_ex_f32 with GCR_INFONCE_EXCLUDE_DIAGONAL on a self-similarity problem of 2..32 tiles (the pipelined
    loop's MODE 0 with the diagonal mask in every step), statistics on the streamed side and on the stationary side,
    against float64."""
    ef = Fn._resolve_engine(unit_rows=True)
    rng = np.random.default_rng(m + d)
    x = (rng.standard_normal((m, d)) * 0.5).astype(np.float32)
    w = rng.standard_normal(m).astype(np.float32)
    xt = _t(x)
    s = Fn.row_inv_norm(xt)
    inv_tau = 4.0
    xn = x.astype(np.float64) / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)
    sc = inv_tau * xn @ xn.T
    np.fill_diagonal(sc, -np.inf)
    lse = np.log(np.exp(sc).sum(1))
    p = np.exp(sc - lse[:, None]) * w[:, None].astype(np.float64)
    lt, wt = _t(lse.astype(np.float32)), _t(w)
    gy = Fn._infonce_bwd_raw(xt, s, xt, s, inv_tau, None, None, lt, wt, exclude_diagonal=True, engine_flag=ef).cpu().numpy()
    gx = Fn._infonce_bwd_raw(xt, s, xt, s, inv_tau, lt, wt, None, None, exclude_diagonal=True, engine_flag=ef).cpu().numpy()
    ref_y, ref_x = inv_tau * p.T @ xn, inv_tau * p @ xn
    assert np.abs(gy - ref_y).max() <= 1e-5 * np.abs(ref_y).max()
    assert np.abs(gx - ref_x).max() <= 1e-5 * np.abs(ref_x).max()


def test_loop_instantiations_random_square(Fn, engine):
    """Every remaining instantiation of the two-product loop on 40 random square self-similarity problems (1..13 tiles,
    ragged last tiles, d in {32, 64}): flash forward and backward with the excluded diagonal, and the backward with
    statistics on BOTH sides (the symmetric loss's P = w_i e^{s - lse_i} + v_j e^{s - col_j}), against float64."""
    if engine == "f32":
        pytest.skip("the pipelined loop is the split-operand engine's")
    ef = Fn._resolve_engine(unit_rows=True)
    rng = np.random.default_rng(77)
    for case in range(40):
        d = int(rng.choice([32, 64]))
        m = int(rng.choice([rng.integers(2, 100), rng.integers(2, 420), 32 * rng.integers(1, 9) + 1]))
        inv_tau = float(rng.choice([2.0, 5.0, 10.0]))
        x = (rng.standard_normal((m, d)) * rng.uniform(0.2, 1.5)).astype(np.float32)
        y = (x + 0.3 * rng.standard_normal((m, d))).astype(np.float32)
        w, v = rng.standard_normal(m).astype(np.float32), rng.standard_normal(m).astype(np.float32)
        xt, yt = _t(x), _t(y)
        sx, sy = Fn.row_inv_norm(xt), Fn.row_inv_norm(yt)
        xn = x.astype(np.float64) / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)
        yn = y.astype(np.float64) / np.linalg.norm(y.astype(np.float64), axis=1, keepdims=True)
        tag = (case, m, d, inv_tau)
        # excluded diagonal: flash forward + both backward roles on the self-similarity of x
        s = inv_tau * xn @ xn.T
        np.fill_diagonal(s, -np.inf)
        lse = np.log(np.exp(s - s.max(1, keepdims=True)).sum(1)) + s.max(1)
        sm = np.exp(s - lse[:, None])
        got_lse, got_o = Fn.infonce_fwd_o_raw(xt, sx, xt, sx, inv_tau, exclude_diagonal=True, engine_flag=ef)
        np.testing.assert_allclose(got_lse.cpu().numpy(), lse, rtol=1e-5, atol=1e-5, err_msg=str(tag))
        ref_o = sm @ xn
        assert np.abs(got_o.cpu().numpy() - ref_o).max() <= 1e-5 * max(np.abs(ref_o).max(), 1e-3), tag
        lt, wt = _t(lse.astype(np.float32)), _t(w)
        for stats_on_streamed in (True, False):
            args = (None, None, lt, wt) if stats_on_streamed else (lt, wt, None, None)
            g = Fn._infonce_bwd_raw(xt, sx, xt, sx, inv_tau, *args, exclude_diagonal=True, engine_flag=ef).cpu().numpy()
            pw = sm * w[:, None].astype(np.float64)
            ref = inv_tau * (pw.T if stats_on_streamed else pw) @ xn
            assert np.abs(g - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-6), (tag, stats_on_streamed)
        # statistics on both sides (x rows vs y rows): row lse of x, column lse over x for each y
        s2 = inv_tau * xn @ yn.T
        row = np.log(np.exp(s2 - s2.max(1, keepdims=True)).sum(1)) + s2.max(1)
        col = np.log(np.exp(s2 - s2.max(0, keepdims=True)).sum(0)) + s2.max(0)
        p2 = np.exp(s2 - row[:, None]) * w[:, None].astype(np.float64) + np.exp(s2 - col[None, :]) * v[None, :].astype(np.float64)
        gx = Fn._infonce_bwd_raw(xt, sx, yt, sy, inv_tau, _t(row.astype(np.float32)), wt, _t(col.astype(np.float32)), _t(v),
                                 engine_flag=ef).cpu().numpy()
        ref_gx = inv_tau * p2 @ yn
        assert np.abs(gx - ref_gx).max() <= 2e-5 * max(np.abs(ref_gx).max(), 1e-6), tag


def test_two_operand_formats_agree_on_random_problems(engine):
    """scripts/stress_infonce_formats.py: 320 random problems (~10 s) x every launch kind (forward +- column sums +-
    excluded diagonal, flash forward, backward with statistics on either or both sides), two f16 planes against three
    bf16 planes: <= 1e-5 relative (max norm) everywhere."""
    if engine != "auto":
        pytest.skip("runs both formats itself")
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "stress_infonce_formats.py")
    spec = importlib.util.spec_from_file_location("stress_infonce_formats", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    worst = mod.run(320, 20.0, verbose=False)
    assert max(worst.values()) <= 1e-5, worst


def test_eight_wave_form_of_the_loop_matches_three_plane_format(Fn, engine):
    """Problems whose splits are >= 256 tiles long run the flash forward as 512-thread workgroups (two wave groups one
    barrier interval apart, sharing each staged tile): lse and o at 2048 x 300K and 50K x 50K (excluded diagonal) against
    the three-bf16-plane format, which never takes that form; the backward launches of the same problems (statistics on the
    stationary rows, on both sides, excluded diagonal; 256-thread form over the pre-scaled image of the streamed side)
    ride along."""
    if engine != "auto":
        pytest.skip("compares the two formats itself")
    g = torch.Generator(device="cuda").manual_seed(9)
    h2, b3 = Fn.INFONCE_UNIT_ROWS, 0

    def close(x, y, tol=1e-5):
        assert float((x - y).abs().max()) <= tol * float(y.abs().max())

    m, n, d, inv_tau = 2000, 300_000, 64, 10.0
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    b[12345] = 3.0 * a[7]
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    w = torch.randn(m, device="cuda", generator=g)
    v = torch.randn(n, device="cuda", generator=g)
    res = {}
    for eng in (h2, b3):
        lse, o = Fn.infonce_fwd_o_raw(a, sa, b, sb, inv_tau, engine_flag=eng)
        col = Fn.infonce_lse_raw(b, sb, a, sa, inv_tau, engine_flag=eng)
        res[eng] = (lse, o, Fn._infonce_bwd_raw(a, sa, b, sb, inv_tau, lse, w, None, None, engine_flag=eng),
                    Fn._infonce_bwd_raw(a, sa, b, sb, inv_tau, lse, w, col, v, engine_flag=eng))
    for x, y in zip(res[h2], res[b3]):
        close(x, y)
    # the lse-only forward takes its own 512-thread form (512 anchors per workgroup) from 2048 x 524K on: ragged last tile,
    # anchors that do not fill the last workgroup
    m2, n2 = 2048 + 300, 600_011
    a2 = torch.randn(m2, d, device="cuda", generator=g)
    b2 = torch.randn(n2, d, device="cuda", generator=g)
    b2[77] = 2.0 * a2[2100]
    sa2, sb2 = Fn.row_inv_norm(a2), Fn.row_inv_norm(b2)
    close(Fn.infonce_lse_raw(a2, sa2, b2, sb2, inv_tau, engine_flag=h2), Fn.infonce_lse_raw(a2, sa2, b2, sb2, inv_tau, engine_flag=b3),
          tol=2e-6)
    k = 50_000
    x = torch.randn(k, d, device="cuda", generator=g)
    sx = Fn.row_inv_norm(x)
    wk = torch.randn(k, device="cuda", generator=g)
    res = {}
    for eng in (h2, b3):
        lse, o = Fn.infonce_fwd_o_raw(x, sx, x, sx, 5.0, exclude_diagonal=True, engine_flag=eng)
        res[eng] = (lse, o, Fn._infonce_bwd_raw(x, sx, x, sx, 5.0, lse, wk, None, None, exclude_diagonal=True, engine_flag=eng),
                    Fn._infonce_bwd_raw(x, sx, x, sx, 5.0, None, None, lse, wk, exclude_diagonal=True, engine_flag=eng))
    for p, q in zip(res[h2], res[b3]):
        close(p, q)
