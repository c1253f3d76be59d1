"""GPU parity of the fused BPR loss (values + gradients vs the reference-generated goldens and the
oracle), and bit-exact parity of the Philox negative sampler / edge-mask kernels vs the CPU
restatements (integer work: bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import oracle_c as C
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Fn():
    from recommendation_amd import functional
    return functional


def _load(golden):
    b = golden("bpr.npz")
    ut = torch.from_numpy(b["user_tab"]).cuda().requires_grad_(True)
    it = torch.from_numpy(b["item_tab"]).cuda().requires_grad_(True)
    return b, ut, it


def _t(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda().requires_grad_(grad)


def _close(t, ref, rel=2e-5):
    got = t.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=rel, atol=rel * max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("variant,name", [(0, "ncl_bpr"), (1, "sept_bpr")])
def test_bpr_golden_value_and_grads(Fn, golden, variant, name):
    """ncl.bpr_loss (ncl.py:116-120) and sept.bpr_loss (sept.py:34-38) + autograd gradients."""
    b, ut, it = _load(golden)
    sums = Fn.bpr_sums(ut, it, b["u_idx"], b["i_idx"], b["j_idx"], variant)
    loss = sums[0] / len(b["u_idx"])
    assert float(loss.detach()) == pytest.approx(float(b[f"{name}_loss"]), rel=1e-5)
    assert float(sums[4].detach()) == 0.0
    loss.backward()
    _close(ut.grad, b[f"{name}_gu"])
    _close(it.grad, b[f"{name}_gi"])


@pytest.mark.parametrize("jkey,name", [("j_idx", "lgcn_block_n1"), ("j_idx3", "lgcn_block_n3")])
def test_lightgcn_loss_block(Fn, golden, jkey, name):
    """lightgcn.py:95-118: -log(sigmoid) BPR (mean of n_neg negatives) + reg * (|u|^2 + |p|^2)."""
    b, ut, it = _load(golden)
    sums = Fn.bpr_sums(ut, it, b["u_idx"], b["i_idx"], b[jkey], Fn.BPR_LOG_SIGMOID)
    loss = sums[0] / len(b["u_idx"]) + 1e-4 * (sums[1] + sums[2])
    assert float(loss.detach()) == pytest.approx(float(b[f"{name}_loss"]), rel=1e-5)
    loss.backward()
    _close(ut.grad, b[f"{name}_gu"])
    _close(it.grad, b[f"{name}_gi"])


def test_gcl_loss_block_and_ncl_l2reg(Fn, golden):
    """gcl.py:216-223 bpr + reg/len(users); ncl.py:122-123 l2_reg_loss from the same sums."""
    b, ut, it = _load(golden)
    n = len(b["u_idx"])
    sums = Fn.bpr_sums(ut, it, b["u_idx"], b["i_idx"], b["j_idx"], Fn.BPR_LOGSIGMOID)
    loss = sums[0] / n + 1e-4 * (sums[1] + sums[2] + sums[3]) / n
    assert float(loss.detach()) == pytest.approx(float(b["gcl_block_loss"]), rel=1e-5)
    loss.backward()
    _close(ut.grad, b["gcl_block_gu"])
    _close(it.grad, b["gcl_block_gi"])
    ut.grad = it.grad = None
    sums = Fn.bpr_sums(ut, it, b["u_idx"], b["i_idx"], b["j_idx"], Fn.BPR_NCL)
    l2 = 1e-4 * (sums[1].sqrt() + sums[2].sqrt() + sums[3].sqrt()) / n
    assert float(l2.detach()) == pytest.approx(float(b["ncl_l2reg_loss"]), rel=1e-5)
    l2.backward()
    _close(ut.grad, b["ncl_l2reg_gu"])
    _close(it.grad, b["ncl_l2reg_gi"])


@pytest.mark.parametrize("d", [64, 32, 128, 50])
def test_bpr_oracle_dims_and_bad_ids(Fn, d):
    rng = np.random.default_rng(d)
    ut = (rng.standard_normal((500, d)) * 0.3).astype(np.float32)
    it = (rng.standard_normal((300, d)) * 0.3).astype(np.float32)
    u, i, j = rng.integers(0, 500, 4099), rng.integers(0, 300, 4099), rng.integers(0, 300, (4099, 2))
    for var in (0, 1, 2):
        sums = Fn.bpr_sums(torch.from_numpy(ut).cuda(), torch.from_numpy(it).cuda(), u, i, j, var)
        assert float(sums[0]) / 4099 == pytest.approx(O.bpr_loss(ut, it, u, i, j, var), rel=1e-5)
        assert float(sums[1]) == pytest.approx(O.sq_norm_reg(ut[u]), rel=1e-5)
        assert float(sums[3]) == pytest.approx(O.sq_norm_reg(it[j.reshape(-1)]), rel=1e-5)
    a = torch.from_numpy(ut).cuda().requires_grad_(True)
    bt = torch.from_numpy(it).cuda().requires_grad_(True)
    (Fn.bpr_sums(a, bt, u, i, j, 1)[0] / 4099).backward()
    gu, gi = O.bpr_grads(ut, it, u, i, j, 1)
    _close(a.grad, gu, 5e-5)
    _close(bt.grad, gi, 5e-5)
    # out-of-range ids are skipped and counted, never dereferenced
    u_bad = u.copy()
    u_bad[[3, 77]] = [500, -1]
    sums = Fn.bpr_sums(torch.from_numpy(ut).cuda(), torch.from_numpy(it).cuda(), u_bad, i, j, 0)
    assert float(sums[4]) == 2.0
    keep = np.ones(4099, bool)
    keep[[3, 77]] = False
    assert float(sums[0]) == pytest.approx(O.bpr_loss(ut, it, u[keep], i[keep], j[keep], 0) * 4097, rel=1e-5)


def test_bpr_full_batch_deterministic(Fn):
    """lightgcn.py:86-108 trains full-batch (B = E): 2M triples; forward is bitwise reproducible."""
    g = torch.Generator(device="cuda").manual_seed(1)
    ut = torch.randn(100_000, 64, device="cuda", generator=g) * 0.2
    it = torch.randn(20_000, 64, device="cuda", generator=g) * 0.2
    u = torch.randint(0, 100_000, (2_000_000,), device="cuda", generator=g)
    i = torch.randint(0, 20_000, (2_000_000,), device="cuda", generator=g)
    j = torch.randint(0, 20_000, (2_000_000,), device="cuda", generator=g)
    s1 = Fn.bpr_sums(ut, it, u, i, j, 2)
    s2 = Fn.bpr_sums(ut, it, u, i, j, 2)
    assert torch.equal(s1, s2)
    x = (ut[u] * it[i]).sum(1) - (ut[u] * it[j]).sum(1)
    ref = torch.nn.functional.softplus(-x.double()).sum()
    assert float(s1[0]) == pytest.approx(float(ref), rel=1e-5)


def _user_csr(u, i, n_users):
    order = np.lexsort((i, u))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=n_users))]).astype(np.int64)
    return rowptr, i[order].astype(np.int32)


@pytest.mark.parametrize("n_negs,trials", [(1, 101), (3, 101), (1, 0), (2, 2)])
def test_neg_sampler_bit_exact(Fn, n_negs, trials):
    u, i = O.synthetic_interactions(400, 60, 9000, seed=8)     # dense rows -> plenty of rejections
    rowptr, items = _user_csr(u, i, 400)
    ub = u[:3000]
    seed, offset = 0xDEADBEEF12345, 2 ** 40 + 3
    got = Fn.neg_sample(torch.from_numpy(rowptr).cuda(), torch.from_numpy(items).cuda(), torch.from_numpy(ub).cuda(),
                        n_negs, 60, seed, offset, trials).cpu().numpy()
    ref = C.neg_sample(rowptr, items, ub, n_negs, 60, seed, offset, trials)
    assert np.array_equal(got, ref)
    ref_np = O.neg_sample_uniform(rowptr, items, ub[:200], n_negs, 60, seed, offset, trials)
    assert np.array_equal(got[: 200 * n_negs], ref_np)
    if trials >= 100:
        pos = set(zip(u.tolist(), i.tolist()))
        assert got.min() >= 0 and got.max() < 60
        assert all((a, b) not in pos for a, b in zip(np.repeat(ub, n_negs).tolist(), got.tolist()))
    if trials == 2:
        assert (got == -1).any()      # exhausted slots are reported, not silently accepted


def test_neg_sampler_uniformity_and_large(Fn):
    n_users, n_items = 50_000, 10_000
    u, i = O.synthetic_interactions(n_users, n_items, 500_000, seed=9)
    rowptr, items = _user_csr(u, i, n_users)
    ub = torch.from_numpy(u).cuda()
    got = Fn.neg_sample(torch.from_numpy(rowptr).cuda(), torch.from_numpy(items).cuda(), ub, 2, n_items, 5, 0, 101)
    assert got.numel() == 1_000_000 and int(got.min()) >= 0 and int(got.max()) < n_items
    keys = torch.from_numpy(u * n_items + i).cuda()
    assert not bool(torch.isin(ub.repeat_interleave(2) * n_items + got, keys).any())
    counts = torch.bincount(got, minlength=n_items).double()
    assert abs(float(counts.mean()) - 100.0) < 1e-9 and float(counts.std()) < 14.0   # ~Poisson(100)


@pytest.mark.parametrize("nnz", [1, 31, 32, 33, 100_003])
def test_edge_mask_bit_exact(Fn, nnz):
    bits = Fn.edge_mask_bits(nnz, 0.3, 1234567, "cuda").cpu().numpy().view(np.uint8)
    got = np.unpackbits(bits, bitorder="little")
    ref = O.edge_keep_mask(nnz, 0.3, 1234567)
    assert np.array_equal(got[:nnz].astype(bool), ref)
    assert not got[nnz:].any()
    if nnz > 1000:
        assert np.array_equal(C.edge_keep_mask(nnz, 0.3, 1234567), ref)
        perm = np.random.default_rng(0).permutation(nnz)
        pb = Fn.edge_mask_bits(nnz, 0.3, 1234567, "cuda", edge_id=torch.from_numpy(perm).cuda()).cpu().numpy()
        assert np.array_equal(np.unpackbits(pb.view(np.uint8), bitorder="little")[:nnz].astype(bool), ref[perm])
        assert abs(ref.mean() - 0.7) < 0.01


@pytest.mark.parametrize("jkey,name,variant", [("j_idx", "ncl_bpr", 0), ("j_idx", "lgcn_block_n1", 2), ("j_idx3", "lgcn_block_n3", 2)])
def test_sorted_backward_matches_goldens(Fn, golden, monkeypatch, jkey, name, variant):
    """The large-batch backward (three sorted orders, one row atomic per run) on the reference-generated
    gradients: forced on at golden size through the batch threshold."""
    monkeypatch.setattr(Fn, "BPR_SORTED_MIN_BATCH", 1)
    b, ut, it = _load(golden)
    sums = Fn.bpr_sums(ut, it, b["u_idx"], b["i_idx"], b[jkey], variant)
    loss = sums[0] / len(b["u_idx"]) + (1e-4 * (sums[1] + sums[2]) if variant == 2 else 0.0)
    assert float(loss.detach()) == pytest.approx(float(b[f"{name}_loss"]), rel=1e-5)
    loss.backward()
    _close(ut.grad, b[f"{name}_gu"])
    _close(it.grad, b[f"{name}_gi"])


@pytest.mark.parametrize("d,n_neg", [(64, 1), (64, 2), (100, 1), (200, 3)])
def test_sorted_backward_equals_atomic_backward(Fn, monkeypatch, d, n_neg):
    """Random large-ish batch with heavy key repetition, bad ids in every index vector and all four
    upstream gradients: the sorted and the atomic backward must agree to summation-order noise."""
    rng = np.random.default_rng(d + n_neg)
    n_u, n_i, bsz = 3000, 700, 50_000
    ut0 = rng.standard_normal((n_u, d)).astype(np.float32) * 0.3
    it0 = rng.standard_normal((n_i, d)).astype(np.float32) * 0.3
    u = rng.integers(0, n_u, bsz)
    i = (rng.pareto(1.2, bsz) * 5).astype(np.int64) % n_i            # popular items: long runs
    j = rng.integers(0, n_i, (bsz, n_neg)) if n_neg > 1 else rng.integers(0, n_i, bsz)
    u[17], i[99] = -1, n_i + 5
    if n_neg > 1:
        j[123, 1] = n_i
    else:
        j[123] = -7
    w = torch.tensor([1.0 / bsz, 3e-4, 2e-4, 1e-4, 0.0], device="cuda")
    grads = {}
    # sorted: the negatives' item rows from the sort that carries (user, coefficient) with the key; sorted_index: from the
    # index sort (gcr_bpr_bwd_sorted_f32's third launch)
    for mode, thr, payload in (("atomic", 1 << 40, True), ("sorted", 1, True), ("sorted_index", 1, False)):
        monkeypatch.setattr(Fn, "BPR_SORTED_MIN_BATCH", thr)
        monkeypatch.setattr(Fn, "BPR_NEG_PAYLOAD_SORT", payload)
        ut = torch.from_numpy(ut0).cuda().requires_grad_(True)
        it = torch.from_numpy(it0).cuda().requires_grad_(True)
        sums = Fn.bpr_sums(ut, it, u, i, j, Fn.BPR_LOGSIGMOID)
        assert float(sums[4]) == 3.0
        (sums * w).sum().backward()
        grads[mode] = (ut.grad.cpu().numpy(), it.grad.cpu().numpy())
    for other in ("sorted", "sorted_index"):
        for a, s in zip(grads["atomic"], grads[other]):
            np.testing.assert_allclose(s, a, rtol=2e-4, atol=2e-6 * np.abs(a).max())
    monkeypatch.setattr(Fn, "BPR_NEG_PAYLOAD_SORT", True)
    # and against the float64 oracle on the valid samples
    ok = np.ones(bsz, bool)
    ok[[17, 99, 123]] = False
    jo = j[ok] if n_neg == 1 else j[ok]
    gu, gi = O.bpr_grads(ut0, it0, u[ok], i[ok], jo, variant=1) if n_neg == 1 else (None, None)
    if gu is not None:
        # oracle gradient of sum_b loss_b / |valid|: rescale to the weights used above (loss term only)
        ut = torch.from_numpy(ut0).cuda().requires_grad_(True)
        it = torch.from_numpy(it0).cuda().requires_grad_(True)
        monkeypatch.setattr(Fn, "BPR_SORTED_MIN_BATCH", 1)
        (Fn.bpr_sums(ut, it, u, i, j, Fn.BPR_LOGSIGMOID)[0] / ok.sum()).backward()
        _close(ut.grad, gu, rel=5e-5)
        _close(it.grad, gi, rel=5e-5)


def test_fused_ncl_rec_loss_equals_gathered_form(Fn, golden):
    """NCLModel.train_step's fused form — bpr_sums on the tables, l2 from the squared-norm outputs — equals the
    reference's expression on gathered rows (ncl.py:314-317,326: bpr_loss(u, p, n) + l2_reg_loss(reg, u, p, n) / B),
    value (golden ncl_bpr / ncl_l2reg) and gradients."""
    from recommendation_amd import losses as Ls
    b = golden("bpr.npz")
    ui, pi, ni = (torch.from_numpy(b[k]).cuda() for k in ("u_idx", "i_idx", "j_idx"))
    reg, n_b = 1e-4, ui.numel()
    ut, it = _t(b["user_tab"], True), _t(b["item_tab"], True)
    s = Fn.bpr_sums(ut, it, ui, pi, ni, Fn.BPR_NCL)
    l2 = reg * (s[1].sqrt() + s[2].sqrt() + s[3].sqrt()) / n_b
    fused = s[0] / n_b + l2
    assert float(s[0] / n_b) == pytest.approx(float(b["ncl_bpr_loss"]), rel=1e-5)
    assert float(l2) == pytest.approx(float(b["ncl_l2reg_loss"]), rel=1e-5)
    fused.backward()
    ut2, it2 = _t(b["user_tab"], True), _t(b["item_tab"], True)
    gathered = Ls.bpr_loss(ut2[ui], it2[pi], it2[ni]) + Ls.l2_reg_loss(reg, ut2[ui], it2[pi], it2[ni])
    gathered.backward()
    assert float(fused) == pytest.approx(float(gathered), rel=1e-6)
    for a, c in ((ut.grad, ut2.grad), (it.grad, it2.grad)):
        assert float((a - c).abs().max()) <= 2e-6 * float(c.abs().max())
    np.testing.assert_allclose(ut.grad.cpu().numpy(), b["ncl_bpr_gu"] + b["ncl_l2reg_gu"], rtol=2e-5, atol=2e-6 * np.abs(b["ncl_bpr_gu"]).max())


@pytest.mark.gpu
def test_bpr_on_split_table_writes_one_gradient_buffer():
    """bpr_sums on the two halves of `split_rows(stacked)` (ncl.py:314-317 after :422-423): the backward fills ONE [N, d]
    buffer that the split hands back without a copy; same gradient as separate leaf tables."""
    from recommendation_amd import functional as Fn
    g = torch.Generator(device="cuda").manual_seed(5)
    n_u, n_i, d, b = 700, 300, 64, 512
    stacked = torch.randn(n_u + n_i, d, device="cuda", generator=g)
    u = torch.randint(0, n_u, (b,), device="cuda", generator=g)
    i = torch.randint(0, n_i, (b,), device="cuda", generator=g)
    j = torch.randint(0, n_i, (b,), device="cuda", generator=g)
    s1 = stacked.clone().requires_grad_(True)
    ue, ie = Fn.split_rows(s1 * 1.0, n_u)                       # a non-leaf stacked table, as the propagation output is
    sums = Fn.bpr_sums(ue, ie, u, i, j, Fn.BPR_NCL)
    (sums[0] + 0.1 * (sums[1] + sums[2] + sums[3])).backward()
    ut = stacked[:n_u].clone().requires_grad_(True)
    it = stacked[n_u:].clone().requires_grad_(True)
    ref = Fn.bpr_sums(ut, it, u, i, j, Fn.BPR_NCL)
    (ref[0] + 0.1 * (ref[1] + ref[2] + ref[3])).backward()
    assert torch.equal(sums.detach(), ref.detach())
    want = torch.cat([ut.grad, it.grad])
    assert float((s1.grad - want).abs().max()) <= 1e-6 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("payload", [True, False])
@pytest.mark.parametrize("n_neg,variant,d", [(1, 2, 64), (3, 2, 64), (1, 0, 32), (2, 2, 128)])
def test_full_batch_over_the_graphs_edges_equals_the_sampled_form(Fn, monkeypatch, n_neg, variant, d, payload):
    """functional.bpr_edge_sums — the full batch of lightgcn.py:91-118 over the training graph's own edge list, backward's
    positive-pair parts as ONE SpMM with per-edge coefficients — against bpr_sums on the same (u, i, j) triples: same
    sums, gradients of both tables equal to 2e-5 of their largest entry.  Duplicate interactions (kept by the raw
    multigraph), isolated users / items, n_neg = 1 / 3, a negative id out of range.  payload: the negatives' item rows from
    the sort that carries (user, coefficient) with the key (gcr_sort_pairs_u64 + gcr_bpr_neg_items_sorted_f32), or from
    the index sort."""
    import recommendation_amd as ra
    monkeypatch.setattr(Fn, "BPR_NEG_PAYLOAD_SORT", payload)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(17 + n_neg)
    n_u, n_i, e = 20_000, 5_000, 300_000
    u = torch.randint(0, n_u - 50, (e,), device=dev, generator=g)            # the last 50 users have no interaction
    i = torch.randint(0, n_i - 20, (e,), device=dev, generator=g)
    u[:100], i[:100] = u[100:200], i[100:200]                                # 100 duplicate pairs
    rows = torch.cat([u, i + n_u])
    cols = torch.cat([i + n_u, u])
    graph = ra.CsrGraph.from_coo(rows, cols, torch.ones(2 * e, device=dev), n_u + n_i, n_u + n_i, dev, symmetric=True)
    uu, ii = graph.user_major_edges(n_u)
    assert uu.numel() == e
    shape = (e,) if n_neg == 1 else (e, n_neg)
    j = torch.randint(0, n_i, shape, device=dev, generator=g)
    j.view(-1)[7] = n_i + 3                                                  # out of range: that sample drops out
    tab_a = (torch.randn(n_u + n_i, d, device=dev, generator=g) * 0.3).requires_grad_()
    tab_b = tab_a.detach().clone().requires_grad_()
    w = torch.tensor([1.0 / e, 1e-4, 2e-4, 3e-4, 0.0], device=dev)
    sa = Fn.bpr_edge_sums(graph, tab_a, n_u, j, variant)
    ub, ib = Fn.split_rows(tab_b, n_u)
    sb = Fn.bpr_sums(ub, ib, uu, ii, j, variant)
    assert torch.equal(sa, sb)
    (sa * w).sum().backward()
    (sb * w).sum().backward()
    ga, gb = tab_a.grad, tab_b.grad
    assert float((ga[:n_u] - gb[:n_u]).abs().max()) <= 2e-5 * float(gb[:n_u].abs().max())
    assert float((ga[n_u:] - gb[n_u:]).abs().max()) <= 2e-5 * float(gb[n_u:].abs().max())
    assert float(ga[n_u - 50:n_u].abs().max()) == 0.0                        # users without an interaction
