"""Randomised stress of the SpMM plan / kernel pair: heavy-tailed degrees around every plan
boundary (empty rows, rows of exactly L and L+1 non-zeros, > 64 rows per partition, hub rows split
into many chunks), all kernel variants (values / all-ones, edge mask, row normalise, fused layer
combine, d = 64 and d != 64), every partition size.  Each case against the float64 oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _degrees(rng, n_rows, L):
    kind = rng.integers(0, 4)
    if kind == 0:
        deg = rng.poisson(3, n_rows)
    elif kind == 1:
        deg = np.minimum((rng.pareto(1.1, n_rows) * 3).astype(np.int64), 20 * L)
    elif kind == 2:
        deg = rng.integers(0, 3, n_rows)              # many tiny rows: > 64 rows would fit one partition
    else:
        deg = rng.poisson(40, n_rows)
    special = rng.choice(n_rows, min(n_rows, 8), replace=False)
    deg[special] = [L, L + 1, L - 1, 0, 2 * L, 2 * L + 1, 7 * L + 3, 1][: special.size]
    return deg.astype(np.int64)


@pytest.mark.parametrize("seed", range(24))
def test_random_graphs_all_variants(seed):
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    rng = np.random.default_rng(seed)
    L = int(rng.choice([64, 128, 256, 512]))
    n_rows = int(rng.integers(1, 900))
    n_cols = int(rng.integers(1, 700))
    d = int(rng.choice([64, 64, 32, 48, 128, 200]))
    deg = _degrees(rng, n_rows, L)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    nnz = int(rowptr[-1])
    col = rng.integers(0, n_cols, nnz).astype(np.int32)
    has_val = bool(rng.integers(0, 2))
    val = rng.standard_normal(nnz).astype(np.float32) if has_val else None
    x = rng.standard_normal((n_cols, d)).astype(np.float32)
    g = ra.CsrGraph(rowptr, col, val, n_rows, n_cols, "cuda", nnz_per_part=L)
    xt = torch.from_numpy(x).cuda()
    vref = val if has_val else np.ones(nnz, np.float32)
    keep, bits = None, None
    if rng.integers(0, 2) and nnz > 0:
        keep = O.edge_keep_mask(nnz, 0.35, seed=seed)
        bits = Fn.edge_mask_bits(nnz, 0.35, seed, "cuda")
    scale = float(rng.choice([1.0, 1.0 / 0.65]))
    ref = O.spmm_csr(rowptr, col, vref, x, keep=keep, scale=scale)
    tol = 1e-5 * max(np.abs(ref).max(), 1e-30)

    y = torch.empty(n_rows, d, device="cuda")
    acc_in = torch.from_numpy(rng.standard_normal((n_rows, d)).astype(np.float32)).cuda()
    acc_out = torch.empty_like(acc_in)
    Fn.spmm_into(g, xt, y=y, acc_in=acc_in, acc_out=acc_out, acc_scale=0.25, val_scale=scale, keep_bits=bits)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(acc_out.cpu().numpy(), (acc_in.cpu().numpy() + ref) * 0.25, rtol=1e-5,
                               atol=1e-5 * max(np.abs(ref).max(), np.abs(acc_in.cpu().numpy()).max()))
    # acc only (no y), in place
    Fn.spmm_into(g, xt, acc_in=acc_in, acc_out=acc_in, acc_scale=1.0, val_scale=scale, keep_bits=bits)
    np.testing.assert_allclose(acc_in.cpu().numpy(), acc_out.cpu().numpy() * 4.0, rtol=2e-5,
                               atol=4e-5 * max(np.abs(ref).max(), 1.0))
    # row normalise + saved inverse norms
    inv = torch.empty(n_rows, device="cuda")
    Fn.spmm_into(g, xt, y=y, l2norm=True, inv_norm_out=inv, val_scale=scale, keep_bits=bits)
    nrm = np.sqrt((ref ** 2).sum(1))
    big = nrm > 1e-6 * max(nrm.max(), 1e-30)
    np.testing.assert_allclose(y.cpu().numpy()[big], (ref / np.maximum(nrm, 1e-12)[:, None])[big], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(inv.cpu().numpy()[big], 1.0 / nrm[big], rtol=1e-4)
    # the plan covers [0, nnz) once
    desc = g.plan.desc_host
    if desc.shape[0]:
        assert desc[0, 0] == 0 and desc[-1, 1] == nnz and np.array_equal(desc[1:, 0], desc[:-1, 1])
        assert (desc[:, 1] - desc[:, 0]).max() <= L


def test_column_bitmap_skips_only_zero_rows():
    """gcr_spmm_csr_acc2_f32 with col_active_bits: a launch whose input is zero outside a few thousand rows (the first
    backward launch of the NCL step) skips the non-zeros of every other column — and a skipped term is w * 0, so the result
    is bit for bit the unmasked one (also through split rows and the fused combine)."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    n_u, n_i = 20000, 1500
    u, i = O.synthetic_interactions(n_u, n_i, 300000, seed=11)
    g = ra.CsrGraph.bipartite_sym_norm(u, i, n_u, n_i, "cuda")
    assert g.plan.n_long > 0
    n = n_u + n_i
    gen = torch.Generator(device="cuda").manual_seed(2)
    rows = torch.randint(0, n, (700,), device="cuda", generator=gen)
    x = torch.zeros(n, 64, device="cuda")
    x[rows] = torch.randn(700, 64, device="cuda", generator=gen)
    acc = torch.randn(n, 64, device="cuda", generator=gen)
    bits = Fn.active_rows_bitmap(rows, n)
    a, b = torch.empty_like(x), torch.empty_like(x)
    Fn.spmm_into(g, x, acc_in=acc, acc_out=a, acc_scale=0.25)
    Fn.spmm_into(g, x, acc_in=acc, acc_out=b, acc_scale=0.25, col_active_bits=bits)
    assert torch.equal(a, b)
    ref = torch.zeros(n, dtype=torch.bool, device="cuda")
    ref[rows] = True
    got = ((bits.view(-1, 1) >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1).reshape(-1)[:n].bool()
    assert torch.equal(got, ref)
