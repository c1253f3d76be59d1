"""k-means E-step of NCL (ncl.py:340-356) on the GPU vs the numpy Lloyd restatement.  faiss (the
reference's engine) is not installed: parity with faiss itself is UNPINNED; checked here are the
assignment kernel (exact arg-min incl. ragged centroid counts and ties) and the Lloyd loop."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,k,d", [(5000, 37, 64), (3000, 100, 32), (2000, 33, 128), (1500, 7, 48)])
def test_assignment_is_exact_argmin(n, k, d):
    from recommendation_amd import functional as Fn
    from recommendation_amd.kmeans import kmeans_assign
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    c = rng.standard_normal((k, d)).astype(np.float32)
    c[3] = c[1]                                            # an exact tie: the smaller id must win
    xt, ct = Fn._pad_dim(torch.from_numpy(x).cuda()).contiguous(), Fn._pad_dim(torch.from_numpy(c).cuda()).contiguous()
    half = 0.5 * (ct * ct).sum(1)
    got = kmeans_assign(xt, ct, half).cpu().numpy()
    d2 = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1)
    ref = d2.argmin(1)
    agree = got == ref
    # fp32 near-ties may legitimately flip: the chosen centroid must then be as close within 1e-5
    assert agree.mean() > 0.999
    bad = np.nonzero(~agree)[0]
    assert np.all(d2[bad, got[bad]] <= d2[bad, ref[bad]] * (1 + 1e-5) + 1e-6)
    assert not np.any(got == 3)


def test_lloyd_iterations_match_restatement():
    from recommendation_amd.kmeans import run_kmeans
    rng = np.random.default_rng(0)
    k, d = 20, 64
    centers = rng.standard_normal((k, d)) * 4
    x = (centers[rng.integers(0, k, 8000)] + rng.standard_normal((8000, d))).astype(np.float32)
    init = x[rng.choice(8000, k, replace=False)]
    xt, it = torch.from_numpy(x).cuda(), torch.from_numpy(init).cuda()
    # one Lloyd step: the update must be the exact mean of the members of OUR first assignment,
    # and both assignments must agree with the restatement up to fp32 near-ties
    from recommendation_amd.kmeans import kmeans_assign
    a0 = kmeans_assign(xt, it, 0.5 * (it * it).sum(1)).cpu().numpy()
    ref_a0 = O.kmeans_lloyd(x, init, niter=0)[1]
    assert (a0 == ref_a0).mean() > 0.999
    cent, assign = run_kmeans(xt, k, niter=1, init_centroids=it, max_points_per_centroid=0)   # all 8000 points train
    sums = np.zeros((k, d))
    np.add.at(sums, a0, x.astype(np.float64))
    cnt = np.bincount(a0, minlength=k)
    expect = np.where(cnt[:, None] > 0, sums / np.maximum(cnt, 1)[:, None], init)
    np.testing.assert_allclose(cent.cpu().numpy(), expect, rtol=1e-5, atol=1e-5)
    d2 = ((x[:, None, :].astype(np.float64) - expect[None]) ** 2).sum(-1)
    assert (assign.cpu().numpy() == d2.argmin(1)).mean() > 0.999
    # 20 steps: the trajectory is chaotic at cluster boundaries (a flipped near-tie moves centroids),
    # so compare what k-means optimises: the within-cluster sum of squares
    cent, assign = run_kmeans(xt, k, niter=20, init_centroids=it, max_points_per_centroid=0)
    ref_c, ref_a = O.kmeans_lloyd(x, init, niter=20)
    wss = float(((xt - cent[assign]) ** 2).sum())
    ref_wss = float(((x.astype(np.float64) - ref_c[ref_a]) ** 2).sum())
    assert wss == pytest.approx(ref_wss, rel=2e-3)
    # ncl.py:350-351 clamps k to max(2, n // 39)
    c2, a2 = run_kmeans(torch.from_numpy(x[:100]).cuda(), 2000)
    assert c2.shape == (2, d) and a2.shape == (100,) and int(a2.max()) <= 1


def test_kmeans_at_ncl_scale():
    """NCL e_step size (ncl.py:340-345): all users of cfg2 (1M x 64), k = 2000, 3 iterations; the
    objective must not increase and every point sits with its nearest centroid."""
    from recommendation_amd.kmeans import run_kmeans
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1_000_000, 64, device="cuda", generator=g)
    c1, a1 = run_kmeans(x, 2000, niter=1, seed=5)
    c3, a3 = run_kmeans(x, 2000, niter=3, seed=5)
    obj = lambda c, a: float(((x - c[a]) ** 2).sum(1).mean())
    assert obj(c3, a3) <= obj(c1, a1) + 1e-4
    sub = torch.arange(0, 1_000_000, 997, device="cuda")
    d2 = torch.cdist(x[sub].double(), c3.double()) ** 2
    assert bool((d2.gather(1, a3[sub, None]).squeeze(1) <= d2.min(1).values * (1 + 1e-5) + 1e-6).all())


def test_training_subsample_like_faiss():
    """faiss.Clustering trains on at most 256 points per centroid and then assigns every point
    (index.search): with n = 40 x 256 k the subsample path runs; well-separated blobs must still be
    recovered, every point must sit with its nearest centroid, and switching the cap off must give the
    same partition quality."""
    from recommendation_amd.kmeans import run_kmeans
    rng = np.random.default_rng(3)
    k, d, n = 8, 64, 8 * 256 * 40
    centers = rng.standard_normal((k, d)) * 6
    lab = rng.integers(0, k, n)
    x = torch.from_numpy((centers[lab] + rng.standard_normal((n, d))).astype(np.float32)).cuda()
    init = x[torch.from_numpy(np.array([np.flatnonzero(lab == c)[0] for c in range(k)])).cuda()]
    c_sub, a_sub = run_kmeans(x, k, niter=20, init_centroids=init)
    c_all, a_all = run_kmeans(x, k, niter=20, init_centroids=init, max_points_per_centroid=0)
    assert a_sub.shape == (n,)
    assert float((a_sub.cpu() == torch.from_numpy(lab)).float().mean()) > 0.999     # init[c] came from blob c
    assert float((a_sub == a_all).float().mean()) > 0.999
    np.testing.assert_allclose(c_sub.cpu().numpy(), centers, atol=0.35)            # means of ~256 samples each
    d2 = torch.cdist(x[::97].double(), c_sub.double()) ** 2
    assert bool((d2.gather(1, a_sub[::97, None]).squeeze(1) <= d2.min(1).values * (1 + 1e-5) + 1e-6).all())


def test_empty_cluster_is_reseeded_from_a_big_one():
    """faiss' split_clusters rule (Clustering.cpp): a centroid that attracts no point takes over a perturbed copy of a
    populated cluster's centroid instead of staying dead (ncl.py:352 relies on faiss.Kmeans for this)."""
    from recommendation_amd.kmeans import run_kmeans
    rng = np.random.default_rng(1)
    d, k = 64, 4
    centers = rng.standard_normal((3, d)) * 5
    x = (centers[rng.integers(0, 3, 3000)] + 0.3 * rng.standard_normal((3000, d))).astype(np.float32)
    init = np.concatenate([centers, np.full((1, d), 1e3)]).astype(np.float32)      # the 4th start is far from every point
    cent, assign, info = run_kmeans(torch.from_numpy(x).cuda(), k, niter=10, init_centroids=torch.from_numpy(init).cuda(),
                                    return_info=True)
    counts = np.bincount(assign.cpu().numpy(), minlength=k)
    assert (counts > 0).all(), counts
    assert float(cent.abs().max()) < 100.0                  # the dead centroid at 1e3 is gone
    assert int(info["n_split"]) >= 1                        # counted on the device, never read during the iterations


def test_split_step_matches_restatement_bit_for_bit():
    """One Lloyd update with forced empty clusters against oracle_np.kmeans_split_clusters: same Philox trials, so the
    SAME donor clusters are picked and the perturbed centroids agree to fp32 rounding; no empty cluster is left after
    one more iteration (the re-seed must never be a no-op)."""
    from recommendation_amd.kmeans import run_kmeans
    rng = np.random.default_rng(7)
    d, k, n = 64, 12, 6000
    centers = rng.standard_normal((8, d)) * 5
    lab = rng.integers(0, 8, n)
    x = (centers[lab] + 0.3 * rng.standard_normal((n, d))).astype(np.float32)
    far = 1e3 * (1 + np.arange(4))[:, None] * np.ones((4, d))
    init = np.concatenate([centers, far]).astype(np.float32)                      # clusters 8..11 start dead
    for seed in (1234, 99):
        cent, assign, info = run_kmeans(torch.from_numpy(x).cuda(), k, niter=1, seed=seed,
                                        init_centroids=torch.from_numpy(init).cuda(), max_points_per_centroid=0,
                                        return_info=True)
        ns = []
        ref_c, ref_a = O.kmeans_lloyd(x, init, niter=1, split_seed=seed, n_split_out=ns)
        assert int(info["n_split"]) == ns[0] == 4
        np.testing.assert_allclose(cent.cpu().numpy(), ref_c, rtol=2e-6, atol=1e-6)
        cent2, assign2 = run_kmeans(torch.from_numpy(x).cuda(), k, niter=2, seed=seed,
                                    init_centroids=torch.from_numpy(init).cuda(), max_points_per_centroid=0)
        assert (np.bincount(assign2.cpu().numpy(), minlength=k) > 0).all()


def test_split_falls_back_to_the_largest_cluster():
    """n == k: the acceptance probability (size - 1) / (n - k) is undefined, no walk can end, and the re-seed falls back
    to the largest cluster (never a no-op, never a cluster with <= 1 point)."""
    from recommendation_amd import _lib
    k, d = 5, 64
    cent = torch.arange(k * d, dtype=torch.float32, device="cuda").reshape(k, d) / 100 + 1
    half = torch.empty(k, device="cuda")
    sums = torch.zeros(k, d, device="cuda")
    counts = torch.zeros(k, device="cuda")
    x = cent[[0, 1, 2, 3, 3]].clone().contiguous()
    assign = torch.tensor([0, 1, 2, 3, 3], device="cuda")
    ns = torch.zeros(1, dtype=torch.int32, device="cuda")
    L = _lib.lib()
    _lib.check(L.gcr_kmeans_lloyd_update_f32(_lib.dptr(x), 5, d, _lib.dptr(assign), None, None, k, _lib.dptr(cent), _lib.dptr(half),
                                             _lib.dptr(sums), _lib.dptr(counts), 1, 1234, 0, _lib.dptr(ns), _lib.cur_stream()), "lloyd")
    ref_c = (np.arange(k * d, dtype=np.float64).reshape(k, d) / 100 + 1).astype(np.float32).astype(np.float64)
    cnt = np.array([1, 1, 1, 2, 0], dtype=np.float64)
    assert O.kmeans_split_clusters(ref_c, cnt, 5, 1234, 0) == 1
    assert cnt.tolist() == [1, 1, 1, 1, 1]
    assert int(ns) == 1
    np.testing.assert_allclose(cent.cpu().numpy(), ref_c, rtol=2e-6)
    np.testing.assert_allclose(half.cpu().numpy(), 0.5 * (ref_c ** 2).sum(1), rtol=1e-5)
    assert float(counts.abs().sum()) == 0.0 and float(sums.abs().sum()) == 0.0      # scratch left zeroed


@pytest.mark.parametrize("d", [64, 32])
@pytest.mark.parametrize("n,k", [(76800, 300), (1000, 37), (129, 33), (5000, 600)])
def test_image_search_gives_the_tiled_search_bit_for_bit(n, k, d):
    """The wave-independent search over the pre-split centroid image (both register budgets) runs the same products in the
    same order as the tiled kernel: identical assignments, ties and ragged centroid tiles included."""
    from recommendation_amd import _lib
    from recommendation_amd.kmeans import kmeans_assign
    L = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(n + k + d)
    x = torch.randn(n, d, device="cuda", generator=g)
    c = x[torch.randperm(n, device="cuda", generator=g)[:k]].clone()
    c[3] = c[1]
    half = 0.5 * (c * c).sum(1)
    ref = kmeans_assign(x, c, half)
    img = torch.empty(int(L.gcr_kmeans_image_bytes(k, d)), dtype=torch.uint8, device="cuda")
    st = _lib.cur_stream(x.device)
    _lib.check(L.gcr_kmeans_centroid_image_f32(_lib.dptr(c), _lib.dptr(half), k, d, _lib.dptr(img), st), "image")
    for flags in (0, 1):
        got = torch.full((n,), -7, dtype=torch.int64, device="cuda")
        _lib.check(L.gcr_kmeans_search_image_f32(_lib.dptr(x), n, _lib.dptr(img), k, d, _lib.dptr(got), None, None, 1, flags, st), "search")
        assert torch.equal(got, ref), flags
    assert not bool((ref == 3).any())


def test_incremental_fixed_point_sums_equal_a_fresh_accumulation():
    """run_kmeans' default path at the e_step's kind of size: sums kept in 64-bit fixed point, touched only by points that
    changed cluster.  After every iteration count the centroids must be the exact means of the CURRENT assignment (no drift
    from 25 rounds of +/- updates), two runs must agree bitwise, and one iteration must agree with the float-atomic path."""
    from recommendation_amd import kmeans as K
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.nn.functional.normalize(torch.randn(30000, 64, device="cuda", generator=g) + 0.4, dim=1)
    init = x[torch.randperm(30000, device="cuda", generator=g)[:120]].clone()
    assert K.IMAGE_SEARCH and K.INCREMENTAL_UPDATE
    for niter in (1, 7, 25):
        c1, a1 = K.run_kmeans(x, 120, niter=niter, init_centroids=init, max_points_per_centroid=0)
        c2, _ = K.run_kmeans(x, 120, niter=niter, init_centroids=init, max_points_per_centroid=0)
        assert torch.equal(c1, c2)
        # the centroids are the means of the assignment of iteration `niter` (against the previous centroids): recompute
        # that assignment by running niter - 1 iterations and assigning once more
        cp, _ = K.run_kmeans(x, 120, niter=niter - 1, init_centroids=init, max_points_per_centroid=0) if niter > 1 else (init, None)
        a = K.assign_to_centroids(x, cp)
        sums = torch.zeros(120, 64, dtype=torch.float64, device="cuda").index_add_(0, a, x.double())
        cnt = torch.bincount(a, minlength=120).double()
        expect = torch.where(cnt[:, None] > 0, sums / cnt.clamp_min(1)[:, None], cp.double())
        assert float((c1.double() - expect).abs().max()) <= 2e-7
    try:
        K.INCREMENTAL_UPDATE = False
        cf, _ = K.run_kmeans(x, 120, niter=1, init_centroids=init, max_points_per_centroid=0)
        K.IMAGE_SEARCH = False
        ct, _ = K.run_kmeans(x, 120, niter=1, init_centroids=init, max_points_per_centroid=0)
    finally:
        K.IMAGE_SEARCH = K.INCREMENTAL_UPDATE = True
    c1, _ = K.run_kmeans(x, 120, niter=1, init_centroids=init, max_points_per_centroid=0)
    assert float((c1 - cf).abs().max()) <= 1e-6 and float((c1 - ct).abs().max()) <= 1e-6
