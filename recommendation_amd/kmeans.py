"""k-means E-step of NCL (ncl.py:340-356) on the GPU: `run_kmeans(x)` -> (centroids, assignment).

The reference delegates to `faiss.Kmeans(d, k, gpu=False).train(x)` + `kmeans.index.search(x, 1)`;
faiss is an un-vendored dependency that is not installed here, so parity with it is UNPINNED.  What
is implemented is faiss' published `Clustering::train` with its defaults: if n > 256 k the
centroids are trained on a random subsample of 256 k points (max_points_per_centroid), initial
centroids = a random sample of k training points, niter = 20 Lloyd iterations with L2 nearest
centroid, then EVERY point is assigned against the final centroids (the `index.search`).  Differences:
the random draws are torch's / numpy's, not faiss' generator (an empty cluster is re-seeded from a big one with
faiss' `split_clusters` rule).  The assignment runs on the MFMA tile engine
(gcr_kmeans_assign_f32), the update is float-atomic row adds.
"""
from __future__ import annotations

import torch

from . import _lib
from . import functional as Fn


SORTED_UPDATE_MIN_POINTS = 1 << 16


def kmeans_assign(x, centroids, half_sq):
    L = _lib.lib()
    n, d = x.shape
    assign = torch.empty(n, dtype=torch.int64, device=x.device)
    _lib.check(L.gcr_kmeans_assign_f32(_lib.dptr(x), n, _lib.dptr(centroids), _lib.dptr(half_sq), centroids.shape[0], d,
                                       _lib.dptr(assign), None, _lib.cur_stream(x.device)), "gcr_kmeans_assign_f32")
    return assign


def run_kmeans(x, k, niter=20, seed=1234, init_centroids=None, max_points_per_centroid=256):
    """ncl.py:347-356.  x: float32 [n, d] on the GPU.  Returns (centroids [k', d], assignment int64 [n])
    with k' = min(k, max(2, n // 39)) exactly as ncl.py:350-351 clamps it."""
    _lib.require_cuda(x)
    if x.dim() != 2 or x.dtype != torch.float32:
        raise ValueError("x must be float32 [n, d]")
    n, d_orig = x.shape
    k = min(int(k), max(2, n // 39))
    xp = Fn._pad_dim(x.detach()).contiguous()
    g = torch.Generator(device=x.device).manual_seed(int(seed))
    xt = xp                                            # training set
    if max_points_per_centroid and n > k * max_points_per_centroid:
        xt = xp[torch.randperm(n, device=x.device, generator=g)[:k * max_points_per_centroid]]
    n_train = xt.shape[0]
    if init_centroids is None:
        cent = xt[torch.randperm(n_train, device=x.device, generator=g)[:k]].clone()
    else:
        cent = Fn._pad_dim(init_centroids.detach().to(torch.float32)).contiguous().clone()
    k = cent.shape[0]
    d = xp.shape[1]
    L = _lib.lib()
    half_sq = torch.empty(k, dtype=torch.float32, device=x.device)
    sums = torch.empty(k, d, dtype=torch.float32, device=x.device)
    counts = torch.empty(k, dtype=torch.float32, device=x.device)
    stream = _lib.cur_stream(x.device)

    def update(assign, n_rows):
        if n_rows >= SORTED_UPDATE_MIN_POINTS:
            # points ordered by cluster: one row atomic per run instead of one per point
            keys, perm, _ = Fn._sorted_order(assign, k)
            _lib.check(L.gcr_kmeans_update_sorted_f32(_lib.dptr(xt), n_rows, d, _lib.dptr(keys), _lib.dptr(perm), k,
                                                      _lib.dptr(cent), _lib.dptr(half_sq), _lib.dptr(sums),
                                                      _lib.dptr(counts), stream), "gcr_kmeans_update_sorted_f32")
            return
        _lib.check(L.gcr_kmeans_update_f32(_lib.dptr(xt), n_rows, d, _lib.dptr(assign), k, _lib.dptr(cent),
                                           _lib.dptr(half_sq), _lib.dptr(sums), _lib.dptr(counts), stream),
                   "gcr_kmeans_update_f32")

    def split_empty(it):
        """faiss `split_clusters` (Clustering.cpp): every empty cluster takes over a copy of a big cluster's centroid —
        chosen by walking the clusters and accepting cluster j with probability (size_j - 1) / (n - k) — and the two
        copies are pushed apart by the symmetric perturbation (1 +- 1/1024) alternating over the dimensions; the sizes
        are split in half.  Rare (never at the bench sizes), so it runs on the host: one count read-back per iteration."""
        if not bool((counts == 0).any()):
            return
        import numpy as np
        cnt = counts.cpu().numpy().astype(np.float64)
        rng = np.random.default_rng(int(seed) * 7919 + it)
        eps = 1.0 / 1024.0
        sign = torch.ones(d, device=x.device)
        sign[1::2] = -1.0
        for ci in np.nonzero(cnt == 0)[0]:
            cj = 0
            for _ in range(64 * k):                       # bounded walk (faiss loops until a draw succeeds)
                if rng.random() < (cnt[cj] - 1.0) / max(n_train - k, 1):
                    break
                cj = (cj + 1) % k
            cent[ci] = cent[cj] * (1.0 + eps * sign)
            cent[cj] = cent[cj] * (1.0 - eps * sign)
            cnt[ci] = cnt[cj] // 2
            cnt[cj] -= cnt[ci]
        update(None, 0)                                   # refresh 0.5 |c|^2

    update(None, 0)                      # half_sq of the initial centroids
    for it in range(niter):
        update(kmeans_assign(xt, cent, half_sq), n_train)
        split_empty(it)
    assign = kmeans_assign(xp, cent, half_sq)       # kmeans.index.search(x, 1) against the final centroids
    return cent[:, :d_orig].contiguous(), assign
