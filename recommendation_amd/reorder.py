"""Locality-aware renumbering of users and items, carried as a permutation (SURVEY §7 "Gather locality", §8f.3).

The reference numbers users and items by sorted raw id (ncl.py:60-61) or first appearance (selfcf.py:281-288): ids
that say nothing about who interacts with whom.  The SpMM is a row gather — an item row gathers its users' embedding
rows, a user row its items' — so with such ids every gather of the 2.6 GB user table (cfg4) goes to the memory system:
46 GB of fabric traffic per layer against 7.3 GB of compulsory bytes.  Real interaction graphs are not
uniform: users form taste communities that concentrate on subsets of the items.  Numbering the members of a community
contiguously, and walking the work plan community by community on ONE XCD (graph.SpmmPlan `row_group`), turns most
gathers of a community's rows into hits in that XCD's 4 MB L2.

The communities are found with the hot path's own kernels — spectral co-clustering (Dhillon 2001) on the operator the
model propagates with anyway, A_hat = D^-1/2 (R + R^T) D^-1/2 (selfcf.py:240-255):
  1. subspace iteration X <- (X + A_hat X) / 2 (gcr_spmm_csr_acc2_f32, the Horner epilogue) on `dim` random vectors,
     with the trivial eigenvector D^1/2 1 projected out and a Cholesky re-orthonormalisation every few steps;
  2. Rayleigh-Ritz rotation, rows scaled by D^-1/2 and normalised (Ng-Jordan-Weiss);
  3. k-means of the N rows (kmeans.run_kmeans, the NCL e_step's kernels) -> one label per user and per item.
The permutation sorts users by (label, degree descending), items likewise.  It is NOT part of the reference's
semantics: `Interaction(reorder=...)` applies it to the dense ids it hands out, keeps old -> new in `perm_user` /
`perm_item`, and tests/test_reorder_gpu.py checks that un-permuting the operator gives the reference's neighbour lists bit
for bit (tests/golden/graph_build.npz).  On a graph without community structure (the uniform synthetic benchmark graph)
no numbering helps and none is applied by default.
"""
from __future__ import annotations

import numpy as np
import torch

from . import functional as Fn
from .graph import CsrGraph
from .kmeans import run_kmeans

DEFAULT_ROWS_PER_CLUSTER = 8192       # 8192 x 256 B = 2 MB of user rows: half of one XCD's 4 MB L2


def spectral_labels(uid, iid, n_users, n_items, device, n_clusters, dim=64, iters=14, reortho_every=4, seed=0,
                    graph=None, min_structure=None, info=None):
    """int64 [n_users + n_items] co-cluster label of every user and item (device tensor).
    min_structure: give up BEFORE the k-means (return (None, deg)) when the largest Ritz value of the iterated subspace is
    below `min_structure` x the noise bulk edge of a structureless graph with these mean degrees (`noise_bulk_edge`): no
    community direction stands out of the bulk, no numbering can help.  info: a dict that receives the evidence."""
    dev = torch.device(device)
    n = n_users + n_items
    g = graph if graph is not None else CsrGraph.bipartite_sym_norm(uid, iid, n_users, n_items, dev)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).to(torch.float32)
    v0 = deg.sqrt()
    v0 = (v0 / v0.norm()).unsqueeze(1)                                   # A_hat v0 = v0 (the trivial eigenvector)
    gen = torch.Generator(device=dev).manual_seed(int(seed))
    x = torch.randn(n, dim, device=dev, generator=gen)

    def orthonormalise(x):
        x = x - v0 * (v0.t() @ x)
        gram = (x.t() @ x).double().cpu()                                # dim x dim: factorised on the host (one-off ingest)
        gram = gram + 1e-10 * torch.trace(gram) / dim * torch.eye(dim, dtype=torch.float64)
        r = torch.linalg.cholesky(gram).t()                              # x = q r
        return x @ torch.linalg.inv(r).to(device=dev, dtype=torch.float32)

    x = orthonormalise(x)
    for it in range(iters):
        nxt = torch.empty_like(x)
        Fn.spmm_into(g, x, acc_in=x, acc_out=nxt, acc_scale=0.5)          # (x + A_hat x) / 2: spectrum mapped to [0, 1]
        x = nxt
        if (it + 1) % reortho_every == 0 or it == iters - 1:
            x = orthonormalise(x)
    ax = torch.empty_like(x)
    Fn.spmm_into(g, x, y=ax)
    m = (x.t() @ ax).double().cpu()
    evals, evecs = torch.linalg.eigh((m + m.t()) / 2)                    # Rayleigh-Ritz on the iterated subspace
    bulk = noise_bulk_edge(n_users, n_items, g.nnz // 2)
    if info is not None:
        info.update(ritz_top=round(float(evals.max()), 4), ritz_median=round(float(evals.median()), 4),
                    noise_bulk_edge=round(bulk, 4))
    if min_structure is not None and float(evals.max()) < float(min_structure) * bulk:
        return None, deg
    keep = evecs[:, evals > max(0.05, float(evals.max()) * 0.25)]        # drop directions that are still noise
    if keep.shape[1] < 2:
        keep = evecs[:, -min(dim, 8):]
    z = x @ keep.to(device=dev, dtype=torch.float32)
    dinv = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg)).unsqueeze(1)
    z = z * dinv
    z = z / z.norm(dim=1, keepdim=True).clamp_min(1e-20)
    _, labels = run_kmeans(z.contiguous(), int(n_clusters), niter=15, seed=int(seed) + 1)
    return labels, deg


def noise_bulk_edge(n_users, n_items, n_edges):
    """Edge of the singular-value bulk of D_u^-1/2 R D_i^-1/2 for a bipartite graph WITHOUT structure (edges independent
    given the degrees): 1 / sqrt(mean user degree) + 1 / sqrt(mean item degree) (Marchenko-Pastur scaling of a random
    rectangular matrix with these row / column sums).  Community directions of a planted partition with in-community
    probability p sit near p, far above it when p is large."""
    du, di = max(n_edges / max(n_users, 1), 1e-9), max(n_edges / max(n_items, 1), 1e-9)
    return du ** -0.5 + di ** -0.5


def order_from_labels(labels, deg, n_users):
    """(perm_user [U], perm_item [I], group_of_new_row [N]): old dense id -> new dense id per side (members of a label
    contiguous, larger degree first inside a label) and the label of every row of the re-numbered operator."""
    n = labels.numel()
    big = int(deg.max().item()) + 2
    key = labels.to(torch.int64) * big + (big - 1 - deg.to(torch.int64))
    out = []
    group = torch.empty(n, dtype=torch.int64, device=labels.device)
    for lo, hi in ((0, n_users), (n_users, n)):
        order = torch.argsort(key[lo:hi], stable=True)                   # new position -> old id
        perm = torch.empty_like(order)
        perm[order] = torch.arange(hi - lo, device=labels.device)
        out.append(perm)
        group[lo:hi] = labels[lo:hi][order]
    return out[0], out[1], group


def locality_permutation(uid, iid, n_users, n_items, device, rows_per_cluster=DEFAULT_ROWS_PER_CLUSTER, seed=0, **kw):
    """Spectral co-clustering of the interaction graph -> (perm_user, perm_item, group_of_new_row), device int64 tensors.
    uid / iid: dense ids of the interactions (numpy or tensors)."""
    n_clusters = max(2, -(-int(n_users) // int(rows_per_cluster)))
    labels, deg = spectral_labels(uid, iid, n_users, n_items, device, n_clusters, seed=seed, **kw)
    return order_from_labels(labels, deg, n_users)


def xcd_grouped_order(desc_host, row_group_host, n_xcd=8, waves_per_block=4):
    """Order of the SpMM partitions that keeps every row group on one XCD (host, one-off per graph).

    gcr_spmm's workgroup b runs partitions 4 b .. 4 b + 3 and workgroups are dealt round-robin over the 8 XCDs (observed
    placement, used for speed only: nothing depends on it for correctness — every partition writes its own rows whatever
    the order).  Groups are spread over the XCDs by partition count (largest first onto the lightest XCD), each XCD's list
    walks its groups one after the other, and the lists are interleaved four partitions at a time; a list that runs out
    is padded with empty partitions (-1 in the returned index array)."""
    row0 = (desc_host[:, 2] & 0xFFFFFFFF).astype(np.int64)
    pgroup = np.asarray(row_group_host)[row0]
    groups, inv = np.unique(pgroup, return_inverse=True)
    count = np.bincount(inv, minlength=groups.size)
    load = np.zeros(n_xcd, dtype=np.int64)
    xcd_of = np.empty(groups.size, dtype=np.int64)
    for gi in np.argsort(-count, kind="stable"):
        x = int(np.argmin(load))
        xcd_of[gi] = x
        load[x] += count[gi]
    part_xcd = xcd_of[inv]
    # inside an XCD: group by group (in label order), natural partition order inside a group
    lists = [np.flatnonzero(part_xcd == x) for x in range(n_xcd)]
    lists = [l[np.argsort(inv[l], kind="stable")] for l in lists]
    blocks = max((l.size + waves_per_block - 1) // waves_per_block for l in lists)
    order = np.full((blocks, n_xcd, waves_per_block), -1, dtype=np.int64)
    for x, l in enumerate(lists):
        padded = np.full(blocks * waves_per_block, -1, dtype=np.int64)
        padded[: l.size] = l
        order[:, x, :] = padded.reshape(blocks, waves_per_block)
    return order.reshape(-1)


def _layer_ms(graph, x, reps=3):
    y = torch.empty_like(x)
    Fn.spmm_into(graph, x, y=y)
    torch.cuda.synchronize(x.device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Fn.spmm_into(graph, x, y=y)
    e1.record()
    torch.cuda.synchronize(x.device)
    return e0.elapsed_time(e1) / reps


def guarded_locality_permutation(uid, iid, n_users, n_items, device, rows_per_cluster=DEFAULT_ROWS_PER_CLUSTER, seed=0,
                                 min_structure=0.9, min_gain=0.03, d=64, graph=None):
    """`locality_permutation` that can say no: returns (perm_user, perm_item, group, decision) with the three tensors None
    when the renumbering is NOT to be applied.  It must never slow a graph down, so it is kept only if
      1. the graph is larger than the caches a numbering could help with (>= 4 clusters of `rows_per_cluster` rows),
      2. the Ritz spectrum of the subspace iteration reaches the noise bulk of a structureless graph (`min_structure` x
         `noise_bulk_edge`, checked before the k-means is paid for; measured at cfg2 sizes, scripts/perf_reorder.py --sweep:
         uniform 0.86 x the edge, planted communities with p_in = 0.5 / 0.7 / 0.85 at 1.03 / 1.67 / 2.06 x — the threshold
         only weeds out the clearly structureless, the measurement below decides the rest), and
      3. MEASURED: one SpMM layer (d columns) on the renumbered operator with its XCD-grouped plan is at least `min_gain`
         faster than on the operator as it came.
    decision: {"applied": bool, "reason": str, ...evidence} — logged by the caller."""
    dev = torch.device(device)
    decision = {"applied": False}
    n_clusters = max(2, -(-int(n_users) // int(rows_per_cluster)))
    if n_clusters < 4:
        decision["reason"] = f"{n_users} user rows are {n_clusters} cluster(s) of {rows_per_cluster}: the tables fit the caches"
        return None, None, None, decision
    g0 = graph if graph is not None else CsrGraph.bipartite_sym_norm(uid, iid, n_users, n_items, dev)
    labels, deg = spectral_labels(uid, iid, n_users, n_items, dev, n_clusters, seed=seed, graph=g0,
                                  min_structure=min_structure, info=decision)
    if labels is None:
        decision["reason"] = ("largest Ritz value %.3f is below %.2f x the noise bulk edge %.3f of a structureless graph with "
                              "these degrees: no community direction to number by" %
                              (decision["ritz_top"], min_structure, decision["noise_bulk_edge"]))
        return None, None, None, decision
    pu, pi, group = order_from_labels(labels, deg, n_users)
    u, i = torch.as_tensor(uid, device=dev), torch.as_tensor(iid, device=dev)
    g1 = CsrGraph.bipartite_sym_norm(pu[u], pi[i], n_users, n_items, dev, row_group=group)
    x = torch.randn(n_users + n_items, d, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    t0, t1 = _layer_ms(g0, x), _layer_ms(g1, x)
    decision.update(ms_per_layer_before=round(t0, 4), ms_per_layer_after=round(t1, 4))
    if t1 > (1.0 - min_gain) * t0:
        decision["reason"] = "measured: %.3f -> %.3f ms per layer is not a gain of %.0f %%" % (t0, t1, 100 * min_gain)
        return None, None, None, decision
    decision.update(applied=True, reason="measured: %.3f -> %.3f ms per layer" % (t0, t1))
    return pu, pi, group, decision
