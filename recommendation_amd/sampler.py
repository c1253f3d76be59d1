"""Batch iterators and augmentations with the reference's generator protocol, driven by the device
Philox kernels (gcr_neg_sample / gcr_edge_mask_bits) instead of per-sample Python loops.

  next_batch_pairwise   ncl.py:91-114, ssl4rec.py:33-50, gcl.py:111-125, univariate/sept.py:12-30
  randint_negatives     lightgcn.py:91-94
  EdgeRemoving          gcl.py:18-25
"""
from __future__ import annotations

import torch

from . import functional as Fn

_MASK64 = 2 ** 64 - 1


def next_batch_pairwise(data, batch_size, n_negs=1, *, seed=0, epoch=0, max_trials=101, drop_incomplete=True, prefetch=32):
    """Yields (u_idx, i_idx, j_idx) int64 device tensors per batch of shuffled training pairs
    (j_idx has batch * n_negs entries, ssl4rec.py:43-47).  Negatives are uniform over items and
    never a training positive of the user; `max_trials` = 101 and `drop_incomplete` restate ncl.py's
    bail-out (a batch with an exhausted slot is skipped, ncl.py:110-114); gcl.py / ssl4rec.py retry
    forever, which a large `max_trials` reproduces.  The shuffle and the draws are functions of
    (seed, epoch), so runs are reproducible (the reference never seeds its generators)."""
    dev = data.device
    n = data.uid_dev.numel()
    g = torch.Generator(device=dev).manual_seed((int(seed) * 1_000_003 + int(epoch)) & (2 ** 63 - 1))
    perm = torch.randperm(n, device=dev, generator=g)
    base = (int(epoch) << 40) & _MASK64
    # negatives are drawn for `chunk` batches per launch (slot s of the epoch uses the same counter whatever the launch
    # boundaries, so the draws equal the per-batch ones) and the incomplete-batch test is ONE read-back per chunk
    # instead of one per batch — a per-batch read-back would drain the stream in front of every training step
    chunk = max(1, int(prefetch)) * batch_size
    for c0 in range(0, n, chunk):
        sel_c = perm[c0:c0 + chunk]
        u_c, i_c = data.uid_dev[sel_c], data.iid_dev[sel_c]
        j_c = Fn.neg_sample(data.user_rowptr, data.user_items_sorted, u_c, n_negs, data.item_num,
                            seed, base + c0 * n_negs, max_trials)
        n_b = (sel_c.numel() + batch_size - 1) // batch_size
        bad = None
        if max_trials > 0 and drop_incomplete:
            miss = (j_c < 0).view(-1, n_negs).any(1)
            pad = n_b * batch_size - miss.numel()
            if pad:
                miss = torch.cat([miss, miss.new_zeros(pad)])
            bad = miss.view(n_b, batch_size).any(1).tolist()
        for b in range(n_b):
            if bad is not None and bad[b]:
                continue
            lo, hi = b * batch_size, min((b + 1) * batch_size, sel_c.numel())
            yield u_c[lo:hi], i_c[lo:hi], j_c[lo * n_negs:hi * n_negs]


def randint_negatives(num_samples, num_items, n_neg=1, *, seed=0, step=0, device="cuda"):
    """lightgcn.py:91-94: `torch.randint(0, num_items, (E,))` or `(E, n_neg)` — uniform, positives
    NOT rejected — from the same counter RNG (one slot per (sample, k))."""
    dev = torch.device(device)
    dummy_rowptr = torch.zeros(2, dtype=torch.int64, device=dev)
    u = torch.zeros(num_samples, dtype=torch.int64, device=dev)
    out = Fn.neg_sample(dummy_rowptr, None, u, n_neg, num_items, seed, (int(step) << 40) & _MASK64, 0)
    return out if n_neg == 1 else out.view(num_samples, n_neg)


class EdgeView:
    """An edge_index with a per-edge keep bitmap instead of a re-indexed copy (what
    gcl.EdgeRemoving returns, consumed lazily).  `materialize()` gives the reference's tensor."""

    def __init__(self, edge_index, keep_bits, pe):
        self.edge_index, self.keep_bits, self.pe = edge_index, keep_bits, pe

    def keep_mask(self):
        n = self.edge_index.size(1)
        shifts = torch.arange(32, device=self.keep_bits.device, dtype=torch.int32)
        bits = (self.keep_bits.unsqueeze(1) >> shifts) & 1
        return bits.reshape(-1)[:n].bool()

    def materialize(self):
        return self.edge_index[:, self.keep_mask()]

    def size(self, dim=None):
        kept = int(self.keep_mask().sum())
        shape = (2, kept)
        return shape if dim is None else shape[dim]


class EdgeRemoving:
    """gcl.py:18-25: Bernoulli keep mask `rand(num_edges) >= pe` over the columns of edge_index,
    every directed edge independently.  Each call advances a counter so that two calls give two
    different views (gcl.py:209-210)."""

    def __init__(self, pe=0.2, seed=0):
        self.pe, self.seed, self.calls = pe, seed, 0

    def __call__(self, edge_index):
        self.calls += 1
        bits = Fn.edge_mask_bits(edge_index.size(1), self.pe, (self.seed * 0x9E3779B97F4A7C15 + self.calls) & _MASK64,
                                 edge_index.device)
        return EdgeView(edge_index, bits, self.pe)
