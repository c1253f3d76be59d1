"""CPU-baseline legs of SURVEY.md §8(d) beyond the SpMM: the reference's own dense formulations of
the contrast losses and its Python rejection sampler, restated for TIMING on the host cores of
the GPU box (the reference files do not travel there).

Test infrastructure: only bench.py's `cpu_baseline` leg and tests/ import this module; the
product path never does.  Each function cites the reference lines it follows; values are checked
against oracle_np in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import random

import torch
import torch.nn.functional as F


def ncl_structure_denominator(anchors, table, tau):
    """ncl.py:363-364: `torch.exp(torch.matmul(norm_cu, F.normalize(iu).T) / t).sum(1)` — the
    B x U dense logits of NCLModel.ssl_layer_loss (materialised, naive exp)."""
    return torch.exp(torch.matmul(F.normalize(anchors), F.normalize(table).T) / tau).sum(1)


def gcl_info_nce_loss(z1, z2, temp=0.2):
    """gcl.py:28-35: symmetric InfoNCE over all rows (dense N x N similarity, two cross-entropies)."""
    z1 = F.normalize(z1, dim=1)
    z2 = F.normalize(z2, dim=1)
    sim = torch.mm(z1, z2.t()) / temp
    labels = torch.arange(z1.size(0))
    return (F.cross_entropy(sim, labels) + F.cross_entropy(sim.T, labels)) / 2


def python_pairwise_sampler(training_data, user_map, item_map, training_set_u, batch_size, rng=None):
    """ncl.py:91-114 `next_batch_pairwise`: shuffle, then per positive draw `choice(list(item keys))`
    until the item is not in the user's training set (<= 101 trials).  The per-draw
    `list(data.item.keys())` rebuild (O(I) per trial) is kept: it is what the reference
    pays.  Generator of (u_idx, i_idx, j_idx) lists."""
    rng = rng or random
    training_data = list(training_data)
    rng.shuffle(training_data)
    ptr = 0
    while ptr < len(training_data):
        batch_end = min(ptr + batch_size, len(training_data))
        batch = training_data[ptr:batch_end]
        ptr = batch_end
        u_idx, i_idx, j_idx = [], [], []
        for user, item in batch:
            u_idx.append(user_map[user])
            i_idx.append(item_map[item])
            neg_trials = 0
            while True:
                neg_item = rng.choice(list(item_map.keys()))
                if neg_item not in training_set_u[user]:
                    j_idx.append(item_map[neg_item])
                    break
                neg_trials += 1
                if neg_trials > 100:
                    break
        if len(u_idx) == len(i_idx) == len(j_idx) and u_idx:
            yield u_idx, i_idx, j_idx
