#!/usr/bin/env python3
"""End-to-end NCLModel (the reference's protocol, ncl.py:282-394) on a synthetic data set through the
mirror API: wall time per batch / epoch and the final ranking metrics.  Usage:
    python scripts/train_ncl_synth.py [users] [items] [interactions] [epochs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.ncl import NCLModel

n_u, n_i, n_e, epochs = (int(v) for v in (sys.argv[1:5] + ["200000", "20000", "2000000", "1"][len(sys.argv) - 1:]))
# planted structure so that the metrics mean something: users and items belong to 50 groups, 80 % of a
# user's interactions fall in its own group
rng = np.random.default_rng(0)
groups = 50
ug, ig = rng.integers(0, groups, n_u), rng.integers(0, groups, n_i)
items_of = [np.flatnonzero(ig == g) for g in range(groups)]
u = rng.integers(0, n_u, n_e)
own = rng.random(n_e) < 0.8
i = np.empty(n_e, np.int64)
for g in range(groups):
    sel = np.flatnonzero(own & (ug[u] == g))
    i[sel] = items_of[g][rng.integers(0, len(items_of[g]), sel.size)]
i[~own] = rng.integers(0, n_i, int((~own).sum()))
keys = np.unique(u * n_i + i)
rng.shuffle(keys)
u, i = keys // n_i, keys % n_i
n_test = len(keys) // 10
train = [(f"u{a}", f"i{b}", 1.0) for a, b in zip(u[n_test:], i[n_test:])]
test = [(f"u{a}", f"i{b}", 1.0) for a, b in zip(u[:n_test], i[:n_test])]
seen_u, seen_i = {t[0] for t in train}, {t[1] for t in train}
test = [t for t in test if t[0] in seen_u and t[1] in seen_i]
conf = {"NCL": {"n_layers": 3, "tau": 0.1, "ssl_reg": 1e-6, "proto_reg": 1e-7, "hyper_layers": 1, "alpha": 1.5,
                "num_clusters": 1000}, "batch.size": 2048, "embedding.size": 64, "learning.rate": 0.001,
        "reg.lambda": 1e-4, "max.epoch": epochs, "item.ranking.topN": [10, 20, 50]}
t0 = time.time()
model = NCLModel(conf, train, test)
torch.cuda.synchronize()
print(f"data + graph: {time.time() - t0:.1f} s  ({model.data.user_num} users, {model.data.item_num} items, {len(train)} pairs)")
t0 = time.time()
metrics = model.train()
torch.cuda.synchronize()
dt = time.time() - t0
n_batches = epochs * (len(train) // 2048 + 1)
print(f"train + evaluate: {dt:.1f} s, about {1e3 * dt / n_batches:.1f} ms per batch of 2048 (e_step every batch)")
print(metrics)
