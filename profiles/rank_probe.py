#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: full-ranking evaluation of 100K users against 100K items,
top-50, training positives masked (lightgcn.py:48-57 / ncl.py:253-264 at cfg2 scale) — the fused path
(gcr_rank_fused_f32) and, for comparison, the two-call path that materialises the score chunks."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import evaluate as ev  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
x0 = torch.empty(n_u + n_i, 64, device=dev)
torch.nn.init.xavier_uniform_(x0, generator=torch.Generator(device=dev).manual_seed(0))
ut, it = x0[:n_u].contiguous(), x0[n_u:].contiguous()
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
q = torch.arange(0, 100000, device=dev)
for fused in (True, False):
    ev.FUSED_RANK = fused
    ev.rank_topk(ut, it, q, rowptr_u, items_u, 50)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        ti, ts = ev.rank_topk(ut, it, q, rowptr_u, items_u, 50)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"fused={fused}: {1e3 * dt:.2f} ms per 100K users = {q.numel() / dt / 1e6:.2f} M users/s")
    if fused:
        ref_i = ti.clone()
    else:
        same = float((ref_i == ti).float().mean())
        print(f"item lists identical to the two-call path: {100 * same:.3f} %")
