#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: lightgcn.py's full-batch training step (lightgcn.py:91-118) at cfg2
scale exactly as bench.py's `lightgcn_full_batch_step_ms` leg runs it: fresh negatives, 3-layer propagation, -log sigmoid BPR
over all 10M training edges + L2, backward, Adam.  6 steps."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn  # noqa: E402
from recommendation_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
x0 = torch.empty(n_u + n_i, 64, device=dev)
torch.nn.init.xavier_uniform_(x0, generator=torch.Generator(device=dev).manual_seed(0))
xp = torch.nn.Parameter(x0.clone())
opt = FusedAdam([xp], lr=1e-3)
gen = torch.Generator(device=dev).manual_seed(1)
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
eu = torch.repeat_interleave(torch.arange(n_u, device=dev), rowptr_u[1:] - rowptr_u[:-1])
ei = items_u.to(torch.int64)
for _ in range(6):
    neg = torch.randint(0, n_i, (eu.numel(),), device=dev, generator=gen)
    final = Fn.lightgcn_propagate(graph, xp, wl["layers"], "sum")
    if "--sampled" in sys.argv:                                              # the general (u, i, j)-triple form, for A/B
        ue, ie = Fn.split_rows(final, n_u)
        s = Fn.bpr_sums(ue, ie, eu, ei, neg, Fn.BPR_LOG_SIGMOID)
    else:
        s = Fn.bpr_edge_sums(graph, final, n_u, neg, Fn.BPR_LOG_SIGMOID)    # (eu, ei) = the graph's own edge list
    loss = s[0] / eu.numel() + 1e-4 * (s[1] + s[2]) / eu.numel()
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("lightgcn step probe done: 6 steps,", eu.numel(), "edges")
