#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: whole NCL training steps at cfg3 scale (1M users x
100K items / 10M interactions, B = 2048; ncl.py:311-329 without the per-batch e_step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn, losses as Ls  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
xp = torch.nn.Parameter(torch.empty(n_u + n_i, 64, device=dev))
torch.nn.init.xavier_uniform_(xp)
from recommendation_amd.optim import FusedAdam  # noqa: E402
opt = FusedAdam([xp], lr=1e-3)
gen = torch.Generator(device=dev).manual_seed(1)
bsz = 2048
uidx = torch.randint(0, n_u, (bsz,), device=dev, generator=gen)
iidx = torch.randint(0, n_i, (bsz,), device=dev, generator=gen)
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
jn = Fn.neg_sample(rowptr_u, items_u, uidx, 1, n_i, 3, 0, 101)
cent = torch.randn(1000, 64, device=dev, generator=gen)
u2c = torch.randint(0, 1000, (n_u,), device=dev, generator=gen)
i2c = torch.randint(0, 1000, (n_i,), device=dev, generator=gen)
for _ in range(6):
    final, layers = Fn.lightgcn_propagate(graph, xp, 3, "mean", return_layers=True)
    ue, ie = Fn.split_rows(final, n_u)
    bs = Fn.bpr_sums(ue, ie, uidx, iidx, jn, Fn.BPR_NCL)
    loss = bs[0] / bsz + 1e-4 * (bs[1].sqrt() + bs[2].sqrt() + bs[3].sqrt()) / bsz / bsz + \
        Ls.ssl_layer_loss(layers[2], layers[0], uidx, iidx, n_u, 0.1, 1e-6, 1.0) + \
        Ls.ProtoNCE_loss(layers[0], uidx, iidx, n_u, cent, u2c, cent, i2c, 0.1, 1e-7, bsz)
    opt.zero_grad()
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("ncl probe done")
