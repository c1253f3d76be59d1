#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: whole NCL training iterations at cfg3 scale (1M users x 100K items /
10M interactions, B = 2048) exactly as bench.py's `ncl_train_step_full_ms` leg runs them: NCLModel.train_step = the loop
body ncl.py:311-329 INCLUDING the per-batch e_step, on the hand-derived launch sequence (ncl_step.FusedNCLStep), eager
(a graph replay shows up as one opaque graph launch per step in some rocprofv3 versions).
usage: ncl_step_probe.py [num_clusters] [autograd|graph]   (graph: the captured step replayed, as bench.py times it)"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn  # noqa: E402
from recommendation_amd.ncl import NCLModel  # noqa: E402
from recommendation_amd.optim import FusedAdam  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fused = not (len(sys.argv) > 2 and sys.argv[2] == "autograd")
capture = len(sys.argv) > 2 and sys.argv[2] == "graph"
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
conf = copy.deepcopy(bench.NCL_CFG3)
conf["NCL"]["num_clusters"] = k
model = NCLModel.from_graph(conf, graph, n_u, n_i, graph_capture=capture)
opt = FusedAdam(model.model.parameters(), lr=1e-3, capturable=capture)
gen = torch.Generator(device=dev).manual_seed(1)
bsz = 2048
uidx = torch.randint(0, n_u, (bsz,), device=dev, generator=gen)
iidx = torch.randint(0, n_i, (bsz,), device=dev, generator=gen)
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
jn = Fn.neg_sample(rowptr_u, items_u, uidx, 1, n_i, 3, 0, 101)
model.e_step()
if len(sys.argv) > 3:                              # e_step issue point of the fused step: late (default) | early
    model.train_step((uidx, iidx, jn), opt, check_negatives=False, fused=fused)
    model._fused.e_step_issue = sys.argv[3]
for _ in range(9 if capture else 6):              # graph: 2 eager warm-ups + the capture, then 6 replays
    model.train_step((uidx, iidx, jn), opt, check_negatives=False, fused=fused)
torch.cuda.synchronize()
print("ncl probe done: 1 initial e_step + 6 steps, num_clusters", k, "fused" if fused else "autograd")
