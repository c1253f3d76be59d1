#!/usr/bin/env python3
"""ncl_train_step_full (graph replay, k = 300) at cfg3 scale under the k-means search variants: tiled (IMAGE_SEARCH off),
image search at 234 registers, image search at <= 128 registers (co-resident with the InfoNCE loops)."""
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn, kmeans as K  # noqa: E402
from recommendation_amd.ncl import NCLModel  # noqa: E402
from recommendation_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
x0 = torch.empty(n_u + n_i, 64, device=dev)
torch.nn.init.xavier_uniform_(x0, generator=torch.Generator(device=dev).manual_seed(0))
gen = torch.Generator(device=dev).manual_seed(1)
bsz = 2048
uidx = torch.randint(0, n_u, (bsz,), device=dev, generator=gen)
iidx = torch.randint(0, n_i, (bsz,), device=dev, generator=gen)
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
jn = Fn.neg_sample(rowptr_u, items_u, uidx, 1, n_i, 3, 0, 101)
batch = (uidx, iidx, jn)


ISSUE = "late"
PRIO = 0


def leg(k, capture=True, e_step=True, reps=10):
    conf = copy.deepcopy(bench.NCL_CFG3)
    conf["NCL"]["num_clusters"] = k
    m = NCLModel.from_graph(conf, graph, n_u, n_i, graph_capture=capture)
    m.e_stream_priority = PRIO
    with torch.no_grad():
        m.model.table.copy_(x0)
    opt = FusedAdam(m.model.parameters(), lr=1e-3, capturable=capture)
    m.e_step()
    if not e_step:
        m.train_step(batch, opt, check_negatives=False, fused=True)
        m._fused.e_step_every_batch = False
    m.train_step(batch, opt, check_negatives=False, fused=True)
    m._fused.e_step_issue = ISSUE
    for _ in range(3):
        m.train_step(batch, opt, check_negatives=False, fused=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        m.train_step(batch, opt, check_negatives=False, fused=True)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps


print(f"no e_step: {leg(300, e_step=False):.3f} ms", flush=True)
for prio in (0, -1):
    PRIO = prio
    for name, img, low in (("tiled", False, False), ("image, <= 128 registers", True, True)):
        K.IMAGE_SEARCH, K.IMAGE_SEARCH_LOW_REGISTERS = img, low
        print(f"e_step stream priority {prio}, {name}: step {leg(300):.3f} ms (graph replay), eager {leg(300, capture=False, reps=5):.3f} ms",
              flush=True)
