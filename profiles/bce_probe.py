#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: the `loss_type == "bce"` branch of lightgcn.py (lightgcn.py:109-113) —
the all-pairs softplus row sums forward + both gradients at 2^18 user rows x 100K items x 64 (three rounds), and three
whole training steps at the cfg1 size (943 x 1682 / 80 000 edges) through LightGCN.loss."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from recommendation_amd import functional as Fn  # noqa: E402
from recommendation_amd.encoders import LightGCN  # noqa: E402
from recommendation_amd.optim import FusedAdam  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
a = (torch.randn(1 << 18, 64, device=dev, generator=g) * 0.1).requires_grad_(True)
b = (torch.randn(100_000, 64, device=dev, generator=g) * 0.1).requires_grad_(True)
for _ in range(3):
    a.grad = b.grad = None
    Fn.bce_softplus_rowsum(a, b).sum().backward()
w1 = bench.WORKLOADS["cfg1"]
u1, i1 = bench.synth_interactions_device(w1["users"], w1["items"], w1["edges"], bench.SEED, dev)
ei = torch.stack([torch.cat([u1, i1 + w1["users"]]), torch.cat([i1 + w1["users"], u1])])
model = LightGCN(w1["users"], w1["items"], 64, 2).to(dev)
graph = model.prepare(ei)
opt = FusedAdam(model.parameters(), lr=1e-3)
for _ in range(3):
    opt.zero_grad()
    model.loss(graph, loss_type="bce", reg_weight=1e-4).backward()
    opt.step()
torch.cuda.synchronize()
print("bce probe done: pairs per all-pairs call", (1 << 18) * 100_000)
