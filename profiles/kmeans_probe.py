#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: NCL's e_step shape, 1M x 64 points, k = 2000, 20 Lloyd iterations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.kmeans import run_kmeans  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(1_000_000, 64, device="cuda", generator=g)
for _ in range(2):
    run_kmeans(x, 2000, niter=20, seed=1)
torch.cuda.synchronize()
print("kmeans probe done")
