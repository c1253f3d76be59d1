#!/usr/bin/env python3
"""Workload for rocprofv3 --kernel-trace --stats: the e_step's k-means (1M x 64 -> 76.8 K sampled points, k = 300, 25
iterations), the tiled search with float row atomics, then the image search with incremental fixed-point sums, eager."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import kmeans as K  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.nn.functional.normalize(torch.randn(1_000_000, 64, device="cuda", generator=g) + 0.5, dim=1)
for img, incr in ((False, False), (True, True)):
    K.IMAGE_SEARCH, K.INCREMENTAL_UPDATE = img, incr
    for _ in range(4):
        K.run_kmeans(x, 300, assign_points=False)
torch.cuda.synchronize()
print("kmeans e_step probe done")
