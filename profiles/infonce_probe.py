#!/usr/bin/env python3
"""Workload for the rocprofv3 passes on the InfoNCE kernels: NCL structure-contrast shape
(ncl.py:358-367) 2048 anchors x 1M table rows, d = 64; 5 forward + 3 forward/backward rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
m, n, d = 2048, 1_000_000, 64
a = torch.randn(m, d, device="cuda", generator=g).requires_grad_(True)
b = torch.randn(n, d, device="cuda", generator=g).requires_grad_(True)
pos = torch.randint(0, n, (m,), device="cuda", generator=g)
for _ in range(5):
    with torch.no_grad():
        Fn.infonce_stats(a, b, pos, 0.2)
for _ in range(3):
    a.grad = b.grad = None
    lse, pl = Fn.infonce_stats(a, b, pos, 0.2)
    (lse - pl).sum().backward()
torch.cuda.synchronize()
import json  # noqa: E402
import bench  # noqa: E402
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
with open(os.path.join(root, "gpurun_out", "pmc_probe_infonce.json"), "w") as f:
    json.dump({"shape": f"{m} x {n} x {d}", "source_digest": bench.infonce_source_digest()}, f)
print("infonce probe done: pairs per call", m * n, "flop per fwd call", 2 * m * n * d)
