#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: bench.py's `gcl_step` leg alone (the sharded SSL4Rec / GCL training step of
gcl.py:205-227 on its 2^18-user workload, single rank: 1 warm-up + 2 timed steps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_PORT", "29541")
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import distributed as gdist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist = bench._init_dist(dev)
out = bench.gcl_step_leg(ra, gdist, dist, dev, 0, 1, 64, None)
print({k: v for k, v in out.items() if k != "note"})
dist.destroy_process_group()
