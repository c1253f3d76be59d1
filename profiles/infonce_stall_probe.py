#!/usr/bin/env python3
"""Workload for the stall-counter passes on the two-product InfoNCE loop (rocprofv3 --pmc ...): flash forward and
table-side backward at 2048 x 1M x 64, three launches each, nothing else."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
m, n, d = 2048, 1_000_000, 64
a = torch.randn(m, d, device="cuda", generator=g)
b = torch.randn(n, d, device="cuda", generator=g)
sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
ef = Fn._resolve_engine(unit_rows=True) if "--b3" not in sys.argv else 0
w = torch.ones(m, device="cuda")
for _ in range(3):
    lse, o = Fn.infonce_fwd_o_raw(a, sa, b, sb, 10.0, engine_flag=ef)
    Fn._infonce_bwd_raw(b, sb, a, sa, 10.0, None, None, lse, w, engine_flag=ef)
    Fn.infonce_lse_raw(a, sa, b, sb, 10.0, engine_flag=ef)
torch.cuda.synchronize()
print("stall probe done")
