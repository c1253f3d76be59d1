#!/usr/bin/env python3
"""In-process interleaved A/B: InfoNCE forward with the row mask applied on every tile vs only on
the ragged last tile (GCR_INFONCE_FORCE_MASK)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn

def once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

g = torch.Generator(device="cuda").manual_seed(0)
for (m, n, d) in [(2048, 1_000_000, 64), (100_000, 100_000, 64), (2048, 1_000_000, 128)]:
    a = torch.randn(m, d, device="cuda", generator=g); b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    fn = lambda: Fn.infonce_lse_raw(a, sa, b, sb, 5.0)
    res = {"0": [], "1": []}
    for rnd in range(7):
        for v in ("0", "1"):
            os.environ["GCR_INFONCE_FORCE_MASK"] = v
            if rnd == 0: fn(); torch.cuda.synchronize()
            res[v].append(once(fn, 3))
    for v in ("0", "1"):
        med = statistics.median(res[v])
        print(f"M={m} N={n} d={d} force_mask={v}: median {med:.3f} ms  {2*m*n*d/med/1e9:.1f} TF")
