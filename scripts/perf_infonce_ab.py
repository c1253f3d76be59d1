#!/usr/bin/env python3
"""Interleaved A/B of the InfoNCE grid-size knob in ONE process (cdna_hip_programming.md rule 24):
variants x rounds, median and min per variant."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn  # noqa: E402


def once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device="cuda").manual_seed(0)
variants = [256, 512, 768, 1024, 1536, 2048, 3072]
for (m, n, d, mode) in [(2048, 1_000_000, 64, "fwd"), (2048, 1_000_000, 64, "bwd"), (100_000, 100_000, 64, "fwd"),
                        (8192, 8192, 64, "fwd"), (2048, 1_000_000, 128, "fwd")]:
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    lse = Fn.infonce_lse_raw(a, sa, b, sb, 5.0)
    w = torch.full((m,), 1.0 / m, device="cuda")
    if mode == "fwd":
        fn = lambda: Fn.infonce_lse_raw(a, sa, b, sb, 5.0)
        flop = 2.0 * m * n * d
    else:
        fn = lambda: Fn._infonce_bwd_raw(a, sa, b, sb, 5.0, lse, w, None, None)
        flop = 4.0 * m * n * d
    res = {v: [] for v in variants}
    for rnd in range(6):
        for v in variants:
            os.environ["GCR_INFONCE_BLOCKS"] = str(v)
            if rnd == 0:
                fn()
                torch.cuda.synchronize()
            res[v].append(once(fn, 3))
    print(f"{mode} M={m} N={n} d={d}")
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print(f"   blocks {v:5d}: median {med:8.3f} ms ({flop / med / 1e9:6.1f} TF)   min {mn:8.3f} ms ({flop / mn / 1e9:6.1f} TF)")
