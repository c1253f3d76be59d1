#!/usr/bin/env python3
"""Symmetric info_nce_loss forward (row + column LSE): one pass with column sums vs two passes, per engine."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn


def once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator(device="cuda").manual_seed(0)
for (m, d) in [(100_000, 64), (20_000, 64)]:
    a = torch.randn(m, d, device="cuda", generator=g)
    b = a + 0.3 * torch.randn(m, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    res = {}
    for rnd in range(5):
        for eng in ("f32", "b3", "auto"):
            Fn.INFONCE_ENGINE = eng
            ef = Fn._resolve_engine(unit_rows=True)        # rows normalised by sa / sb
            one = lambda: Fn.infonce_lse_raw(a, sa, b, sb, 5.0, col_bound=5.0, engine_flag=ef)
            two = lambda: (Fn.infonce_lse_raw(a, sa, b, sb, 5.0, engine_flag=ef), Fn.infonce_lse_raw(b, sb, a, sa, 5.0, engine_flag=ef))
            for name, fn in (("one-pass", one), ("two-pass", two)):
                if rnd == 0:
                    out = fn()
                    torch.cuda.synchronize()
                    if name == "one-pass":
                        ref = Fn.infonce_lse_raw(b, sb, a, sa, 5.0, engine_flag=ef)
                        print(f"  {eng} column lse one-pass vs swapped-role pass: max abs diff {float((out[1] - ref).abs().max()):.2e}")
                res.setdefault((eng, name), []).append(once(fn, 2))
    for k, v in res.items():
        print(f"M=N={m} d={d} {k[0]} {k[1]}: median {statistics.median(v):.3f} ms", flush=True)
