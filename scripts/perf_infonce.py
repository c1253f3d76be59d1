#!/usr/bin/env python3
"""Bare-kernel timings of the InfoNCE tile engines at the NCL structure-contrast shape (2048 x 1M x 64) and the
symmetric 100K x 100K shape: forward (lse), flash forward (lse + weighted row sum), table-side backward; HIP events
on the launch stream, median of 5 rounds."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import _lib
if "--lib" in sys.argv:            # A/B of two builds of the library on one box: --lib recommendation_amd/libgcr_x.so
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from recommendation_amd import functional as Fn


def ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out)


g = torch.Generator(device="cuda").manual_seed(0)
shapes = [(2048, 1_000_000, 64)] if "--quick" in sys.argv else \
    [(2048, 1_000_000, 64), (100_000, 100_000, 64), (2048, 100_000, 128), (2048, 1_000_000, 32)]
for (m, n, d) in shapes:
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    lse = Fn.infonce_lse_raw(a, sa, b, sb, 10.0)
    w = torch.ones(m, device="cuda")
    ef = Fn._resolve_engine(unit_rows=True) if "--b3" not in sys.argv else 0     # rows normalised by sa / sb
    t_f = ms(lambda: Fn.infonce_lse_raw(a, sa, b, sb, 10.0, engine_flag=ef))
    t_o = ms(lambda: Fn.infonce_fwd_o_raw(a, sa, b, sb, 10.0, engine_flag=ef))
    t_b = ms(lambda: Fn._infonce_bwd_raw(b, sb, a, sa, 10.0, None, None, lse, w, engine_flag=ef))
    fl = 2.0 * m * n * d / 1e9
    if "--sides" in sys.argv and m == n:       # the other statistics layouts of the backward loop (symmetric / self-similarity)
        col = Fn.infonce_lse_raw(b, sb, a, sa, 10.0)
        wn = torch.ones(n, device="cuda")
        t0 = ms(lambda: Fn._infonce_bwd_raw(a, sa, b, sb, 10.0, lse, w, col, wn, engine_flag=ef))
        t1 = ms(lambda: Fn._infonce_bwd_raw(a, sa, b, sb, 10.0, lse, w, None, None, engine_flag=ef))
        tx = ms(lambda: Fn._infonce_bwd_raw(a, sa, a, sa, 10.0, lse, w, lse, w, exclude_diagonal=True, engine_flag=ef))
        print(f"M={m} N={n} d={d}: bwd both sides {t0:.3f} ms  stationary side only {t1:.3f} ms  "
              f"self-similarity, excluded diagonal {tx:.3f} ms", flush=True)
    print(f"M={m} N={n} d={d}: fwd {t_f:.3f} ms ({fl / t_f:.0f} TF)  fwd_o {t_o:.3f} ms ({2 * fl / t_o:.0f} TF)  "
          f"bwd(table) {t_b:.3f} ms ({2 * fl / t_b:.0f} TF)", flush=True)
