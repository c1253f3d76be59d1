#!/usr/bin/env python3
"""Ad-hoc throughput probe of the InfoNCE kernels (pairs/s and fp32-MFMA TFLOP/s)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

g = torch.Generator(device="cuda").manual_seed(0)
for (m, n, d) in [(2048, 1_000_000, 64), (2048, 100_000, 64), (2048, 2048, 64), (8192, 8192, 64), (100_000, 100_000, 64),
                  (2048, 1_000_000, 128), (16384, 16384, 128)]:
    a = torch.randn(m, d, device="cuda", generator=g); b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    t = timeit(lambda: Fn.infonce_lse_raw(a, sa, b, sb, 5.0), 5 if m * n > 1e9 else 20)
    print(f"fwd M={m} N={n} d={d}: {t*1e3:.3f} ms  {m*n/t/1e9:.1f} Gpairs/s  {2*m*n*d/t/1e12:.1f} TFLOP/s", flush=True)
