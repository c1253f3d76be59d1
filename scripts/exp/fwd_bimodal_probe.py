#!/usr/bin/env python3
"""Why does the lse-only InfoNCE forward (2048 x 1M x 64) measure 0.80-0.82 ms in some runs and 0.86-0.89 in others with
one binary?  Time it in ONE process under different conditions: repeated, after re-allocating the operands at shifted
addresses, after a heavy unrelated kernel (clock / power state), with more repetitions."""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd import functional as Fn

g = torch.Generator(device="cuda").manual_seed(0)
m, n, d = 2048, 1_000_000, 64
ef = Fn._resolve_engine(unit_rows=True)


def ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


keep = []
for trial in range(8):
    pad = torch.empty((trial * 37 + 1) * 4096 + trial * 64, device="cuda")      # shifts the following allocations
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    f = lambda: Fn.infonce_lse_raw(a, sa, b, sb, 10.0, engine_flag=ef)
    t = [ms(f) for _ in range(4)]
    print(f"trial {trial}: b.data_ptr % 2MiB = {b.data_ptr() % (1 << 21):8d}  a % 4096 = {a.data_ptr() % 4096:5d}  "
          f"fwd ms {' '.join('%.3f' % x for x in t)}", flush=True)
    keep.append(pad)
    del a, b, sa, sb
# clock state: idle 0.5 s, then time; heavy matmul burst, then time
a = torch.randn(m, d, device="cuda", generator=g); b = torch.randn(n, d, device="cuda", generator=g)
sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
f = lambda: Fn.infonce_lse_raw(a, sa, b, sb, 10.0, engine_flag=ef)
time.sleep(0.5)
print("after idle:", " ".join("%.3f" % ms(f, 3) for _ in range(4)))
x = torch.randn(8192, 8192, device="cuda")
for _ in range(20):
    x @ x
torch.cuda.synchronize()
print("after matmul burst:", " ".join("%.3f" % ms(f, 3) for _ in range(4)))
print("long:", "%.3f" % ms(f, 100))
