#!/usr/bin/env python3
"""BCE softplus row sums 2^18 x 100K x 64: forward without a gradient (one tile product) vs with (two)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd import functional as Fn
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randn(1 << 18, 64, device="cuda", generator=g) * 0.1
b = torch.randn(100000, 64, device="cuda", generator=g) * 0.1
for grad in (False, True):
    aa = a.clone().requires_grad_(grad)
    for _ in range(2):
        Fn.bce_softplus_rowsum(aa, b)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        Fn.bce_softplus_rowsum(aa, b)
    torch.cuda.synchronize()
    print("with gradient (row sums + o)" if grad else "no gradient (row sums only)", f"{(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
# forward + both gradients (the item side through gcr_bce_bwd_f32)
aa = a.clone().requires_grad_(True)
bb = b.clone().requires_grad_(True)
w = torch.rand(1 << 18, device="cuda", generator=g)
for i in range(7):
    if i == 2:
        torch.cuda.synchronize()
        t = time.perf_counter()
    aa.grad = bb.grad = None
    torch.dot(Fn.bce_softplus_rowsum(aa, bb), w).backward()
torch.cuda.synchronize()
print(f"forward + both gradients {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
