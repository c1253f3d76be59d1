#!/usr/bin/env python3
"""cfg4-size structure contrast (ncl.py:358-367 at 10M users): 2048 anchors x 10M table rows, d = 64, forward
and both gradients on the default engine, checked against float64 on a sample (index arithmetic at the
largest shape of BASELINE.json)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn

g = torch.Generator(device="cuda").manual_seed(0)
m, n, d = 2048, 10_000_000, 64
a = torch.randn(m, d, device="cuda", generator=g).requires_grad_(True)
b = torch.randn(n, d, device="cuda", generator=g).requires_grad_(True)
pos = torch.randint(0, n, (m,), device="cuda", generator=g)
torch.cuda.synchronize(); t = time.perf_counter()
lse, pl = Fn.infonce_stats(a, b, pos, 0.1, normalize=True)
loss = (lse - pl).sum()
loss.backward()
torch.cuda.synchronize(); print("fwd+bwd %.1f ms" % (1e3 * (time.perf_counter() - t)))
sel = torch.arange(0, m, 64, device="cuda")
an = torch.nn.functional.normalize(a.detach()[sel].double(), dim=1)
ref = torch.full((sel.numel(),), -float("inf"), dtype=torch.float64, device="cuda")
ga = torch.zeros(sel.numel(), d, dtype=torch.float64, device="cuda")
chunks = []
for j0 in range(0, n, 1_000_000):
    bn = torch.nn.functional.normalize(b.detach()[j0:j0 + 1_000_000].double(), dim=1)
    chunks.append(torch.logsumexp(an @ bn.T * 10.0, 1))
ref = torch.logsumexp(torch.stack(chunks, 1), 1)
print("max |lse - ref64| on 32 anchors: %.2e" % float((lse.detach()[sel].double() - ref).abs().max()))
# gradient of the table rows at the tail of the table (largest row indices)
tail = torch.arange(n - 4096, n, device="cuda")
bn_t = torch.nn.functional.normalize(b.detach()[tail].double(), dim=1).requires_grad_(True)
an_all = torch.nn.functional.normalize(a.detach().double(), dim=1)
p = torch.exp(an_all @ bn_t.T * 10.0 - lse.detach().double()[:, None])          # softmax probabilities of the tail rows
g_hat = 10.0 * p.T @ an_all                                                     # d loss / d bhat (lse part only)
hit = (pos[None, :] == tail[:, None])
g_hat = g_hat - 10.0 * hit.double() @ an_all
bt = b.detach()[tail].double()
nrm = bt.norm(dim=1, keepdim=True)
bh = bt / nrm
want = (g_hat - bh * (bh * g_hat).sum(1, keepdim=True)) / nrm
got = b.grad[tail].double()
print("tail-row gradient: max abs err %.2e of max %.2e" % (float((got - want).abs().max()), float(want.abs().max())))
