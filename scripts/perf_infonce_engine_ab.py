#!/usr/bin/env python3
"""In-process interleaved A/B of the two InfoNCE engines (GCR_INFONCE_ENGINE = f32 | b3): time, and
the error of each against a float64 torch reference of the row logsumexp on a sample of anchors."""
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
shapes = [(2048, 1_000_000, 64), (100_000, 100_000, 64), (2048, 100_000, 32)]
for (m, n, d) in shapes:
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    for inv_tau in (5.0, 20.0):
        fn = lambda: Fn.infonce_lse_raw(a, sa, b, sb, inv_tau, unit_rows=True)
        # float64 reference on 64 anchors
        sel = torch.arange(0, m, max(1, m // 64), device="cuda")[:64]
        an = (a[sel].double() * sa[sel].double()[:, None])
        ref = torch.zeros(sel.numel(), dtype=torch.float64, device="cuda")
        chunks = []
        for j0 in range(0, n, 250_000):
            bn = b[j0:j0 + 250_000].double() * sb[j0:j0 + 250_000].double()[:, None]
            chunks.append(torch.logsumexp(an @ bn.T * inv_tau, 1))
        ref = torch.logsumexp(torch.stack(chunks, 1), 1)
        variants = ("b3/0", "b3/1", "b3/0/16")
        res, err = {v: [] for v in variants}, {}
        for rnd in range(5):
            for v in variants:
                os.environ["GCR_INFONCE_ENGINE"] = v.split("/")[0]
                os.environ["GCR_INFONCE_PIPE"] = v.split("/")[1]
                os.environ["GCR_INFONCE_MFMA"] = v.split("/")[2] if v.count("/") == 2 else "32"
                if rnd == 0:
                    out = fn()
                    torch.cuda.synchronize()
                    err[v] = float((out[sel].double() - ref).abs().max())
                res[v].append(once(fn, 3))
        for v in variants:
            med = statistics.median(res[v])
            print(f"M={m} N={n} d={d} inv_tau={inv_tau} engine={v}: median {med:.3f} ms  {2*m*n*d/med/1e9:.1f} TF(alg)  "
                  f"max|lse-ref64|={err[v]:.2e}", flush=True)
