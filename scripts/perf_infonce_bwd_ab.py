#!/usr/bin/env python3
"""In-process interleaved A/B of the InfoNCE backward engines (GCR_INFONCE_ENGINE = f32 | b3):
time of both input gradients of a row-softmax problem and their error against float64."""
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
for (m, n, d) in [(2048, 1_000_000, 64), (2048, 1_000_000, 128), (20_000, 20_000, 128), (2048, 100_000, 32)]:
    a = torch.randn(m, d, device="cuda", generator=g)
    b = torch.randn(n, d, device="cuda", generator=g)
    sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
    inv_tau = 10.0
    w = torch.full((m,), 1.0 / m, device="cuda")
    variants = ("f32", "b3/1", "h2/1")
    res = {v: {"ga": [], "gb": []} for v in variants}
    out = {}
    for rnd in range(5):
        for v in variants:
            os.environ["GCR_INFONCE_ENGINE"] = v.split("/")[0]
            os.environ["GCR_INFONCE_BWD_ILV"] = v.split("/")[-1]
            lse = Fn.infonce_lse_raw(a, sa, b, sb, inv_tau, unit_rows=True)
            f_a = lambda: Fn._infonce_bwd_raw(a, sa, b, sb, inv_tau, lse, w, None, None, False, True)
            f_b = lambda: Fn._infonce_bwd_raw(b, sb, a, sa, inv_tau, None, None, lse, w, False, True)
            if rnd == 0:
                out[v] = (lse, f_a(), f_b())
                torch.cuda.synchronize()
            res[v]["ga"].append(once(f_a, 2))
            res[v]["gb"].append(once(f_b, 2))
    # float64 reference on a sample of anchors (ga) and the matching partial check of gb
    sel = torch.arange(0, m, max(1, m // 32), device="cuda")[:32]
    an = a[sel].double() * sa[sel].double()[:, None]
    bn = b.double() * sb.double()[:, None]
    s64 = an @ bn.T * inv_tau
    p64 = torch.softmax(s64, 1) * (1.0 / m)
    ga64 = inv_tau * p64 @ bn
    for v in variants:
        lse, ga, gb = out[v]
        ea = float((ga[sel].double() - ga64).abs().max() / ga64.abs().max())
        el = float((lse[sel].double() - torch.logsumexp(s64, 1)).abs().max())
        ta, tb = statistics.median(res[v]["ga"]), statistics.median(res[v]["gb"])
        print(f"M={m} N={n} d={d} engine={v}: g_a {ta:.3f} ms ({4*m*n*d/ta/1e9:.0f} TF alg)  g_b {tb:.3f} ms "
              f"({4*m*n*d/tb/1e9:.0f} TF alg)  |lse err|={el:.1e}  ga rel-to-max err={ea:.1e}", flush=True)
    db = float((out["f32"][2] - out["h2/1"][2]).abs().max() / out["f32"][2].abs().max())
    d01 = float((out["b3/1"][2] - out["h2/1"][2]).abs().max())
    print(f"   b3 vs h2 g_b max abs diff {d01:.1e}", flush=True)
    print(f"   g_b engines differ by {db:.1e} of max", flush=True)
