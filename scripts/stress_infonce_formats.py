#!/usr/bin/env python3
"""Randomised cross-check of the two operand formats of the split-operand InfoNCE kernels: every launch kind
(forward with / without column sums / excluded diagonal, flash forward, backward with statistics on either or both
sides, excluded diagonal) on random shapes, d in {32, 64, 128}, 1/tau in [1, 60] — two f16 planes vs three bf16 planes,
which must agree to a few f32 roundings.  usage: python scripts/stress_infonce_formats.py [n_cases [max 1/tau]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn  # noqa: E402

def run(n_cases=300, max_inv_tau=60.0, verbose=True):
    """Returns the worst relative difference per launch kind; raises AssertionError on a mismatch above 2e-5."""
    rng = np.random.default_rng(12345)
    dev = "cuda"
    H2, B3 = Fn.INFONCE_UNIT_ROWS, 0
    worst = {}


    def note(kind, x, y, tag):
        ref = float(y.abs().max())
        err = float((x - y).abs().max()) / max(ref, 1e-30)
        if not np.isfinite(err) or err > 2e-5:
            raise AssertionError(f"formats disagree: {kind} {tag} rel err {err}")
        worst[kind] = max(worst.get(kind, 0.0), err)


    for case in range(n_cases):
        d = int(rng.choice([32, 64, 128]))
        m = int(rng.choice([rng.integers(1, 200), rng.integers(1, 1500), 128 * rng.integers(1, 12)]))
        n = int(rng.choice([rng.integers(1, 300), rng.integers(1, 6000), 32 * rng.integers(1, 40)]))
        inv_tau = float(rng.choice([t for t in (1.0, 2.0, 5.0, 10.0, 20.0, 40.0, 60.0) if t <= max_inv_tau]))
        a = torch.from_numpy((rng.standard_normal((m, d)) * rng.uniform(0.05, 3.0)).astype(np.float32)).to(dev)
        b = torch.from_numpy((rng.standard_normal((n, d)) * rng.uniform(0.05, 3.0)).astype(np.float32)).to(dev)
        if n > 8 and m > 2:
            b[n // 3] = a[0] * 2.0                                   # a near-duplicate: a dominant logit
        sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
        tag = (case, m, n, d, inv_tau)
        w = torch.from_numpy(rng.standard_normal(m).astype(np.float32)).to(dev)
        v = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(dev)
        out = {}
        for eng in (H2, B3):
            lse = Fn.infonce_lse_raw(a, sa, b, sb, inv_tau, engine_flag=eng)
            lse2, col = Fn.infonce_lse_raw(a, sa, b, sb, inv_tau, col_bound=inv_tau * 1.0001, engine_flag=eng)
            flse, o = Fn.infonce_fwd_o_raw(a, sa, b, sb, inv_tau, engine_flag=eng)
            colx = Fn.infonce_lse_raw(b, sb, a, sa, inv_tau, engine_flag=eng)          # exact column lse for the both-sides launch
            res = dict(lse=lse, lse_colpass=lse2, col=col, flash_lse=flse, o=o,
                       bwd_y=Fn._infonce_bwd_raw(b, sb, a, sa, inv_tau, None, None, lse, w, engine_flag=eng),
                       bwd_x=Fn._infonce_bwd_raw(a, sa, b, sb, inv_tau, lse, w, None, None, engine_flag=eng),
                       bwd_both=Fn._infonce_bwd_raw(a, sa, b, sb, inv_tau, lse, w, colx, v, engine_flag=eng))
            if m == n or case % 3 == 0:
                k = min(m, n)
                aa, ss = a[:k].contiguous(), sa[:k].contiguous()
                xl = Fn.infonce_lse_raw(aa, ss, aa, ss, inv_tau, exclude_diagonal=True, engine_flag=eng)
                if k > 1:
                    xfl, xo = Fn.infonce_fwd_o_raw(aa, ss, aa, ss, inv_tau, exclude_diagonal=True, engine_flag=eng)
                    res.update(exd_lse=xl, exd_flash_lse=xfl, exd_o=xo,
                               exd_bwd_y=Fn._infonce_bwd_raw(aa, ss, aa, ss, inv_tau, None, None, xl, w[:k].contiguous(),
                                                             exclude_diagonal=True, engine_flag=eng),
                               exd_bwd_x=Fn._infonce_bwd_raw(aa, ss, aa, ss, inv_tau, xl, w[:k].contiguous(), None, None,
                                                             exclude_diagonal=True, engine_flag=eng))
            out[eng] = res
        for kind in out[B3]:
            note(kind, out[H2][kind], out[B3][kind], tag)
        if verbose and case % 50 == 49:
            print(f"{case + 1} cases, worst relative differences so far:", {k: f"{x:.1e}" for k, x in worst.items()}, flush=True)
    if verbose:
        print("OK", n_cases, "cases; worst relative differences:", {k: f"{x:.1e}" for k, x in sorted(worst.items())})
    return worst


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0)

