#!/usr/bin/env python3
"""Accuracy of the InfoNCE engines at the NCL structure-contrast shape (2048 anchors x 1M table rows, d = 64,
1/tau = 10) against a float64 evaluation on the GPU (torch.float64 matmul, 256 anchors at a time): row lse, the
softmax-weighted row sum o, and the table-side gradient, per engine setting."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(3)
m, n, d, inv_tau = 2048, 1_000_000, 64, 10.0
table = torch.randn(n, d, device=dev, generator=g) * 0.1
idx = torch.randint(0, n, (m,), device=dev, generator=g)
anchors = table[idx] + 0.05 * torch.randn(m, d, device=dev, generator=g)       # every anchor has a near-duplicate row
w = torch.rand(m, device=dev, generator=g) + 0.5
sa, sb = Fn.row_inv_norm(anchors), Fn.row_inv_norm(table)

an = torch.nn.functional.normalize(anchors.double(), dim=1)
bn = torch.nn.functional.normalize(table.double(), dim=1)
lse64 = torch.empty(m, dtype=torch.float64, device=dev)
o64 = torch.empty(m, d, dtype=torch.float64, device=dev)
gb64 = torch.zeros(n, d, dtype=torch.float64, device=dev)
for i0 in range(0, m, 256):
    s = inv_tau * an[i0:i0 + 256] @ bn.T
    l = torch.logsumexp(s, dim=1)
    p = torch.exp(s - l[:, None])
    lse64[i0:i0 + 256] = l
    o64[i0:i0 + 256] = p @ bn
    gb64 += inv_tau * (p * w[i0:i0 + 256, None].double()).T @ an[i0:i0 + 256]
    del s, p


def rel(x, ref):
    return float((x.double() - ref).abs().max() / ref.abs().max())


print(f"shape {m} x {n} x {d}, 1/tau = {inv_tau}; errors are max |x - f64| / max |f64|")
for name, flag in (("two f16 planes (EngH2)", Fn.INFONCE_UNIT_ROWS), ("three bf16 planes (EngB3)", 0),
                   ("f32 MFMA", Fn.INFONCE_ENGINE_F32)):
    lse = Fn.infonce_lse_raw(anchors, sa, table, sb, inv_tau, engine_flag=flag)
    line = f"{name:28s} lse {rel(lse, lse64):.2e} (abs {float((lse.double() - lse64).abs().max()):.2e})"
    if Fn.infonce_fwd_o_supported(d, flag):
        lse_o, o = Fn.infonce_fwd_o_raw(anchors, sa, table, sb, inv_tau, engine_flag=flag)
        line += f"  flash lse {rel(lse_o, lse64):.2e}  o {rel(o, o64):.2e}"
    gb = Fn._infonce_bwd_raw(table, sb, anchors, sa, inv_tau, None, None, lse, w, engine_flag=flag)
    line += f"  table gradient {rel(gb, gb64):.2e}"
    print(line, flush=True)
