#!/usr/bin/env python3
"""Accuracy of the BCE second products (o = sum_j sigmoid(s_ij) b_j; g_j = sum_i w_i sigmoid(s_ij) a_i) on two f16 planes vs
three bf16 planes against float64, in three score regimes: moderate (|s| ~ 1), converged one-hot BCE (sigmoid ~ 1e-5: s ~ -11.5)
and very negative (s ~ -20); weights spread over two orders of magnitude like degree / (E I)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd import functional as Fn

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
m, n, d = 1 << 17, 8192, 64
for name, shift in (("moderate", 0.0), ("converged (sigmoid ~ 1e-5)", 11.5), ("very negative (s ~ -20)", 20.0)):
    a = torch.randn(m, d, device=dev, generator=g) * 0.12
    b = torch.randn(n, d, device=dev, generator=g) * 0.12
    a[:, 0] = shift ** 0.5
    b[:, 0] = -(shift ** 0.5)
    w = torch.exp(torch.rand(m, device=dev, generator=g) * 4.6) * 1e-12          # 1 .. 100 x 1e-12
    s64 = a.double() @ b.double().T
    sg = torch.sigmoid(s64)
    o_ref = sg @ b.double()                                   # [m, d]
    g_ref = (sg * w.double().unsqueeze(1)).T @ a.double()     # [n, d]
    rows_ref = torch.nn.functional.softplus(s64).sum(1)
    del s64, sg
    for eng in ("auto", "b3"):
        fl = Fn._bce_flags(eng)
        rows, o = Fn.bce_fwd_raw(a, b, want_o=True, engine_flag=fl)
        gj = Fn.bce_bwd_raw(b, a, w_y=w, engine_flag=fl)
        e_r = float(((rows.double() - rows_ref).abs() / rows_ref.abs()).max())
        e_o = float((o.double() - o_ref).abs().max() / o_ref.abs().max())
        e_g = float((gj.double() - g_ref).abs().max() / g_ref.abs().max())
        e_or = float(((o.double() - o_ref).abs().amax(1) / o_ref.abs().amax(1)).max())     # worst row, relative to its own max
        e_gr = float(((gj.double() - g_ref).abs().amax(1) / g_ref.abs().amax(1)).max())
        print(f"{name:32s} {eng:5s} rowsum {e_r:.2e}  o {e_o:.2e} (worst row {e_or:.2e})  g {e_g:.2e} (worst row {e_gr:.2e})")
