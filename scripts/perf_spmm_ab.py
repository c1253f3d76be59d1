#!/usr/bin/env python3
"""Interleaved A/B of the SpMM partition -> XCD mapping knob (GCR_SPMM_XCD) in one process on the bench graphs."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[name]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
rp, c, v = bench.sym_norm_csr_device(users, items, wl["users"], wl["items"])
n = wl["users"] + wl["items"]
d = 64
x = torch.randn(n, d, device=dev)
y = torch.empty_like(x)
variants = [512]
unrs = ["0", "1"]
graphs = {L: ra.CsrGraph(rp, c, v, n, n, dev, symmetric=True, nnz_per_part=L, validate=False) for L in variants}
res = {L: [] for L in unrs}
for rnd in range(7):
    for L in unrs:
        os.environ["GCR_SPMM_XCD"] = L
        g = graphs[512]
        if rnd == 0:
            Fn.spmm_into(g, x, y=y)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            Fn.spmm_into(g, x, y=y)
        e1.record()
        torch.cuda.synchronize()
        res[L].append(e0.elapsed_time(e1) / 5)
nnz = graphs[512].nnz
bytes_alg = nnz * 264 + n * 260
print(name, "nnz", nnz)
for L in unrs:
    med = statistics.median(res[L])
    print(f"  XCD-chunked {L:>3s}: median {med:.4f} ms  min {min(res[L]):.4f} ms  {bytes_alg / med / 1e6:.0f} GB/s alg")
