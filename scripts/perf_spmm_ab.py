#!/usr/bin/env python3
"""Interleaved A/B of the issue order of the SpMM partitions (GCR_SPMM_INTERLEAVE = stride of a round-robin
shuffle of the descriptor list; GCR_SPMM_XCD=1 in the environment adds the XCD-chunked mapping) on the bench graphs."""
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
unrs = ["0", "2", "8", "64", "1024"]
graphs = {}
for L in unrs:
    os.environ["GCR_SPMM_INTERLEAVE"] = L
    graphs[L] = ra.CsrGraph(rp, c, v, n, n, dev, symmetric=True, nnz_per_part=512, validate=False)
os.environ["GCR_SPMM_INTERLEAVE"] = "0"
res = {L: [] for L in unrs}
for rnd in range(7):
    for L in unrs:
        g = graphs[L]
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
nnz = graphs[unrs[0]].nnz
bytes_alg = nnz * 264 + n * 260
print(name, "nnz", nnz)
for L in unrs:
    med = statistics.median(res[L])
    print(f"  interleave stride {L:>5s}: median {med:.4f} ms  min {min(res[L]):.4f} ms  {bytes_alg / med / 1e6:.0f} GB/s alg")
