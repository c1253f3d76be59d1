#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md §HBM prescribes).  Dispatch order of spmm_parts:
  1..3   calibration: diagonal graph, N = 4M rows, d = 64 -> every byte is known
         (read x once = N*256 B + col/val/rowptr/desc, write y once = N*256 B)
  4..    the bench's own timed step (bench.py: `lightgcn_propagate(graph, x0, K, combine="sum")`, the
         Horner form), 2 forward passes x K layers, on the graph bench.py builds (library ingest)
Usage (the program goes directly after `--`):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof/pmc_fetch_cfg2 -- \
      python3 profiles/pmc_probe.py --workload cfg2
Writes gpurun_out/pmc_probe_<workload>.json (sizes + the digest of the kernel sources it ran).
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cfg2"

n_cal = 4_000_000
rowptr = torch.arange(n_cal + 1, dtype=torch.int64)
col = torch.arange(n_cal, dtype=torch.int32)
val = torch.ones(n_cal)
gcal = ra.CsrGraph(rowptr, col, val, n_cal, n_cal, dev, symmetric=True)
xc = torch.randn(n_cal, 64, device=dev)
yc = torch.empty_like(xc)
for _ in range(3):
    Fn.spmm_into(gcal, xc, y=yc)
torch.cuda.synchronize()
cal_parts = gcal.plan.n_parts
del xc, yc, gcal

# `cfg2c` / `cfg4c`: the planted-community graph of bench.community_leg; `--reorder` applies the spectral renumbering +
# XCD-grouped plan first (its own SpMM launches come BEFORE the measured ones; the summariser takes the last 2 x K)
community = name.endswith("c")
wl = bench.WORKLOADS[name[:-1] if community else name]
if community:
    n_comm = max(2, wl["users"] // 8192)
    users, items, _, _ = bench.synth_community_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev,
                                                                   n_comm, bench.COMMUNITY["p_in"])
    graph = ra.CsrGraph.bipartite_sym_norm(users, items, wl["users"], wl["items"], dev)
    if "--reorder" in sys.argv:
        from recommendation_amd import reorder as R
        pu, pi, group = R.locality_permutation(users, items, wl["users"], wl["items"], dev, graph=graph)
        graph = ra.CsrGraph.bipartite_sym_norm(pu[users], pi[items], wl["users"], wl["items"], dev, row_group=group)
    name = name + ("_after" if "--reorder" in sys.argv else "_before")
else:
    users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
    graph = ra.CsrGraph.bipartite_sym_norm(users, items, wl["users"], wl["items"], dev)
del users, items
n = wl["users"] + wl["items"]
x0 = torch.empty(n, 64, device=dev)
torch.nn.init.xavier_uniform_(x0, generator=torch.Generator(device=dev).manual_seed(0))
with torch.no_grad():
    for _ in range(2):
        Fn.lightgcn_propagate(graph, x0, wl["layers"], combine="sum")
torch.cuda.synchronize()
info = {"workload": name, "nnz": graph.nnz, "n": n, "layers": wl["layers"], "d": 64, "cal_rows": n_cal,
        "cal_parts": cal_parts, "parts": graph.plan.n_parts, "long_rows": graph.plan.n_long,
        "source_digest": bench.spmm_source_digest()}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"pmc_probe_{name}.json"), "w") as f:
    json.dump(info, f)
print("probe done", json.dumps(info))
