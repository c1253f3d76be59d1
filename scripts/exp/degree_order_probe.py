#!/usr/bin/env python3
"""Does a degree-sorted numbering alone (no communities to find) help the SpMM on the UNIFORM benchmark graphs?
Hot (popular) item rows become contiguous.  usage: degree_order_probe.py cfg2|cfg4"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import recommendation_amd as ra
from recommendation_amd import functional as Fn, reorder as R

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
wl = bench.WORKLOADS[name]
dev = torch.device("cuda", 0)
n_u, n_i = wl["users"], wl["items"]
users, items = bench.synth_interactions_device(n_u, n_i, wl["edges"], bench.SEED, dev)
x = torch.randn(n_u + n_i, 64, device=dev)


def layer_ms(g, reps=20):
    y = torch.empty_like(x)
    for _ in range(3):
        Fn.spmm_into(g, x, acc_in=x, acc_out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Fn.spmm_into(g, x, acc_in=x, acc_out=y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


g0 = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
print(name, "as generated:        %.3f ms / layer" % layer_ms(g0))
deg = (g0.rowptr[1:] - g0.rowptr[:-1]).float()
pu, pi, _ = R.order_from_labels(torch.zeros(n_u + n_i, dtype=torch.int64, device=dev), deg, n_u)
del g0
g1 = ra.CsrGraph.bipartite_sym_norm(users, pi[items], n_u, n_i, dev)
print(name, "items by degree:     %.3f ms / layer" % layer_ms(g1))
del g1
g2 = ra.CsrGraph.bipartite_sym_norm(pu[users], pi[items], n_u, n_i, dev)
print(name, "both by degree:      %.3f ms / layer" % layer_ms(g2))
