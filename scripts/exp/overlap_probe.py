#!/usr/bin/env python3
"""Does an HBM-bound SpMM layer hide under an MFMA-bound InfoNCE launch when both are in flight on two HIP streams?
cfg2 graph (1 layer = 20M edges) next to the 2048 x 1M x 64 flash forward.  Prints serial and concurrent times."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import recommendation_amd as ra
from recommendation_amd import functional as Fn

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
graph = ra.CsrGraph.bipartite_sym_norm(users, items, wl["users"], wl["items"], dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(wl["users"] + wl["items"], 64, device=dev, generator=g)
a = torch.randn(2048, 64, device=dev, generator=g)
b = x[: wl["users"]]
sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def nce():
    return Fn.infonce_fwd_o_raw(a, sa, b, sb, 10.0)


def layer():
    return Fn.spmm(graph, x)


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out)


def both(n_layers):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        nce()
    with torch.cuda.stream(s2):
        for _ in range(n_layers):
            layer()
    cur.wait_stream(s1); cur.wait_stream(s2)


with torch.no_grad():
    t_n, t_l = timed(nce), timed(layer)
    print(f"serial: flash fwd {t_n:.3f} ms, one SpMM layer {t_l:.3f} ms")
    for k in (1, 2, 3):
        t = timed(lambda: both(k))
        print(f"concurrent flash fwd + {k} layer(s): {t:.3f} ms (serial sum {t_n + k * t_l:.3f})", flush=True)
