#!/usr/bin/env python3
"""A/B of FusedNCLStep's stream overlaps at cfg3 scale (bench.py's NCL workload): overlap_forward / overlap_backward /
early_e_step on and off, eager and hipGraph replay, with and without the per-batch e_step."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import recommendation_amd as ra
from recommendation_amd import functional as Fn
from recommendation_amd.ncl import NCLModel
from recommendation_amd.optim import FusedAdam

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
users, items = bench.synth_interactions_device(wl["users"], wl["items"], wl["edges"], bench.SEED, dev)
n_u, n_i = wl["users"], wl["items"]
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
gen = torch.Generator(device=dev).manual_seed(1)
bsz = 2048
uidx = torch.randint(0, n_u, (bsz,), device=dev, generator=gen)
iidx = torch.randint(0, n_i, (bsz,), device=dev, generator=gen)
rowptr_u = graph.rowptr[: n_u + 1].contiguous()
items_u = (graph.col[: int(rowptr_u[-1])] - n_u).contiguous()
jn = Fn.neg_sample(rowptr_u, items_u, uidx, 1, n_i, 3, 0, 101)
batch = (uidx, iidx, jn)


def leg(capture, e_step, **flags):
    conf = copy.deepcopy(bench.NCL_CFG3)
    conf["NCL"]["num_clusters"] = 300
    m = NCLModel.from_graph(conf, graph, n_u, n_i, graph_capture=capture)
    opt = FusedAdam(m.model.parameters(), lr=1e-3, capturable=capture)
    m.e_step()
    from recommendation_amd.ncl_step import FusedNCLStep
    m._fused = FusedNCLStep(m, opt)                       # flags first: a capture freezes the launch sequence
    for k, v in flags.items():
        setattr(m._fused, k, v)
    m._fused.e_step_every_batch = e_step
    if capture:
        m._fused.capture(bsz)
    for _ in range(4):
        m.train_step(batch, opt, check_negatives=False, fused=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            m.train_step(batch, opt, check_negatives=False, fused=True)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
    del m, opt
    torch.cuda.empty_cache()
    return best


for e_step in (True, False):
    for capture in (True, False):
        for of in (False, True):
            for ob in (False, True):
                t = leg(capture, e_step, overlap_forward=of, overlap_backward=ob)
                print(f"e_step={e_step} graph={capture} overlap_forward={of} overlap_backward={ob}: {t:.3f} ms", flush=True)
