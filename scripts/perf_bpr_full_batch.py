#!/usr/bin/env python3
"""lightgcn.py's full-batch training step at cfg2 scale (B = E = 10M triples, lightgcn.py:91-118): forward
propagation, BPR forward, BPR backward (row-gradient scatter), backward propagation, timed separately."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import recommendation_amd as ra
from recommendation_amd import functional as Fn

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["cfg2"]
n_u, n_i = wl["users"], wl["items"]
users, items = bench.synth_interactions_device(n_u, n_i, wl["edges"], bench.SEED, dev)
graph = ra.CsrGraph.bipartite_sym_norm(users, items, n_u, n_i, dev)
x = torch.nn.Parameter(torch.empty(n_u + n_i, 64, device=dev))
torch.nn.init.xavier_uniform_(x)
neg = torch.randint(0, n_i, (users.numel(),), device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t) / reps


def step(order_u, order_i, order_j):
    x.grad = None
    final = Fn.lightgcn_propagate(graph, x, 3, "sum")
    ue, ie = Fn.split_rows(final, n_u)
    s = Fn.bpr_sums(ue, ie, order_u, order_i, order_j, Fn.BPR_LOG_SIGMOID)
    (s[0] / order_u.numel() + 1e-4 * (s[1] + s[2] + s[3])).backward()


with torch.no_grad():
    final = Fn.lightgcn_propagate(graph, x, 3, "sum")
    print("propagate fwd      %.2f ms" % timeit(lambda: Fn.lightgcn_propagate(graph, x, 3, "sum")))
    print("bpr fwd            %.2f ms" % timeit(lambda: Fn.bpr_sums(final[:n_u], final[n_u:], users, items, neg, Fn.BPR_LOG_SIGMOID)))
o = torch.argsort(users)
us, its, ns = users[o].contiguous(), items[o].contiguous(), neg[o].contiguous()
if os.environ.get("ONLY") != "sorted":
    print("whole step (edges in generation order)   %.2f ms" % timeit(lambda: step(users, items, neg)))
if os.environ.get("ONLY") != "random":
    print("whole step (edges sorted by user)         %.2f ms" % timeit(lambda: step(us, its, ns)))
