#!/usr/bin/env python3
"""SpMM per-layer time on the planted-community graph (bench.synth_community_interactions_device, cfg2 sizes) with
(a) the shuffled ids it arrives with, (b) the spectral renumbering of recommendation_amd/reorder.py + XCD-grouped plan,
(c) the hidden community labels (what a perfect clustering would give), (d) the renumbering WITHOUT the grouped plan.
usage: perf_reorder.py [users items edges n_comm p_in]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import recommendation_amd as ra  # noqa: E402
from recommendation_amd import functional as Fn, reorder as R  # noqa: E402

a = [float(x) for x in sys.argv[1:] if not x.startswith("--")]
n_u, n_i, n_e = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (1_000_000, 100_000, 10_000_000)
n_comm = int(a[3]) if len(a) > 3 else 128
p_in = a[4] if len(a) > 4 else 0.85
dev = torch.device("cuda", 0)
users, items, cu, ci = bench.synth_community_interactions_device(n_u, n_i, n_e, bench.SEED, dev, n_comm, p_in)
n = n_u + n_i
x = torch.randn(n, 64, device=dev)


def layer_ms(graph, reps=20):
    y = torch.empty_like(x)
    for _ in range(3):
        Fn.spmm_into(graph, x, acc_in=x, acc_out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        Fn.spmm_into(graph, x, acc_in=x, acc_out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def build(pu, pi, group):
    u2, i2 = (users if pu is None else pu[users]), (items if pi is None else pi[items])
    n_uu = n_u
    g = ra.CsrGraph.bipartite_sym_norm(u2, i2, n_u, n_i, dev, row_group=group)
    return g


g0 = build(None, None, None)
print("shuffled ids:            %.3f ms / layer   parts %d" % (layer_ms(g0), g0.plan.n_parts))
torch.cuda.synchronize()
t = time.time()
rpc = int(a[5]) if len(a) > 5 else R.DEFAULT_ROWS_PER_CLUSTER
pu, pi, group = R.locality_permutation(users, items, n_u, n_i, dev, rows_per_cluster=rpc, graph=g0)
torch.cuda.synchronize()
t_re = time.time() - t
# clustering quality: fraction of users whose label's majority hidden community is their own
lab_u = group[:n_u][pu]            # label of old user id
best = torch.zeros(int(group.max()) + 1, n_comm, device=dev).index_put_((lab_u, cu), torch.ones(n_u, device=dev), accumulate=True)
purity = float(best.max(1).values.sum() / n_u)
print("spectral renumbering:    %.2f s, %d clusters, user purity %.3f" % (t_re, int(group.max()) + 1, purity))
g1 = build(pu, pi, group)
print("renumbered + grouped:    %.3f ms / layer   parts %d" % (layer_ms(g1), g1.plan.n_parts))
g2 = build(pu, pi, None)
print("renumbered, plain plan:  %.3f ms / layer" % layer_ms(g2))
lab = torch.cat([cu, ci])
deg = (g0.rowptr[1:] - g0.rowptr[:-1]).float()
pu3, pi3, group3 = R.order_from_labels(lab, deg, n_u)
g3 = build(pu3, pi3, group3)
print("hidden labels + grouped: %.3f ms / layer" % layer_ms(g3))
g4 = build(pu3, pi3, None)
print("hidden labels, plain:    %.3f ms / layer" % layer_ms(g4))

# --- robustness sweep: weaker structure and none at all, guard decisions included --------------
if "--sweep" in sys.argv:
    import json
    rows = []
    for tag, pin in (("uniform", None), ("p_in=0.5", 0.5), ("p_in=0.7", 0.7), ("p_in=0.85", 0.85)):
        if pin is None:
            uu, ii = bench.synth_interactions_device(n_u, n_i, n_e, bench.SEED, dev)
        else:
            uu, ii, _, _ = bench.synth_community_interactions_device(n_u, n_i, n_e, bench.SEED, dev, max(2, n_u // 8192), pin)
        gg = ra.CsrGraph.bipartite_sym_norm(uu, ii, n_u, n_i, dev)
        t_b = layer_ms(gg)
        ppu, ppi, grp = R.locality_permutation(uu, ii, n_u, n_i, dev, graph=gg)          # forced
        gf = ra.CsrGraph.bipartite_sym_norm(ppu[uu], ppi[ii], n_u, n_i, dev, row_group=grp)
        t_f = layer_ms(gf)
        del gf
        _, _, _, dec = R.guarded_locality_permutation(uu, ii, n_u, n_i, dev, graph=gg)
        rows.append({"graph": tag, "ms_per_layer_as_is": round(t_b, 4), "ms_per_layer_forced_renumbering": round(t_f, 4),
                     "guard": dec})
        print(json.dumps(rows[-1]))
        del gg
        torch.cuda.empty_cache()
