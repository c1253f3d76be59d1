#!/usr/bin/env python3
"""The NCL e_step's k-means at its own shape (ncl.py:340-356: 1M users / 100K items, d = 64, k = 300 -> 76.8 K sampled
training points, 25 Lloyd iterations), replayed from a hipGraph (no host launch gaps): tiled search (LDS staging, barrier
per centroid tile) vs the wave-independent search over the pre-split centroid image (kmeans.IMAGE_SEARCH).  Also the two
tables side by side on two streams, as NCLModel.e_step runs them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd import kmeans as K  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
xu = torch.nn.functional.normalize(torch.randn(1_000_000, 64, device="cuda", generator=g) + 0.5, dim=1)
xi = torch.nn.functional.normalize(torch.randn(100_000, 64, device="cuda", generator=g) + 0.5, dim=1)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def graph_ms(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        K.run_kmeans(xu, k, assign_points=False)
    with torch.cuda.stream(s2):
        K.run_kmeans(xi, k, assign_points=False)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


ref = None
for name, img, incr in (("tiled search, float row atomics", False, False), ("image search, float row atomics", True, False),
                        ("image search, incremental fixed-point sums", True, True)):
    K.IMAGE_SEARCH, K.INCREMENTAL_UPDATE = img, incr
    cu, _ = K.run_kmeans(xu, k, assign_points=False)
    cu2, _ = K.run_kmeans(xu, k, assign_points=False)
    same_run = bool(torch.equal(cu, cu2))
    if ref is None:
        ref = cu
    t_u = graph_ms(lambda: K.run_kmeans(xu, k, assign_points=False))
    t_i = graph_ms(lambda: K.run_kmeans(xi, k, assign_points=False))
    t_b = graph_ms(both)
    print(f"{name}: users {t_u:.3f} ms ({1e3 * t_u / 25:.1f} us / iteration)  items {t_i:.3f} ms  both tables on two streams "
          f"{t_b:.3f} ms; two runs bitwise equal: {same_run}; max |centroid - tiled run's| {float((ref - cu).abs().max()):.2e}", flush=True)
