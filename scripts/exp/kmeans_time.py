#!/usr/bin/env python3
"""run_kmeans at the e_step's shapes (1M / 100K points x 64, k = 300, faiss defaults; also k = 2000): ms per call and
objective, single stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from recommendation_amd import kmeans as KM
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
xu = torch.randn(1_000_000, 64, device=dev, generator=g) * 0.1
xi = torch.randn(100_000, 64, device=dev, generator=g) * 0.1
for x, k in ((xu, 300), (xi, 300), (xu, 2000)):
    for _ in range(3):
        KM.run_kmeans(x, k, assign_points=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        c = KM.run_kmeans(x, k, assign_points=False)[0]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    a = KM.assign_to_centroids(x, c)
    print(f"n={x.shape[0]} k={k}: {ms:.3f} ms per run_kmeans, objective {float(((x - c[a]) ** 2).sum()):.2f}", flush=True)
