import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.kmeans import run_kmeans
g = torch.Generator(device="cuda").manual_seed(0)
for n, k in ((100_000, 2000), (20_000, 500), (1_000_000, 2000)):
    x = torch.randn(n, 64, device="cuda", generator=g)
    run_kmeans(x, k); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): run_kmeans(x, k)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): run_kmeans(x, k)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} k={k}: wall {1e3*wall:.2f} ms per k-means, GPU span {e0.elapsed_time(e1)/5:.2f} ms")
