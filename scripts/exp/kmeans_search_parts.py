#!/usr/bin/env python3
"""Where a Lloyd iteration's search launch spends its time at the e_step shape (76.8 K x 64 points, k = 300): the search
alone (assignment out, no accumulation) and with the fused row atomics, tiled kernel vs image kernel, n_copies swept."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")
g = torch.Generator(device="cuda").manual_seed(0)
n, k, d = 76800, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 64
x = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g) + 0.5, dim=1)
cent = x[torch.randperm(n, device=dev, generator=g)[:k]].clone()
half = 0.5 * (cent * cent).sum(1)
img = torch.empty(int(L.gcr_kmeans_image_bytes(k, d)), dtype=torch.uint8, device=dev)
st = _lib.cur_stream(dev)
_lib.check(L.gcr_kmeans_centroid_image_f32(_lib.dptr(cent), _lib.dptr(half), k, d, _lib.dptr(img), st), "img")
assign = torch.empty(n, dtype=torch.int64, device=dev)
assign2 = torch.empty(n, dtype=torch.int64, device=dev)


def ms(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


t_old = ms(lambda: _lib.check(L.gcr_kmeans_assign_f32(_lib.dptr(x), n, _lib.dptr(cent), _lib.dptr(half), k, d, _lib.dptr(assign), None, st), "a"))
t_new = ms(lambda: _lib.check(L.gcr_kmeans_search_image_f32(_lib.dptr(x), n, _lib.dptr(img), k, d, _lib.dptr(assign2), None, None, 1, 0, st), "b"))
t_low = ms(lambda: _lib.check(L.gcr_kmeans_search_image_f32(_lib.dptr(x), n, _lib.dptr(img), k, d, _lib.dptr(assign2), None, None, 1, 1, st), "b"))
print(f"low-register image search only: {t_low:.1f} us, same assignment: {bool(torch.equal(assign, assign2))}")
for copies in (1, 4, 16, 64):
    sums = torch.zeros(copies, k, d, device=dev)
    counts = torch.zeros(copies, k, device=dev)
    t_o = ms(lambda: _lib.check(L.gcr_kmeans_assign_accumulate_f32(_lib.dptr(x), n, _lib.dptr(cent), _lib.dptr(half), k, d, None,
                                                                   _lib.dptr(sums), _lib.dptr(counts), copies, st), "c"))
    t_n = ms(lambda: _lib.check(L.gcr_kmeans_search_image_f32(_lib.dptr(x), n, _lib.dptr(img), k, d, None, _lib.dptr(sums),
                                                              _lib.dptr(counts), copies, 0, st), "d"))
    t_l = ms(lambda: _lib.check(L.gcr_kmeans_search_image_f32(_lib.dptr(x), n, _lib.dptr(img), k, d, None, _lib.dptr(sums),
                                                              _lib.dptr(counts), copies, 1, st), "d"))
    print(f"search + row atomics, {copies:2d} copies: tiled {t_o:.1f} us, image {t_n:.1f} us, low-register image {t_l:.1f} us")
