"""Host-side logic added in round 3, on the CPU (no GPU, no libgcr compute call): the id-column encoding of the device id
maps, the XCD-grouped work-plan order, the oracle's restatement of faiss' split_clusters, and the CPU path of
`Interaction` against the reference's own maps (tests/golden/graph_build.npz)."""
import numpy as np
import pytest

from oracle import oracle_np as O


def test_encoded_keys_sort_like_python_sorted():
    """encoders.encode_raw_ids: lexicographic unsigned order of the key words == Python's `sorted()` on the raw ids
    (ncl.py:60-61 sorts the raw id STRINGS: '10' < '9'), for integers, short / long strings and non-ASCII text."""
    from recommendation_amd.encoders import encode_raw_ids
    rng = np.random.default_rng(0)
    cases = {
        "ints": rng.integers(-500, 500, 300).tolist(),
        "digits": [str(v) for v in rng.integers(0, 2000, 300)],
        "prefixes": ["", "a", "ab", "abc", "abcdefgh", "abcdefghi", "abcdefghé", "b", "B", "0", "00"],
        "long": ["customer-%06d-%s" % (v, "x" * (v % 5)) for v in rng.integers(0, 400, 300)],
        "unicode": ["é%d" % v for v in range(40)] + ["z%dü" % v for v in range(40)] + ["中%d" % v for v in range(20)],
    }
    for name, raw in cases.items():
        keys = encode_raw_ids(raw)
        assert keys is not None and keys.dtype == np.uint64 and keys.shape[0] == len(raw), name
        order = sorted(range(len(raw)), key=lambda i: tuple(int(x) for x in keys[i]))
        assert [raw[i] for i in order] == sorted(raw), name
        # equal ids <-> equal keys (the dense-id map is a bijection on distinct ids)
        assert len({tuple(k) for k in keys.tolist()}) == len(set(raw)), name
    assert encode_raw_ids([1.5, 2.5]) is None and encode_raw_ids([("a", 1), ("b", 2)]) is None     # other types: host path


def test_interaction_cpu_path_matches_reference_maps(golden):
    from recommendation_amd.encoders import Interaction
    g = golden("graph_build.npz")
    train = [[u, i, 1.0] for u, i in zip(g["train_user"].tolist(), g["train_item"].tolist())]
    d = Interaction({}, train, train[:5], device="cpu")
    assert [d.id2user[k] for k in range(d.user_num)] == g["sorted_user_ids"].tolist()
    assert [d.id2item[k] for k in range(d.item_num)] == g["sorted_item_ids"].tolist()
    rp, c, _, _ = O.coo_to_csr_stable(g["coo_row"], g["coo_col"], g["coo_data"], d.user_num + d.item_num)
    assert np.array_equal(d.norm_adj.rowptr.numpy(), rp) and np.array_equal(d.norm_adj.col.numpy(), c)
    d = Interaction({}, train, train[:5], device="cpu", normalised=True, id_order="first_seen")
    assert [d.id2user[k] for k in range(d.user_num)] == g["seen_user_ids"].tolist()
    assert np.array_equal(d.norm_adj.rowptr.numpy(), g["norm_indptr"])
    assert np.array_equal(d.norm_adj.col.numpy().astype(np.int64), g["norm_indices"])
    with pytest.raises(ValueError):
        Interaction({}, train, train[:5], device="cpu", reorder="spectral")           # the renumbering runs on the GPU only


def test_xcd_grouped_order_properties():
    """reorder.xcd_grouped_order on synthetic descriptors: a permutation of the partitions plus pads, every group on one
    XCD (workgroup index mod 8), groups contiguous inside an XCD's sequence, XCD loads balanced to within one group."""
    from recommendation_amd.reorder import xcd_grouped_order
    rng = np.random.default_rng(1)
    n_rows, n_groups = 5000, 41
    group = np.sort(rng.integers(0, n_groups, n_rows))
    row0 = np.sort(rng.choice(n_rows, 900, replace=False))
    desc = np.zeros((row0.size, 4), dtype=np.int64)
    desc[:, 2] = row0 | (np.int64(1) << 32)
    desc[:, 3] = -1
    order = xcd_grouped_order(desc, group)
    real = order[order >= 0]
    assert sorted(real.tolist()) == list(range(row0.size)) and order.size % 32 == 0
    pos = np.flatnonzero(order >= 0)
    xcd = (pos // 4) % 8
    pg = group[row0[real]]
    loads = np.bincount(xcd, minlength=8)
    biggest = np.bincount(pg).max()
    assert loads.max() - loads.min() <= biggest
    for gid in np.unique(pg):
        assert np.unique(xcd[pg == gid]).size == 1
    for x in range(8):
        seq = pg[xcd == x]
        assert np.flatnonzero(np.diff(seq) != 0).size + 1 == np.unique(seq).size


def test_split_clusters_restatement_properties():
    """oracle_np.kmeans_split_clusters (faiss Clustering.cpp split_clusters with Philox trials): every empty cluster is
    re-seeded from a cluster with >= 2 points, points are conserved, no cluster ends empty, the perturbation is the
    symmetric (1 +- 1/1024); with n == k (acceptance probabilities undefined) it falls back to the largest cluster."""
    rng = np.random.default_rng(2)
    k, d = 30, 8
    cent = rng.standard_normal((k, d))
    counts = rng.integers(2, 60, k).astype(np.float64)
    empties = [3, 4, 17, 29]
    counts[empties] = 0
    counts[5] = 1                                            # a one-point cluster must never be the donor
    before, total = cent.copy(), counts.sum()
    n = O.kmeans_split_clusters(cent, counts, int(total) + 1000, seed=1234, it=3)
    assert n == len(empties) and counts.sum() == total and (counts > 0).all() and counts[5] == 1
    for ci in empties:
        ratio = cent[ci] / np.where(cent[ci] != 0, cent[ci], 1)
        assert np.all(ratio == 1)
        # some original centroid c0 with cent[ci] = c0 * (1 +- eps) alternating
        donors = [j for j in range(k) if np.allclose(cent[ci] / (1 + (1 / 1024) * np.where(np.arange(d) % 2 == 0, 1, -1)), before[j], rtol=3e-3)]
        assert donors and all(before[j] is not None for j in donors)
    # deterministic in (seed, it); different iterations draw differently
    c2, n2 = before.copy(), counts.copy()
    c2[empties] = before[empties]
    cent_b, counts_b = before.copy(), np.where(np.isin(np.arange(k), empties), 0, rng.integers(2, 60, k)).astype(np.float64)
    a1, a2 = cent_b.copy(), cent_b.copy()
    O.kmeans_split_clusters(a1, counts_b.copy(), 5000, 7, 0)
    O.kmeans_split_clusters(a2, counts_b.copy(), 5000, 7, 0)
    assert np.array_equal(a1, a2)
    # n == k: fallback to the largest cluster
    cent3 = np.arange(20, dtype=np.float64).reshape(5, 4) + 1
    cnt3 = np.array([1, 1, 1, 2, 0], dtype=np.float64)
    assert O.kmeans_split_clusters(cent3, cnt3, 5, 1234, 0) == 1 and cnt3.tolist() == [1, 1, 1, 1, 1]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a plain invocation (no launcher, WORLD_SIZE unset) starts two ranks itself, relays
    rank 0's line and exits with the children's code; GCR_BENCH_INIT_ONLY stops each rank after the rendezvous (no GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GCR_BENCH_INIT_ONLY"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["dist_world"] == 2 and rec["ranks_seen"] == 2


def test_lgcn_encoder_table_follows_rebound_parameters():
    """`LGCNEncoder.table` is what the propagation reads and the two Parameters are views of it — by construction
    only.  deepcopy, load_state_dict(assign=True) and a bare `.data = ...` re-bind the parameters' storage; the encoder
    must re-stack (never read a stale table)."""
    import copy
    import types
    import torch
    from recommendation_amd.encoders import LGCNEncoder
    data = types.SimpleNamespace(user_num=5, item_num=3, device=torch.device("cpu"), norm_adj=None)
    enc = LGCNEncoder(data, 8, 2)

    def aliased(e):
        u, i = e.embedding_dict["user_emb"], e.embedding_dict["item_emb"]
        return u.data_ptr() == e.table.data_ptr() and i.data_ptr() == e.table[5:].data_ptr()

    assert aliased(enc)
    twin = copy.deepcopy(enc)
    assert aliased(twin) and twin.table.data_ptr() != enc.table.data_ptr()
    assert torch.equal(twin.table, enc.table) and twin.norm_adj is enc.norm_adj
    with torch.no_grad():
        twin.embedding_dict["user_emb"].add_(1.0)              # what an optimiser does to the copy's parameter
    assert torch.equal(twin.table[:5], twin.embedding_dict["user_emb"].data) and not torch.equal(twin.table, enc.table)
    sd = {k: v.clone() + 2.0 for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd, assign=True)
    assert aliased(enc) and torch.equal(enc.table[5:], sd["embedding_dict.item_emb"])
    enc.embedding_dict["user_emb"].data = torch.zeros(5, 8)
    assert not enc._aliased()
    assert torch.equal(enc.restack()[:5], torch.zeros(5, 8)) and aliased(enc)


def test_encode_raw_ids_decides_from_the_whole_column():
    import numpy as np
    from recommendation_amd.encoders import encode_raw_ids
    assert encode_raw_ids([3, 1, 2] * 40) is not None and encode_raw_ids(["b", "a"] * 70) is not None
    assert encode_raw_ids([1] * 100 + ["7"]) is None           # mixed after position 64: sorted() would raise
    assert encode_raw_ids(np.array([1, 2 ** 63 + 5], dtype=np.uint64)) is None
    assert encode_raw_ids([True, False]) is None
