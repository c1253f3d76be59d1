"""SURVEY §8f.3 remainder: raw id -> dense id maps on the device (gcr_dense_ids_u64) and the locality-aware renumbering
carried as a permutation (recommendation_amd/reorder.py).  Integer work: bit-exact with the reference's own maps and
adjacency (tests/golden/graph_build.npz, written by directau.Interaction / selfcf.Interaction) — after un-permuting, for the
re-numbered operator."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _triples(g):
    return [[u, i, 1.0] for u, i in zip(g["train_user"].tolist(), g["train_item"].tolist())]


def test_device_id_maps_match_reference_maps(golden):
    from recommendation_amd.encoders import Interaction
    g = golden("graph_build.npz")
    train = _triples(g)
    # ncl.py:55-66: sorted raw-id order; the raw (0/1, duplicates kept) adjacency of ncl.py:74-85
    d = Interaction({}, train, train[:10], device="cuda")
    assert [d.id2user[k] for k in range(d.user_num)] == g["sorted_user_ids"].tolist()
    assert [d.id2item[k] for k in range(d.item_num)] == g["sorted_item_ids"].tolist()
    assert d.user == {u: k for k, u in enumerate(g["sorted_user_ids"].tolist())}
    rp, c, _, _ = O.coo_to_csr_stable(g["coo_row"], g["coo_col"], g["coo_data"], d.user_num + d.item_num)
    assert np.array_equal(d.norm_adj.rowptr.cpu().numpy(), rp) and np.array_equal(d.norm_adj.col.cpu().numpy(), c)
    # selfcf.py:279-306: first-appearance order; normalised adjacency
    d = Interaction({}, train, train[:10], device="cuda", normalised=True, id_order="first_seen")
    assert [d.id2user[k] for k in range(d.user_num)] == g["seen_user_ids"].tolist()
    assert [d.id2item[k] for k in range(d.item_num)] == g["seen_item_ids"].tolist()
    assert np.array_equal(d.norm_adj.rowptr.cpu().numpy(), g["norm_indptr"])
    assert np.array_equal(d.norm_adj.col.cpu().numpy().astype(np.int64), g["norm_indices"])
    np.testing.assert_allclose(d.norm_adj.val.cpu().numpy(), g["norm_data"], rtol=3e-7)
    # the lazily built Python views agree with the reference's containers
    assert d.training_set_u[train[0][0]] == {t[1] for t in train if t[0] == train[0][0]}
    assert d.training_data[:3] == [(t[0], t[1]) for t in train[:3]]


@pytest.mark.parametrize("kind", ["int", "short_str", "long_str", "unicode"])
@pytest.mark.parametrize("order", ["sorted", "first_seen"])
def test_dense_ids_equal_python_maps(kind, order):
    """gcr_dense_ids_u64 against the reference's two constructions in plain Python, on ids that exercise the encoding:
    integers incl. negatives, strings where lexicographic != numeric order ('10' < '9'), ids wider than one 8-byte key
    word (folded word by word) with long common prefixes, and non-ASCII code points."""
    from recommendation_amd.encoders import dense_ids_device, encode_raw_ids
    rng = np.random.default_rng(3)
    n = 5000
    base = rng.integers(0, 700, n)
    if kind == "int":
        raw = (base - 350).tolist()
    elif kind == "short_str":
        raw = [str(v) for v in base]
    elif kind == "long_str":
        raw = ["customer-id-%05d-x" % v if v % 3 else "customer-id-%d" % v for v in base]
    else:
        raw = ["é%d" % v if v % 2 else "z%dü" % v for v in base]
    keys = encode_raw_ids(raw)
    dense, first = dense_ids_device(keys, "cuda", order)
    if order == "sorted":
        ref_list = sorted(set(raw))
    else:
        ref_list = list(dict.fromkeys(raw))
    ref = {r: k for k, r in enumerate(ref_list)}
    assert dense.cpu().tolist() == [ref[r] for r in raw]
    assert [raw[p] for p in first] == ref_list


def test_renumbering_is_a_permutation_of_the_reference_graph(golden):
    """reorder='spectral': ids are re-numbered for locality, `perm_user` / `perm_item` carry reference id -> new id.
    Un-permuting the operator must give the reference's neighbour lists bit for bit, and every id the object hands out
    must be consistent with its own operator."""
    from recommendation_amd.encoders import Interaction
    g = golden("graph_build.npz")
    train = _triples(g)
    ref = Interaction({}, train, train[:10], device="cuda", normalised=True)
    d = Interaction({}, train, train[:10], device="cuda", normalised=True, reorder="spectral", rows_per_cluster=8,
                    reorder_guard=False)                                  # (forced: the guard would skip a toy graph)
    pu, pi = d.perm_user, d.perm_item
    assert sorted(pu.tolist()) == list(range(ref.user_num)) and sorted(pi.tolist()) == list(range(ref.item_num))
    for raw, k in ref.user.items():
        assert d.user[raw] == pu[k]
    for raw, k in ref.item.items():
        assert d.item[raw] == pi[k]
    n_u, n = ref.user_num, ref.user_num + ref.item_num
    perm = np.concatenate([pu, n_u + pi])                       # reference node id -> new node id
    inv = np.argsort(perm)
    rp, col = d.norm_adj.rowptr.cpu().numpy(), d.norm_adj.col.cpu().numpy()
    rrp, rcol, rval = ref.norm_adj.rowptr.cpu().numpy(), ref.norm_adj.col.cpu().numpy(), ref.norm_adj.val.cpu().numpy()
    val = d.norm_adj.val.cpu().numpy()
    for r in range(n):
        new = perm[r]
        got = inv[col[rp[new]:rp[new + 1]]]
        order = np.argsort(got)
        assert np.array_equal(got[order], rcol[rrp[r]:rrp[r + 1]])              # neighbour sets: bit-exact
        np.testing.assert_allclose(val[rp[new]:rp[new + 1]][order], rval[rrp[r]:rrp[r + 1]], rtol=3e-7)
    # propagation commutes with the renumbering (values: fp32, summation order inside a row differs)
    from recommendation_amd import functional as Fn
    x = torch.randn(n, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
    xp = torch.empty_like(x)
    xp[torch.from_numpy(perm).cuda()] = x
    y_ref = Fn.lightgcn_propagate(ref.norm_adj, x, 3, "mean")
    y_new = Fn.lightgcn_propagate(d.norm_adj, xp, 3, "mean")
    torch.testing.assert_close(y_new[torch.from_numpy(perm).cuda()], y_ref, rtol=1e-5, atol=1e-6)
    assert d.norm_adj.plan.grouped


def test_grouped_plan_covers_every_partition_once():
    """reorder.xcd_grouped_order: every partition exactly once, pads only, each group's partitions on one XCD (positions
    whose workgroup index is congruent mod 8), groups contiguous within their XCD's sequence; and the SpMM over a
    grouped plan equals the plain plan's result bit for bit (same partitions, another order)."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    from recommendation_amd.reorder import xcd_grouped_order
    n_u, n_i = 6000, 900
    u, i = O.synthetic_interactions(n_u, n_i, 90000, seed=5)
    rng = np.random.default_rng(0)
    group = np.concatenate([np.sort(rng.integers(0, 37, n_u)), np.sort(rng.integers(0, 37, n_i))])
    plain = ra.CsrGraph.bipartite_sym_norm(u, i, n_u, n_i, "cuda")
    grouped = ra.CsrGraph.bipartite_sym_norm(u, i, n_u, n_i, "cuda", row_group=group)
    desc = plain.plan.desc_host
    order = xcd_grouped_order(desc, group)
    real = order[order >= 0]
    assert sorted(real.tolist()) == list(range(desc.shape[0]))
    pos = np.flatnonzero(order >= 0)
    xcd = (pos // 4) % 8
    pg = group[(desc[real, 2] & 0xFFFFFFFF)]
    for gid in np.unique(pg):
        assert np.unique(xcd[pg == gid]).size == 1
    for x in range(8):
        seq = pg[xcd == x]
        change = np.flatnonzero(np.diff(seq) != 0).size + 1
        assert change == np.unique(seq).size                     # each group contiguous in its XCD's sequence
    xin = torch.randn(n_u + n_i, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    a, b = torch.empty_like(xin), torch.empty_like(xin)
    Fn.spmm_into(plain, xin, y=a)
    Fn.spmm_into(grouped, xin, y=b)
    assert torch.equal(a, b)


def test_renumbered_model_ranks_like_the_reference_numbering():
    """End to end through the drop-in classes: an NCLModel built with reorder='spectral' whose embedding rows are the
    permuted copy of a plain model's must produce the same propagated embeddings (per raw id), the same full-ranking
    recommendation lists and the same metric report — the renumbering is invisible above `Interaction`."""
    from recommendation_amd.ncl import NCLModel
    rng = np.random.default_rng(0)
    n_u, n_i = 400, 150
    pairs = set()
    while len(pairs) < 6000:
        u = int(rng.integers(0, n_u))
        i = int(rng.integers(0, n_i // 5)) * 5 + u % 5 if rng.random() < 0.85 else int(rng.integers(0, n_i))
        pairs.add((u, i))
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    train = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[:5000]]
    test = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[5000:]]
    conf = {"model": {"name": "NCL", "type": "graph"}, "embedding.size": 64, "batch.size": 512, "learning.rate": 0.005,
            "reg.lambda": 1e-4, "max.epoch": 1, "item.ranking.topN": [10, 20],
            "NCL": {"n_layers": 2, "tau": 0.1, "ssl_reg": 1e-4, "proto_reg": 1e-4, "alpha": 1.0, "num_clusters": 8,
                    "hyper_layers": 1}}
    ref = NCLModel(conf, train, test, device="cuda", seed=1)
    new = NCLModel(conf, train, test, device="cuda", seed=1, reorder="spectral", reorder_guard=False)
    assert new.data.perm_user is not None
    pu = torch.from_numpy(new.data.perm_user).cuda()
    pi = torch.from_numpy(new.data.perm_item).cuda()
    with torch.no_grad():
        new.model.embedding_dict["user_emb"][pu] = ref.model.embedding_dict["user_emb"] * 0.1
        new.model.embedding_dict["item_emb"][pi] = ref.model.embedding_dict["item_emb"] * 0.1
        ref.model.table.mul_(0.1)
        for m in (ref, new):
            m.model.eval()
            m.user_emb, m.item_emb, _ = m.model()
    torch.testing.assert_close(new.user_emb[pu], ref.user_emb, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(new.item_emb[pi], ref.item_emb, rtol=1e-5, atol=1e-7)
    rec_ref, rec_new = ref.test(), new.test()
    assert rec_ref.keys() == rec_new.keys()
    same = sum([i for i, _ in rec_ref[u]] == [i for i, _ in rec_new[u]] for u in rec_ref)
    assert same >= 0.98 * len(rec_ref)                      # fp32 near-ties may swap neighbours in a list
    from recommendation_amd.evaluate import ranking_evaluation
    a, b = ranking_evaluation(ref.data.test_set, rec_ref, [10, 20]), ranking_evaluation(new.data.test_set, rec_new, [10, 20])
    va = [float(x.split(":")[1]) for x in a if ":" in x]
    vb = [float(x.split(":")[1]) for x in b if ":" in x]
    np.testing.assert_allclose(va, vb, atol=2e-3)


def test_dense_ids_edge_cases():
    """gcr_dense_ids_u64: a single record, all-equal keys, already-dense keys, and the empty input."""
    from recommendation_amd import _lib
    from recommendation_amd.encoders import dense_ids_device
    for order in ("sorted", "first_seen"):
        d, first = dense_ids_device(np.array([[7]], dtype=np.uint64), "cuda", order)
        assert d.cpu().tolist() == [0] and first.tolist() == [0]
        d, first = dense_ids_device(np.full((1000, 1), 42, dtype=np.uint64), "cuda", order)
        assert int(d.abs().sum()) == 0 and first.tolist() == [0]
        keys = np.arange(5000, dtype=np.uint64)[::-1].copy().reshape(-1, 1)
        d, first = dense_ids_device(keys, "cuda", order)
        want = (np.arange(5000)[::-1] if order == "sorted" else np.arange(5000))
        assert np.array_equal(d.cpu().numpy(), want)
    L = _lib.lib()
    nu = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    _lib.check(L.gcr_dense_ids_u64(None, 0, 0, None, None, _lib.dptr(nu), None, _lib.cur_stream()), "empty")
    assert int(nu) == 0
    assert L.gcr_dense_ids_u64(None, 5, 0, None, None, _lib.dptr(nu), None, _lib.cur_stream()) != 0      # null buffers


def test_bitmap_set_ignores_out_of_range_ids():
    from recommendation_amd import functional as Fn
    idx = torch.tensor([0, 31, 32, 95, 95, -1, 96, 10 ** 9], device="cuda")
    bits = Fn.active_rows_bitmap(idx, 96).cpu().numpy().view(np.uint32)
    assert bits.tolist() == [0x80000001, 0x00000001, 0x80000000]


def test_renumbering_guard_skips_what_it_cannot_speed_up(golden):
    """`Interaction(reorder="spectral")` must never slow a graph down: a toy graph (fits the caches) and a graph WITHOUT
    community structure (uniform users x Zipf items, the headline benchmark's law) keep the reference numbering, with the
    reason recorded; a planted-community graph of the same size is renumbered and measures faster."""
    import bench
    from recommendation_amd import reorder as R
    from recommendation_amd.encoders import Interaction
    g = golden("graph_build.npz")
    toy = Interaction({}, _triples(g), [], device="cuda", normalised=True, reorder="spectral")
    assert toy.perm_user is None and "fit the caches" in toy.reorder_decision["reason"]
    dev = torch.device("cuda", 0)
    n_u, n_i, n_e = 1_000_000, 100_000, 10_000_000           # BASELINE configs[1] sizes (smaller tables live in the caches anyway)
    users, items = bench.synth_interactions_device(n_u, n_i, n_e, 7, dev)
    pu, pi, group, dec = R.guarded_locality_permutation(users, items, n_u, n_i, dev, rows_per_cluster=8192)
    assert pu is None and not dec["applied"], dec
    assert dec["ritz_top"] < 0.9 * dec["noise_bulk_edge"] or dec["ms_per_layer_after"] > 0.97 * dec["ms_per_layer_before"], dec
    del users, items
    users, items, _, _ = bench.synth_community_interactions_device(n_u, n_i, n_e, 7, dev, n_u // 8192, 0.85)
    pu, pi, group, dec = R.guarded_locality_permutation(users, items, n_u, n_i, dev, rows_per_cluster=8192)
    assert pu is not None and dec["applied"] and dec["ritz_top"] > 0.9 * dec["noise_bulk_edge"], dec
    assert dec["ms_per_layer_after"] < 0.97 * dec["ms_per_layer_before"], dec
