"""Full-ranking evaluation: metric bookkeeping pinned against the reference's own
ranking_evaluation output (tests/golden/eval.json, CPU), and the GPU score + mask + top-N kernels
against a numpy argsort of float64 scores."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval.json")


def test_oracle_ranking_report_matches_reference():
    """ncl.py:133-177: Hit Ratio / Precision / Recall / NDCG lines for N in [10, 20, 30, 50] (CPU oracle)."""
    g = json.load(open(GOLDEN))
    res = {u: [tuple(p) for p in lst] for u, lst in g["res"].items()}
    assert O.ranking_report(g["origin"], res, g["N"]) == g["lines"]


@pytest.mark.gpu
def test_ranking_evaluation_strings_match_reference():
    """The same lines from the device-side per-user hits / DCG sums (gcr_rank_metrics)."""
    from recommendation_amd.evaluate import ranking_evaluation
    g = json.load(open(GOLDEN))
    res = {u: [tuple(p) for p in lst] for u, lst in g["res"].items()}
    assert ranking_evaluation(g["origin"], res, g["N"]) == g["lines"]
    assert ranking_evaluation(g["origin"], res, [10, 70]) == O.ranking_report(g["origin"], res, [10, 70])   # cut-off > list
    # a test user that was never ranked (absent from training, evaluate.test() drops it): hits / precision / recall /
    # NDCG ignore it, the Hit Ratio's denominator still counts its test items (Metric.hit_ratio, ncl.py:143-145)
    partial = {u: lst for k, (u, lst) in enumerate(res.items()) if k % 3}
    got = ranking_evaluation(g["origin"], partial, g["N"], device="cuda:0")
    assert got == O.ranking_report(g["origin"], partial, g["N"])
    assert got != ranking_evaluation({u: g["origin"][u] for u in partial}, partial, g["N"])


def _ref_topk(ue, ie, uids, pos, k):
    s = ue[uids].astype(np.float64) @ ie.astype(np.float64).T
    for r, u in enumerate(uids):
        s[r, list(pos.get(int(u), []))] = -np.inf
    order = np.lexsort((np.arange(s.shape[1])[None, :].repeat(len(uids), 0), -s), axis=1)[:, :k]
    return order, np.take_along_axis(s, order, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("n_u,n_i,d,k", [(300, 1682, 64, 50), (200, 257, 32, 10), (100, 5000, 128, 20), (64, 40, 64, 50),
                                         (700, 20000, 64, 50), (300, 70001, 128, 20), (150, 16384, 32, 100)])
def test_rank_topk_matches_numpy(n_u, n_i, d, k):
    """The last three shapes are catalogue-sized (>= 16384 items): the fused score + candidate-filter path
    (gcr_rank_fused_f32), the others the two-call path."""
    from recommendation_amd.evaluate import rank_topk
    rng = np.random.default_rng(n_u + n_i)
    ue = rng.standard_normal((n_u, d)).astype(np.float32)
    ie = rng.standard_normal((n_i, d)).astype(np.float32)
    u, i = O.synthetic_interactions(n_u, n_i, min(n_u * n_i // 3, n_u * 20), seed=1)
    order = np.lexsort((i, u))
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(u, minlength=n_u))]).astype(np.int64)
    items_sorted = i[order].astype(np.int32)
    pos = {}
    for a, b in zip(u, i):
        pos.setdefault(int(a), []).append(int(b))
    uids = rng.permutation(n_u)[: max(1, n_u // 2)]
    got_i, got_s = rank_topk(torch.from_numpy(ue).cuda(), torch.from_numpy(ie).cuda(), uids,
                             torch.from_numpy(rowptr).cuda(), torch.from_numpy(items_sorted).cuda(), k,
                             chunk_bytes=4 * n_i * 37)          # several chunks
    ref_i, ref_s = _ref_topk(ue, ie, uids, pos, k)
    got_i, got_s = got_i.cpu().numpy(), got_s.cpu().numpy()
    kk = min(k, n_i)
    finite = np.isfinite(ref_s[:, :kk])
    np.testing.assert_allclose(got_s[:, :kk][finite], ref_s[:, :kk][finite], rtol=1e-5, atol=1e-5)
    # identical item lists except where two fp32 scores are within rounding of each other
    same = got_i[:, :kk] == ref_i[:, :kk]
    assert same[finite].mean() > 0.995
    for r, c in zip(*np.nonzero(~same & finite)):
        assert abs(ref_s[r, c] - (ue[uids[r]].astype(np.float64) @ ie[got_i[r, c]].astype(np.float64))) < 1e-4
    # never a training positive, never a duplicate
    for r, uq in enumerate(uids):
        row = got_i[r][got_i[r] >= 0]
        assert len(set(row.tolist())) == len(row)
        assert not (set(row[np.isfinite(got_s[r][: len(row)])].tolist()) & set(pos.get(int(uq), [])))
    if k > n_i:
        assert (got_i[:, n_i:] == -1).all()


@pytest.mark.gpu
def test_topk_ties_and_reference_protocol(golden):
    """Exact ties resolve to the smaller item id; `test()` returns the ncl.py:253-264 structure."""
    from recommendation_amd.encoders import Interaction
    from recommendation_amd.evaluate import rank_topk, ranking_evaluation, test as run_test
    ue = torch.zeros(4, 64, device="cuda")
    ue[:, 0] = 1.0
    ie = torch.zeros(300, 64, device="cuda")
    ie[:, 0] = torch.tensor([float(j % 7) for j in range(300)], device="cuda")     # 7 score levels, many ties
    items, scores = rank_topk(ue, ie, [0, 1], None, None, 60)
    exp = sorted(range(300), key=lambda j: (-(j % 7), j))[:60]
    assert items[0].cpu().tolist() == exp and items[1].cpu().tolist() == exp
    assert scores[0].cpu().tolist() == [float(j % 7) for j in exp]
    g = golden("graph_build.npz")
    train = [[a, b, 1.0] for a, b in zip(g["train_user"].tolist(), g["train_item"].tolist())]
    test_set = [[train[k][0], train[(k * 5) % len(train)][1], 1.0] for k in range(0, 90, 3)]
    data = Interaction({}, train, test_set, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(0)
    U = torch.randn(data.user_num, 64, device="cuda", generator=gen)
    V = torch.randn(data.item_num, 64, device="cuda", generator=gen)
    rec = run_test(data, U, V, 20)
    assert set(rec) == set(data.test_set)
    for u, lst in rec.items():
        assert len(lst) <= 20 and all(name not in data.training_set_u[u] for name, _ in lst)
        assert [s for _, s in lst] == sorted((s for _, s in lst), reverse=True)
    lines = ranking_evaluation(data.test_set, rec, [10, 20])
    assert lines[0] == "Top 10\n" and lines[5] == "Top 20\n" and len(lines) == 10
    assert lines == O.ranking_report(data.test_set, rec, [10, 20])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["random", "ascending", "descending", "ties", "hot_tail"])
def test_topk_long_rows_candidate_filter_and_fallback(layout):
    """gcr_topk_masked_f32 on catalogue-length rows (n_items >= 16384: the prefix-threshold candidate filter):
    random order (filter path), scores ascending in item id (the prefix is the row's minimum: > 4096
    candidates, full-select fallback), descending, massive ties at the threshold, and the top scores all at
    the end of the row.  Exact against numpy's stable sort (ties -> smaller item id)."""
    from recommendation_amd import _lib
    rng = np.random.default_rng(7)
    n_q, n_i, k = 9, 40_000, 50
    s = rng.standard_normal((n_q, n_i)).astype(np.float32)
    if layout == "ascending":
        s = np.sort(s, 1)
    elif layout == "descending":
        s = -np.sort(-s, 1)
    elif layout == "ties":
        s = np.round(s * 2).astype(np.float32) / 2          # ~20 distinct values: huge tie groups
    elif layout == "hot_tail":
        s[:, -200:] += 10.0
    train = [np.sort(rng.choice(n_i, 30, replace=False)) for _ in range(n_q)]
    rowptr = np.concatenate([[0], np.cumsum([len(t) for t in train])]).astype(np.int64)
    items = np.concatenate(train).astype(np.int32)
    st = torch.from_numpy(s.copy()).cuda()
    top_i = torch.empty(n_q, k, dtype=torch.int64, device="cuda")
    top_s = torch.empty(n_q, k, dtype=torch.float32, device="cuda")
    rp, it = torch.from_numpy(rowptr).cuda(), torch.from_numpy(items).cuda()
    _lib.check(_lib.lib().gcr_topk_masked_f32(_lib.dptr(st), n_q, n_i, None, n_q, _lib.dptr(rp), _lib.dptr(it), k,
                                              _lib.dptr(top_i), _lib.dptr(top_s), _lib.cur_stream(st.device)),
               "gcr_topk_masked_f32")
    for q in range(n_q):
        row = s[q].copy()
        row[train[q]] = -np.inf
        ref = np.argsort(-row, kind="stable")[:k]           # stable: equal scores keep increasing item id
        assert np.array_equal(top_i[q].cpu().numpy(), ref), (layout, q)
        assert np.array_equal(top_s[q].cpu().numpy(), row[ref])


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["ascending", "descending", "ties", "hot_tail", "few_eligible"])
def test_fused_rank_adversarial_rows(layout):
    """gcr_rank_fused_f32 on rows built to stress the threshold: scores = item_emb[:, 0] (users = e_0), so ascending
    in item id makes the 4096-item sample the row's minimum (candidate overflow -> status 1 -> exact fallback),
    descending puts every winner in the sample, `ties` has huge groups of equal scores at the threshold, `hot_tail`
    the winners at the end, `few_eligible` a user whose training set covers most of the sample's top."""
    from recommendation_amd.evaluate import rank_topk
    rng = np.random.default_rng(3)
    n_q, n_i, k, d = 5, 30_000, 50, 64
    base = rng.standard_normal(n_i).astype(np.float32)
    if layout == "ascending":
        base = np.sort(base)
    elif layout == "descending":
        base = -np.sort(-base)
    elif layout == "ties":
        base = (np.round(base * 2) / 2).astype(np.float32)
    elif layout == "hot_tail":
        base[-200:] += 10.0
    ie = np.zeros((n_i, d), dtype=np.float32)
    ie[:, 0] = base
    ie[:, 1] = rng.standard_normal(n_i).astype(np.float32)          # a second direction so that users differ
    ue = np.zeros((n_q, d), dtype=np.float32)
    ue[:, 0] = 1.0
    ue[:, 1] = np.linspace(0, 0.5, n_q)
    scores = ue.astype(np.float64) @ ie.astype(np.float64).T
    train = [np.sort(rng.choice(n_i, 30, replace=False)) for _ in range(n_q)]
    if layout == "few_eligible":
        train[0] = np.sort(np.argsort(-scores[0, :4096])[:4000])      # almost the whole sample is masked
    rowptr = np.concatenate([[0], np.cumsum([len(t) for t in train])]).astype(np.int64)
    items = np.concatenate(train).astype(np.int32)
    got_i, got_s = rank_topk(torch.from_numpy(ue).cuda(), torch.from_numpy(ie).cuda(), np.arange(n_q),
                             torch.from_numpy(rowptr).cuda(), torch.from_numpy(items).cuda(), k)
    got_i, got_s = got_i.cpu().numpy(), got_s.cpu().numpy()
    for q in range(n_q):
        row = scores[q].copy()
        row[train[q]] = -np.inf
        ref = np.argsort(-row, kind="stable")[:k]
        np.testing.assert_allclose(got_s[q], row[ref], rtol=1e-5, atol=1e-5)
        diff = got_i[q] != ref
        # different ids only where the float64 scores are within f32 rounding of each other
        assert np.all(np.abs(row[got_i[q][diff]] - row[ref[diff]]) < 1e-5), (layout, q)
        assert len(set(got_i[q].tolist())) == k and not (set(got_i[q].tolist()) & set(train[q].tolist()))
