"""Full-ranking evaluation with the reference's interface:

  rank_topk            the U x I score matrix + masking + top-N of lightgcn.py:48-57, gcl.py:87-96,
                       ncl.py:253-264 on the GPU (gcr_score_rows_f32 + gcr_topk_masked_f32)
  test                 GraphRecommender.test  ncl.py:253-264  -> {user: [(item_name, score), ...]}
  ranking_metrics      per-user hits / DCG / IDCG on the device (gcr_rank_metrics), means on the host
  ranking_evaluation   the reference's report lines (ncl.py:165-177) from {user: {item}} / {user: [(item, score)]}
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from . import functional as Fn


# True: catalogues of >= 16384 items rank through gcr_rank_fused_f32 (the score matrix never reaches HBM)
FUSED_RANK = True
FUSED_CHUNK_USERS = 16384          # 48 KB of workspace per user (sample scores + candidate regions)


def _rank_two_call(L, ue, ie, user_ids, user_rowptr, user_items_sorted, top_n, top_items, top_scores, chunk_bytes):
    """[chunk, I] scores at a time through gcr_score_rows_f32 + gcr_topk_masked_f32."""
    dev = ue.device
    q, n_items, d = user_ids.numel(), ie.shape[0], ue.shape[1]
    chunk = max(1, min(q, int(chunk_bytes // (4 * n_items))))
    scores = torch.empty(chunk, n_items, dtype=torch.float32, device=dev)
    stream = _lib.cur_stream(dev)
    for s in range(0, q, chunk):
        ids = user_ids[s:s + chunk]
        n = ids.numel()
        _lib.check(L.gcr_score_rows_f32(_lib.dptr(ue), _lib.dptr(ids), n, ue.shape[0], _lib.dptr(ie), n_items, d,
                                        _lib.dptr(scores), stream), "gcr_score_rows_f32")
        _lib.check(L.gcr_topk_masked_f32(_lib.dptr(scores), n, n_items, _lib.dptr(ids), ue.shape[0],
                                         _lib.dptr(user_rowptr), _lib.dptr(user_items_sorted), int(top_n),
                                         _lib.dptr(top_items[s:s + n]), _lib.dptr(top_scores[s:s + n]), stream),
                   "gcr_topk_masked_f32")


def rank_topk(user_emb, item_emb, user_ids, user_rowptr, user_items_sorted, top_n, chunk_bytes=4 << 30):
    """Top-`top_n` items per query user, training positives excluded.  user_emb [U, d], item_emb [I, d]
    float32 on the GPU; user_ids int64 [Q]; (user_rowptr, user_items_sorted) the CSR of training
    positives (None, None = no masking).  Returns (items int64 [Q, top_n], scores float32 [Q, top_n]),
    best first; ties go to the smaller item id.  Large catalogues take the fused path (score tile + candidate
    filter in one kernel: no U x I matrix, not even chunk-wise); a user whose candidate list overflowed or came
    up short is re-ranked through the two-call path, as are small catalogues."""
    _lib.require_cuda(user_emb, item_emb)
    L = _lib.lib()
    dev = user_emb.device
    ue, ie = Fn._pad_dim(user_emb.detach()).contiguous(), Fn._pad_dim(item_emb.detach()).contiguous()
    user_ids = torch.as_tensor(user_ids, device=dev, dtype=torch.int64).contiguous()
    q, n_items, d = user_ids.numel(), ie.shape[0], ue.shape[1]
    top_items = torch.empty(q, top_n, dtype=torch.int64, device=dev)
    top_scores = torch.empty(q, top_n, dtype=torch.float32, device=dev)
    if not (FUSED_RANK and q > 0 and L.gcr_rank_fused_supported(n_items, d, int(top_n))):
        _rank_two_call(L, ue, ie, user_ids, user_rowptr, user_items_sorted, top_n, top_items, top_scores, chunk_bytes)
        return top_items, top_scores
    chunk = min(q, FUSED_CHUNK_USERS)
    ws = torch.empty(int(L.gcr_rank_fused_workspace_bytes(chunk)), dtype=torch.uint8, device=dev)
    status = torch.empty(q, dtype=torch.int32, device=dev)
    stream = _lib.cur_stream(dev)
    for s in range(0, q, chunk):
        n = min(chunk, q - s)
        _lib.check(L.gcr_rank_fused_f32(_lib.dptr(ue), _lib.dptr(user_ids[s:s + n]), n, ue.shape[0], _lib.dptr(ie), n_items,
                                        d, _lib.dptr(user_rowptr), _lib.dptr(user_items_sorted), int(top_n),
                                        _lib.dptr(top_items[s:s + n]), _lib.dptr(top_scores[s:s + n]),
                                        _lib.dptr(status[s:s + n]), _lib.dptr(ws), stream), "gcr_rank_fused_f32")
    redo = torch.nonzero(status).flatten()
    if redo.numel():
        ti = torch.empty(redo.numel(), top_n, dtype=torch.int64, device=dev)
        ts = torch.empty(redo.numel(), top_n, dtype=torch.float32, device=dev)
        _rank_two_call(L, ue, ie, user_ids[redo].contiguous(), user_rowptr, user_items_sorted, top_n, ti, ts, chunk_bytes)
        top_items[redo], top_scores[redo] = ti, ts
    return top_items, top_scores


def test(data, user_emb, item_emb, max_n):
    """ncl.py:253-264 (GraphRecommender.test): for every user of data.test_set the max_n best unseen
    items as [(item_name, score), ...]."""
    users = [u for u in data.test_set if u in data.user]
    ids = torch.tensor([data.user[u] for u in users], dtype=torch.int64, device=user_emb.device)
    items, scores = rank_topk(user_emb, item_emb, ids, data.user_rowptr, data.user_items_sorted, max_n)
    items, scores = items.cpu().numpy(), scores.cpu().numpy()
    # a user with fewer than max_n unseen items: the reference pads with its -1e8-scored training
    # items (ncl.py:258-260); those can never be hits, so they are dropped here instead
    return {u: [(data.id2item[int(i)], float(s)) for i, s in zip(items[k], scores[k]) if i >= 0 and np.isfinite(s)]
            for k, u in enumerate(users)}


def ranking_metrics(top_items, test_rowptr, test_items_sorted, cutoffs, n_test_total=None):
    """Hit ratio / precision / recall / NDCG at every cut-off from the ranked lists, the per-user work on the
    device (gcr_rank_metrics).  top_items int64 [Q, K] on the GPU (-1 = padding); (test_rowptr, test_items_sorted)
    the CSR of each QUERY row's test items (ascending inside a row).  Users without a test item do not take part
    (the reference only evaluates users of its test set).  Returns {n: {"Hit Ratio", "Precision", "Recall", "NDCG"}}
    rounded to 5 decimals like the reference's report (ncl.py:133-177).
    n_test_total: the Hit Ratio's denominator when it is not the query rows' own test items — `Metric.hit_ratio`
    (ncl.py:143-145) divides by the test items of ALL users of `origin`, ranked or not."""
    L = _lib.lib()
    dev = top_items.device
    q, k = top_items.shape
    cut = torch.tensor(sorted(int(n) for n in cutoffs), dtype=torch.int32, device=dev)
    nc = cut.numel()
    hits = torch.empty(q, nc, dtype=torch.int32, device=dev)
    dcg = torch.empty(q, nc, dtype=torch.float64, device=dev)
    idcg = torch.empty(q, nc, dtype=torch.float64, device=dev)
    test_rowptr = torch.as_tensor(test_rowptr, device=dev, dtype=torch.int64).contiguous()
    test_items_sorted = torch.as_tensor(test_items_sorted, device=dev, dtype=torch.int32).contiguous()
    _lib.check(L.gcr_rank_metrics(_lib.dptr(top_items.contiguous()), q, k, _lib.dptr(test_rowptr), _lib.dptr(test_items_sorted),
                                  _lib.dptr(cut), nc, _lib.dptr(hits), _lib.dptr(dcg), _lib.dptr(idcg), _lib.cur_stream(dev)),
               "gcr_rank_metrics")
    n_test = (test_rowptr[1:] - test_rowptr[:-1]).cpu().numpy().astype(np.float64)
    on = n_test > 0
    hits_h, dcg_h, idcg_h = hits.cpu().numpy()[on].astype(np.float64), dcg.cpu().numpy()[on], idcg.cpu().numpy()[on]
    n_test = n_test[on]
    out = {}
    for c, n in enumerate(cut.tolist()):
        h = hits_h[:, c]
        ndcg = np.divide(dcg_h[:, c], idcg_h[:, c], out=np.zeros_like(h), where=idcg_h[:, c] > 0)
        hr_den = float(n_test_total) if n_test_total is not None else float(n_test.sum())
        out[n] = {"Hit Ratio": round(float(h.sum() / hr_den), 5), "Precision": round(float(h.sum() / (len(h) * n)), 5),
                  "Recall": round(float(np.mean(h / n_test)), 5), "NDCG": round(float(ndcg.sum() / len(h)), 5)}
    return out


def ranking_evaluation(origin, res, N, device=None):
    """The reference's report interface (ncl.py:165-177): origin {user: {item: 1}}, res {user: [(item, score), ...]}
    -> the list of "Top n" / "Hit Ratio:..." lines for every cut-off in N.  The dictionaries are packed into
    index tensors (host plumbing) and the per-user hit / DCG sums run on the device (`ranking_metrics`; `device`:
    the GPU to use, default the current one).  As in the reference, hits / precision / recall / NDCG run over the users
    present in BOTH dictionaries while the Hit Ratio divides by the test items of every user of `origin`
    (ncl.py:136-145) — a test user that was never ranked (absent from training) lowers it."""
    users = [u for u in origin if u in res]
    names = {}
    for u in users:
        for it in origin[u]:
            names.setdefault(it, len(names))
        for it, _ in res[u]:
            names.setdefault(it, len(names))
    k = max((len(res[u]) for u in users), default=1) or 1
    top = np.full((len(users), k), -1, dtype=np.int64)
    rowptr = np.zeros(len(users) + 1, dtype=np.int64)
    test_items = []
    for r, u in enumerate(users):
        ids = [names[it] for it, _ in res[u]]
        top[r, :len(ids)] = ids
        t = sorted(names[it] for it in origin[u])
        test_items += t
        rowptr[r + 1] = rowptr[r] + len(t)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    m = ranking_metrics(torch.from_numpy(top).to(dev), torch.from_numpy(rowptr), np.asarray(test_items, dtype=np.int32), N,
                        n_test_total=sum(len(origin[u]) for u in origin))
    lines = []
    for n in N:
        lines.append(f"Top {n}\n")
        lines += [f"{name}:{m[int(n)][name]}\n" for name in ("Hit Ratio", "Precision", "Recall", "NDCG")]
    return lines
