"""Full-ranking evaluation with the reference's interface:

  rank_topk            the U x I score matrix + masking + top-N of lightgcn.py:48-57, gcl.py:87-96,
                       ncl.py:253-264 on the GPU (gcr_score_rows_f32 + gcr_topk_masked_f32)
  test                 GraphRecommender.test  ncl.py:253-264  -> {user: [(item_name, score), ...]}
  Metric / ranking_evaluation   ncl.py:133-177 (host-side bookkeeping over the top-N lists; same
                       strings, same rounding)
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from . import functional as Fn


def rank_topk(user_emb, item_emb, user_ids, user_rowptr, user_items_sorted, top_n, chunk_bytes=4 << 30):
    """Top-`top_n` items per query user, training positives excluded.  user_emb [U, d], item_emb [I, d]
    float32 on the GPU; user_ids int64 [Q]; (user_rowptr, user_items_sorted) the CSR of training
    positives (None, None = no masking).  Returns (items int64 [Q, top_n], scores float32 [Q, top_n]),
    best first; ties go to the smaller item id.  Scores are produced chunk-wise ([chunk, I] fp32 at a
    time), never as one U x I matrix."""
    _lib.require_cuda(user_emb, item_emb)
    L = _lib.lib()
    dev = user_emb.device
    ue, ie = Fn._pad_dim(user_emb.detach()).contiguous(), Fn._pad_dim(item_emb.detach()).contiguous()
    user_ids = torch.as_tensor(user_ids, device=dev, dtype=torch.int64).contiguous()
    q, n_items, d = user_ids.numel(), ie.shape[0], ue.shape[1]
    top_items = torch.empty(q, top_n, dtype=torch.int64, device=dev)
    top_scores = torch.empty(q, top_n, dtype=torch.float32, device=dev)
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


class Metric:
    """ncl.py:133-163."""

    @staticmethod
    def hits(origin, res):
        return {u: len(set(origin[u]).intersection(i[0] for i in res.get(u, []))) for u in origin if u in res}

    @staticmethod
    def hit_ratio(origin, hits):
        return round(sum(hits.values()) / sum(len(origin[u]) for u in origin), 5)

    @staticmethod
    def precision(hits, n):
        return round(sum(hits.values()) / (len(hits) * n), 5)

    @staticmethod
    def recall(hits, origin):
        return round(np.mean([hits[u] / len(origin[u]) for u in hits]), 5)

    @staticmethod
    def NDCG(origin, res, n):
        score = 0
        for u in res:
            dcg = sum(1.0 / math.log2(i + 2) for i, item in enumerate(res[u]) if item[0] in origin[u])
            idcg = sum(1.0 / math.log2(i + 2) for i in range(min(len(origin[u]), n)))
            score += dcg / idcg if idcg else 0
        return round(score / len(res), 5)


def ranking_evaluation(origin, res, N):
    """ncl.py:165-177: the list of result strings for every cut-off in N."""
    results = []
    for n in N:
        pred = {u: res[u][:n] for u in res}
        hits = Metric.hits(origin, pred)
        results.append(f"Top {n}\n")
        results += [f"Hit Ratio:{Metric.hit_ratio(origin, hits)}\n", f"Precision:{Metric.precision(hits, n)}\n",
                    f"Recall:{Metric.recall(hits, origin)}\n", f"NDCG:{Metric.NDCG(origin, pred, n)}\n"]
    return results
