"""Loss functions with the reference's names, argument order and return values, computed by the
HIP kernels of libgcr (no M x N score matrix is ever materialised).

  InfoNCE, bpr_loss, l2_reg_loss          ncl.py:116-130 (= ssl4rec.py:16-23)
  info_nce_loss                           gcl.py:28-35 (= univariate/gcl_univariate.py:28-35)
  batch_softmax_loss                      ssl4rec.py:25-30
  ssl_layer_loss, ProtoNCE_loss           ncl.py:358-375 (NCLModel methods; here plain functions
                                          taking what the methods read from `self`)
  lightgcn_bce_loss                       lightgcn.py:109-113 (the `loss_type == "bce"` branch of train_model)
"""
from __future__ import annotations

import torch

from . import functional as Fn


def InfoNCE(view1, view2, temperature: float, b_cos: bool = True):
    """ncl.py:125-130: -mean(diag(log_softmax(view1n @ view2n.T / temperature, dim=1)))."""
    lse, pos = Fn.infonce_stats(view1, view2, None, temperature, normalize=b_cos)
    return (lse - pos).mean()


def info_nce_loss(z1, z2, temp=0.2):
    """gcl.py:28-35: (CE(sim, arange) + CE(sim.T, arange)) / 2, sim = z1n @ z2n.T / temp."""
    if z1.shape[0] != z2.shape[0]:
        raise ValueError("info_nce_loss needs as many rows in z2 as in z1 (labels = arange)")
    lse, pos, col = Fn.infonce_stats(z1, z2, None, temp, normalize=True, want_col=True)
    return ((lse - pos).mean() + (col - pos).mean()) / 2


def batch_softmax_loss(user_emb, item_emb, temperature):
    """ssl4rec.py:25-30: mean(-log(exp(pos/t) / sum_j exp(<u, i_j>/t) + 1e-6)) on normalised rows."""
    lse, pos = Fn.infonce_stats(user_emb, item_emb, None, temperature, normalize=True)
    return (-torch.log(torch.exp(pos - lse) + 1e-6)).mean()


def ssl_layer_loss(context, initial, user, item, user_num, ssl_temp, ssl_reg, alpha):
    """ncl.py:358-367 (NCLModel.ssl_layer_loss): structure contrast of the batch's context-layer rows
    against ALL layer-0 rows of the same side; positives = the row's own layer-0 embedding; summed."""
    iu, ii = Fn.split_rows(initial, user_num)
    dev = context.device
    user = torch.as_tensor(user, device=dev, dtype=torch.int64)
    item = torch.as_tensor(item, device=dev, dtype=torch.int64)
    # batch rows gathered straight from the stacked table in ONE gather (one sparse backward into
    # `context` instead of slice + gather twice)
    rows = Fn.gather_rows(context, torch.cat([user, item + user_num]))
    lse_u, pos_u = Fn.infonce_stats(rows[:user.numel()], iu, user, ssl_temp, normalize=True)
    lse_i, pos_i = Fn.infonce_stats(rows[user.numel():], ii, item, ssl_temp, normalize=True)
    return ssl_reg * ((lse_u - pos_u).sum() + alpha * (lse_i - pos_i).sum())


def ProtoNCE_loss(initial_emb, user_idx, item_idx, user_num, user_centroids, user_2cluster, item_centroids,
                  item_2cluster, ssl_temp, proto_reg, batch_size):
    """ncl.py:369-375 (NCLModel.ProtoNCE_loss): InfoNCE(e0[idx], centroids[assign[idx]]) * batch_size."""
    dev = initial_emb.device
    user_idx = torch.as_tensor(user_idx, device=dev, dtype=torch.int64)
    item_idx = torch.as_tensor(item_idx, device=dev, dtype=torch.int64)
    u2c = user_centroids.to(dev)[user_2cluster.to(dev)[user_idx]]
    i2c = item_centroids.to(dev)[item_2cluster.to(dev)[item_idx]]
    rows = Fn.gather_rows(initial_emb, torch.cat([user_idx, item_idx + user_num]))
    loss_user = InfoNCE(rows[:user_idx.numel()], u2c, ssl_temp) * batch_size
    loss_item = InfoNCE(rows[user_idx.numel():], i2c, ssl_temp) * batch_size
    return proto_reg * (loss_user + loss_item)


def _rows(x):
    return torch.arange(x.shape[0], device=x.device)


def bpr_loss(user_emb, pos_item_emb, neg_item_emb):
    """ncl.py:116-120: mean(-log(10e-6 + sigmoid(<u,p> - <u,n>))) on already-gathered rows."""
    r = _rows(user_emb)
    return Fn.bpr_sums(user_emb, torch.cat([pos_item_emb, neg_item_emb]), r, r, r + user_emb.shape[0], Fn.BPR_NCL)[0] \
        / user_emb.shape[0]


def bpr_loss_logsigmoid(user_emb, pos_emb, neg_emb):
    """sept.py:34-38 / gcl.py:219-221: -mean(logsigmoid(pos - neg))."""
    r = _rows(user_emb)
    return Fn.bpr_sums(user_emb, torch.cat([pos_emb, neg_emb]), r, r, r + user_emb.shape[0], Fn.BPR_LOGSIGMOID)[0] \
        / user_emb.shape[0]


def l2_reg_loss(reg, *args):
    """ncl.py:122-123: reg * sum_x ||x||_F / rows(x) (tiny reductions; stays in torch)."""
    return reg * sum(torch.norm(x, p=2) / x.shape[0] for x in args)


def lightgcn_bce_loss(user_emb, item_emb, pos_u, pos_i, engine=None):
    """lightgcn.py:95-96,109-113 (`loss_type == "bce"`): scores = user_emb[pos_u] @ item_emb.T  [B, I], labels one-hot at
    pos_i, `F.binary_cross_entropy_with_logits(scores, labels)` = mean over B * I of softplus(s) - y s.  The [B, I]
    matrix is never formed: the softplus part is the fused row sum (rows of one user are equal, so a batch with at least
    as many samples as users runs over the distinct users weighted by their sample counts), the positive logits are a
    gathered dot product.  For the full batch over a graph's own edge list use `Fn.bce_edge_loss` (LightGCN.loss)."""
    dev = user_emb.device
    pos_u = torch.as_tensor(pos_u, device=dev, dtype=torch.int64).reshape(-1)
    pos_i = torch.as_tensor(pos_i, device=dev, dtype=torch.int64).reshape(-1)
    bsz, n_users, n_items = pos_u.numel(), user_emb.shape[0], item_emb.shape[0]
    if bsz >= n_users:
        cnt = torch.bincount(pos_u, minlength=n_users).to(torch.float32)
        soft = torch.dot(Fn.bce_softplus_rowsum(user_emb, item_emb, engine), cnt)
    else:
        soft = Fn.bce_softplus_rowsum(Fn.gather_rows(user_emb, pos_u), item_emb, engine).sum()
    pos = (Fn.gather_rows(user_emb, pos_u) * Fn.gather_rows(item_emb, pos_i)).sum()
    return (soft - pos) / (float(bsz) * float(n_items))


def bpr_gather_loss(user_tab, item_tab, u_idx, i_idx, j_idx, variant=Fn.BPR_NCL):
    """Fused form used by the model classes: gathers + BPR in one kernel, gradients scattered
    straight into the tables.  Returns (mean bpr loss, sum|u|^2, sum|p|^2, sum|n|^2)."""
    sums = Fn.bpr_sums(user_tab, item_tab, u_idx, i_idx, j_idx, variant)
    return sums[0] / max(len(u_idx), 1), sums[1], sums[2], sums[3]


def neighbor_discrimination(positive, emb, aug_emb, temperature=0.1):
    """univariate/sept_social.py:408-420 on already-selected rows (emb = emb[unique_u], aug_emb =
    aug_user_embeddings[unique_u]): -sum_i log( sum_k exp(<e_i, a_{p_ik}> / t) / sum_j exp(<e_i, a_j> / t) )
    on normalised rows; `positive` int64 [B', k] are the pseudo-label indices into aug_emb.  The B' x B'
    denominator is the fused row logsumexp (gcr_infonce_fwd_f32); the k positives per row stay a
    [B', k] gather."""
    import torch.nn.functional as F
    lse, _ = Fn.infonce_stats(emb, aug_emb, None, temperature, normalize=True)
    e, a = F.normalize(emb, dim=1), F.normalize(aug_emb, dim=1)
    pos = (e.unsqueeze(1) * a[positive]).sum(2) / temperature
    return -(torch.logsumexp(pos, dim=1) - lse).sum()


def grace_infonce_loss(h1, h2, tau, intraview_negs=True, exclude_self=False):
    """univariate/grace.py: `DualBranchContrast(loss=InfoNCE(tau), mode='L2L', intraview_negs=...)(h1, h2)`
    (:462-502 with SameScaleSampler :396-419 and InfoNCE.compute :218-224): for anchor h1_i the positive is
    h2_i, the negatives every other h2_j and — with intra-view negatives — the rows of h1 itself;
    symmetric in the two views, averaged.  The [N, 2N] masked similarity matrix of the reference becomes
    two fused row logsumexps per direction (over the other view, over the own view) joined by logaddexp.

    exclude_self=False is what the reference computes when the model calls it: `add_extra_mask` (:448-455)
    replaces the sampler's negative mask by `1 - pos_mask` whenever extra_neg_mask is None, so the anchor's
    own row h1_i (similarity 1/tau) stays in the denominator.  exclude_self=True is the sampler's own mask
    (`1 - eye` on the intra-view block, :399-404; what the reference gives with extra_neg_mask = ones): the
    diagonal is left out inside the kernel (GCR_INFONCE_EXCLUDE_DIAGONAL).  Both are pinned by
    tests/golden/grace.npz."""
    if h1.shape != h2.shape:
        raise ValueError("grace_infonce_loss needs two views of the same nodes")

    def one_direction(a, b):
        lse_inter, pos = Fn.infonce_stats(a, b, None, tau, normalize=True)
        if not intraview_negs:
            return (lse_inter - pos).mean()
        lse_intra, _ = Fn.infonce_stats(a, a, None, tau, normalize=True, exclude_diagonal=exclude_self)
        return (torch.logaddexp(lse_inter, lse_intra) - pos).mean()

    return (one_direction(h1, h2) + one_direction(h2, h1)) * 0.5
