"""Row-sharded LightGCN propagation over the GPUs of one node (one process per GPU, RCCL/xGMI).

The reference is single-process (no collective anywhere); this is the multi-GPU
form of ncl.py:415-422 / lightgcn.py:21-27 required by BASELINE.json.  Partition:

  * USERS are split into `world` contiguous blocks; rank g owns the embedding rows of its users,
    the CSR block R_g = A[users_g, items] and its transpose R_g^T = A[items, users_g];
  * ITEM embeddings are owned in `world` equal shards but replicated for compute.

Per layer (forward), with X_u local and X_i sharded:
      all-gather   X_i (shards -> full)                 || item-side SpMM  P_i = R_g^T X_u   (local users only)
      reduce-scatter P_i -> Y_i shard (sum over ranks)   || user-side SpMM  Y_u = R_g X_i
so the 2.56 GB user table never crosses xGMI and both collectives hide behind the other half of the
SpMM.  The operator is symmetric and linear, so the backward pass is the same schedule applied to
the gradients (item gradients from the loss are reduce-scattered to their owner shard first).

`spmm` is injectable so that the choreography can be exercised with gloo on CPU tensors by the
tests (world_size 2); the default is the HIP kernel.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import functional as Fn
from .graph import CsrGraph


# True: issue the collectives even at world size 1 (they degenerate to copies).  Used by the single-rank RCCL test
# (tests/test_distributed_gpu.py): the only way to run every collective call of this module on the real `nccl`
# backend on a one-GPU box.
FORCE_COLLECTIVES = False


def _multi(world):
    return world > 1 or (FORCE_COLLECTIVES and dist.is_initialized())


def _hip_spmm(graph, x, acc_in=None, acc_scale=1.0, want_y=True, keep_bits=None, val_scale=1.0, y_out=None):
    """`y_out`: a caller-owned (pooled) buffer for y instead of a fresh allocation."""
    y = None
    if want_y:
        y = y_out if y_out is not None else torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device)
    acc = torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device) if acc_in is not None else None
    Fn.spmm_into(graph, x, y=y, acc_in=acc_in, acc_out=acc, acc_scale=acc_scale, keep_bits=keep_bits, val_scale=val_scale)
    return y, acc


class _Buffers:
    """Per-graph pool of the collective staging buffers (the gathered table, the partial item sums, the
    reduce-scatter output): allocated once per (tag, shape) and reused by every layer, every forward and every
    backward — all on the caller's stream, and every collective is waited on before a call returns, so a
    buffer is never rewritten while a previous use is still in flight."""

    def __init__(self):
        self._b = {}

    def get(self, tag, rows, d, device):
        key = (tag, rows, d, str(device))
        t = self._b.get(key)
        if t is None:
            t = torch.empty(rows, d, dtype=torch.float32, device=device)
            self._b[key] = t
        return t


def shard_bounds(n, world):
    """Contiguous, equal-size (padded) shards: returns (per_rank, padded_total)."""
    per = (n + world - 1) // world
    return per, per * world


class ShardedBipartiteGraph:
    """Rank-local blocks of the symmetric bipartite operator.

    r_ui : CsrGraph [U_g, I_pad]   user rows of this rank -> all items
    r_iu : CsrGraph [I_pad, U_g]   all items -> this rank's users (the transpose block)
    Items are padded to a multiple of `world` (padding items have no edges) so that every owner
    shard has `items_per_rank` rows.
    """

    def __init__(self, r_ui, r_iu, n_local_users, num_items, items_per_rank, rank, world, group=None):
        self.r_ui, self.r_iu = r_ui, r_iu
        self.n_local_users, self.num_items = n_local_users, num_items
        self.items_per_rank, self.rank, self.world, self.group = items_per_rank, rank, world, group
        self.items_padded = items_per_rank * world
        self.buffers = _Buffers()
        self._comm_stream = None

    def comm_stream(self, device):
        """The side stream the item-shard chain (all-gather -> reduce-scatter -> shard update) is issued from."""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm_stream

    @classmethod
    def from_local_interactions(cls, local_uid, iid, n_local_users, num_items, user_deg, item_deg_global, rank, world,
                                device, group=None, graph_cls=CsrGraph, **kw):
        """local_uid in [0, U_g), iid in [0, num_items): this rank's interactions (unique pairs).
        user_deg [U_g], item_deg_global [num_items]: degrees of the GLOBAL graph (the item degrees
        need an all-reduce over ranks, done by the caller).  Values: d_u^-1/2 d_i^-1/2
        (selfcf.py:240-249 on the global operator)."""
        import numpy as np
        local_uid = np.asarray(local_uid, dtype=np.int64)
        iid = np.asarray(iid, dtype=np.int64)
        per, padded = shard_bounds(num_items, world)
        with np.errstate(divide="ignore"):
            du = np.power(np.asarray(user_deg, dtype=np.float32), np.float32(-0.5))
            di = np.power(np.asarray(item_deg_global, dtype=np.float32), np.float32(-0.5))
        du[np.isinf(du)] = 0.0
        di[np.isinf(di)] = 0.0
        val = (du[local_uid] * di[iid]).astype(np.float32)
        r_ui = graph_cls.from_coo(local_uid, iid, val, n_local_users, padded, device, coalesce=True, **kw)
        if torch.device(device).type == "cuda":
            r_iu = r_ui.t         # device transpose; keeps the nnz correspondence ShardedEdgeDrop needs
        else:
            r_iu = graph_cls.from_coo(iid, local_uid, val, padded, n_local_users, device, coalesce=True, **kw)
        return cls(r_ui, r_iu, n_local_users, num_items, per, rank, world, group)


class ShardedEdgeDrop:
    """One augmented view of a rank's blocks: every stored non-zero of the symmetric operator is
    dropped independently with probability `pe` (gcl.py:18-25 over the directed edge list,
    buir.py:300-309 over the non-zeros of the normalised adjacency, which also rescales the kept
    values by 1/(1-pe): `rescale=True`).  (u,i) and (i,u) are separate non-zeros with separate
    draws, so the masked operator is no longer symmetric: the backward pass runs on its transpose,
    i.e. each block's structure with the OTHER block's draws, addressed through the nnz
    correspondence of the device transpose.  Bitmaps come from gcr_edge_mask_bits (counter RNG:
    a function of (seed, rank, block, nnz id) only)."""

    def __init__(self, g: ShardedBipartiteGraph, pe: float, seed: int, rescale: bool = False):
        perm = getattr(g.r_iu, "perm_from_transpose", None)
        if perm is None:
            raise ValueError("ShardedEdgeDrop needs blocks built on the GPU (r_iu = r_ui.t)")
        nnz, dev = g.r_ui.nnz, g.r_ui.device
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(nnz, device=dev, dtype=perm.dtype)
        s_ui = (int(seed) * 0x9E3779B97F4A7C15 + 2 * g.rank + 1) & (2 ** 64 - 1)
        s_iu = (int(seed) * 0x9E3779B97F4A7C15 + 2 * g.rank + 2) & (2 ** 64 - 1)
        self.seeds = (s_ui, s_iu)
        self.ui_fwd = Fn.edge_mask_bits(nnz, pe, s_ui, dev)
        self.iu_fwd = Fn.edge_mask_bits(nnz, pe, s_iu, dev)
        self.ui_bwd = Fn.edge_mask_bits(nnz, pe, s_iu, dev, edge_id=inv)      # (R_iu . m_iu)^T on r_ui's structure
        self.iu_bwd = Fn.edge_mask_bits(nnz, pe, s_ui, dev, edge_id=perm)     # (R_ui . m_ui)^T on r_iu's structure
        self.val_scale = 1.0 / (1.0 - pe) if rescale else 1.0


def _all_gather(full, shard, group, async_op):
    if dist.get_backend(group) == "gloo":
        parts = list(full.chunk(dist.get_world_size(group)))
        return dist.all_gather(parts, shard, group=group, async_op=async_op)
    return dist.all_gather_into_tensor(full, shard, group=group, async_op=async_op)


def _reduce_scatter(shard, full, group, async_op):
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter: all-reduce + slice
        dist.all_reduce(full, group=group)
        w, r = dist.get_world_size(group), dist.get_rank(group)
        shard.copy_(full.chunk(w)[r])
        return None
    return dist.reduce_scatter_tensor(shard, full, group=group, async_op=async_op)


def sharded_propagate_raw(g: ShardedBipartiteGraph, x_user, x_item_shard, n_layers, scale, spmm=_hip_spmm,
                          overlap=True, masks=None, val_scale=1.0):
    """final_u [U_g, d], final_i_shard [I/world, d] = scale * sum_{k=0..K} (A^k x)  on the sharded
    operator; no autograd.  `overlap` issues the collectives asynchronously so that they run
    beside the other half of the SpMM.  `masks` = (keep bits of r_iu, keep bits of r_ui) and
    `val_scale` select an edge-dropped view (ShardedEdgeDrop)."""
    world = g.world
    kw_iu, kw_ui = {}, {}
    if masks is not None:
        kw_iu = dict(keep_bits=masks[0], val_scale=val_scale)
        kw_ui = dict(keep_bits=masks[1], val_scale=val_scale)
    d = x_user.shape[1]
    dev = x_user.device
    if n_layers == 0:
        return x_user * scale, x_item_shard * scale
    # Horner form z <- x + A z (z_0 = x): after K steps z = sum_{k<=K} A^k x; both halves of a step
    # read the OLD z, every step writes one array per side (no separate layer output + running sum)
    z_u, z_i = x_user, x_item_shard
    multi = _multi(world)
    pooled = multi and spmm is _hip_spmm              # injected (test) SpMMs allocate their own outputs
    if pooled and overlap and dev.type == "cuda":
        return _propagate_two_streams(g, x_user, x_item_shard, n_layers, scale, kw_iu, kw_ui)
    z_item_full = g.buffers.get("z_item_full", g.items_padded, d, dev) if multi else None
    part_buf = g.buffers.get("part_i", g.items_padded, d, dev) if pooled else None
    y_buf = g.buffers.get("y_i", g.items_per_rank, d, dev) if multi else None
    for k in range(n_layers):
        s = scale if k == n_layers - 1 else 1.0
        if multi:
            h_ag = _all_gather(z_item_full, z_i.contiguous(), g.group, overlap)
        else:
            z_item_full, h_ag = z_i, None
        if pooled:
            part_i, _ = spmm(g.r_iu, z_u, y_out=part_buf, **kw_iu)       # item side: local users only
        else:
            part_i, _ = spmm(g.r_iu, z_u, **kw_iu)
        if h_ag is not None:
            h_ag.wait()
        if multi:
            y_i = y_buf
            h_rs = _reduce_scatter(y_i, part_i, g.group, overlap)
        else:
            y_i, h_rs = part_i, None
        _, z_u_next = spmm(g.r_ui, z_item_full, acc_in=x_user, acc_scale=s, want_y=False, **kw_ui)
        if h_rs is not None:
            h_rs.wait()
        z_i = (x_item_shard + y_i) * s if s != 1.0 else x_item_shard + y_i
        z_u = z_u_next
    return z_u, z_i


def _propagate_two_streams(g, x_user, x_item_shard, n_layers, scale, kw_iu, kw_ui):
    """The overlapped schedule on two HIP streams with double-buffered staging.

    main stream:  P_i(k) = R_g^T z_u(k)  ->  [wait AG(k)]  z_u(k+1) = x_u + R_g Z_i(k)
    side stream:  AG(k): Z_i(k) <- shards z_i(k)  ->  [wait P_i(k)]  RS(k): y_i <- sum_ranks P_i(k)  ->  z_i(k+1) = x_i + y_i

    The item-shard chain AG -> RS -> update -> AG ... lives on the side stream, so the all-gather of layer k + 1 starts as
    soon as the reduce-scatter of layer k has landed — it no longer queues behind the user-side SpMM of layer k, which is
    what a single gathered buffer (and collectives issued from the compute stream) forced.  Two gathered tables and two
    partial-sum buffers alternate by layer parity; reuse distances, each ordered by an event that is already in the
    schedule: Z_i[p] is rewritten by AG(k + 2), issued behind RS(k + 1), which waited for P_i(k + 1), which follows the
    user-side SpMM(k) that read it; P_i[p] is rewritten by the item-side SpMM(k + 2), issued behind main's wait for
    AG(k + 1), which follows RS(k) that read it."""
    dev, d = x_user.device, x_user.shape[1]
    main = torch.cuda.current_stream(dev)
    side = g.comm_stream(dev)
    z_full = [g.buffers.get(f"z_item_full{p}", g.items_padded, d, dev) for p in (0, 1)]
    part = [g.buffers.get(f"part_i{p}", g.items_padded, d, dev) for p in (0, 1)]
    y_buf = g.buffers.get("y_i", g.items_per_rank, d, dev)
    z_u, z_i = x_user, x_item_shard.contiguous()
    side.wait_stream(main)                                  # the inputs (and the pooled buffers' last users) are done
    for k in range(n_layers):
        p = k & 1
        s = scale if k == n_layers - 1 else 1.0
        with torch.cuda.stream(side):
            _all_gather(z_full[p], z_i, g.group, False)
            ev_ag = torch.cuda.Event()
            ev_ag.record(side)
        _hip_spmm(g.r_iu, z_u, y_out=part[p], **kw_iu)                    # item side: local users only
        ev_part = torch.cuda.Event()
        ev_part.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev_part)
            _reduce_scatter(y_buf, part[p], g.group, False)
            z_i = (x_item_shard + y_buf) * s if s != 1.0 else x_item_shard + y_buf
        main.wait_event(ev_ag)
        _, z_u = _hip_spmm(g.r_ui, z_full[p], acc_in=x_user, acc_scale=s, want_y=False, **kw_ui)
    main.wait_stream(side)
    z_i.record_stream(main)                                 # allocated under the side stream, handed to the caller's
    return z_u, z_i


class _ShardedPropagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_user, x_item_shard, g, n_layers, scale, spmm, overlap, view):
        ctx.g, ctx.n_layers, ctx.scale, ctx.spmm, ctx.overlap, ctx.view = g, n_layers, scale, spmm, overlap, view
        masks, vs = ((view.iu_fwd, view.ui_fwd), view.val_scale) if view is not None else (None, 1.0)
        return sharded_propagate_raw(g, x_user.contiguous(), x_item_shard.contiguous(), n_layers, scale, spmm, overlap,
                                     masks, vs)

    @staticmethod
    def backward(ctx, g_user, g_item_shard):
        # linear operator: same schedule on the gradients with the transposed operator (itself when
        # unmasked: symmetric; an edge-dropped view swaps the two blocks' draws)
        view = ctx.view
        masks, vs = ((view.iu_bwd, view.ui_bwd), view.val_scale) if view is not None else (None, 1.0)
        gu, gi = sharded_propagate_raw(ctx.g, g_user.contiguous(), g_item_shard.contiguous(), ctx.n_layers,
                                       ctx.scale, ctx.spmm, ctx.overlap, masks, vs)
        return gu, gi, None, None, None, None, None, None


def sharded_lightgcn_propagate(g: ShardedBipartiteGraph, x_user, x_item_shard, n_layers, combine="mean",
                               spmm=_hip_spmm, overlap=True, view=None):
    """Differentiable sharded K-layer propagation; returns (user rows of this rank, item shard of
    this rank).  Use `gather_items` to replicate the item side for the loss.  `view`: a
    ShardedEdgeDrop (one augmented view of the graph)."""
    scale = 1.0 / (n_layers + 1) if combine == "mean" else 1.0
    return _ShardedPropagate.apply(x_user, x_item_shard, g, int(n_layers), scale, spmm, overlap, view)


class _GatherItems(torch.autograd.Function):
    """All-gather of the item shards; backward = reduce-scatter of the (partial) item gradients to
    their owner shard — 'a reduce-scatter of embedding gradients after the loss'."""

    @staticmethod
    def forward(ctx, shard, group):
        ctx.group = group
        w = dist.get_world_size(group)
        full = torch.empty(shard.shape[0] * w, shard.shape[1], dtype=shard.dtype, device=shard.device)
        _all_gather(full, shard.contiguous(), group, False)
        return full

    @staticmethod
    def backward(ctx, g_full):
        out = torch.empty(g_full.shape[0] // dist.get_world_size(ctx.group), g_full.shape[1], dtype=g_full.dtype,
                          device=g_full.device)
        _reduce_scatter(out, g_full.contiguous().clone(), ctx.group, False)
        return out, None


def gather_items(item_shard, group=None):
    if not dist.is_initialized() or not _multi(dist.get_world_size(group)):
        return item_shard
    return _GatherItems.apply(item_shard, group)


gather_rows = gather_items       # any row-sharded table: all-gather forward, reduce-scatter of the gradient backward


class _AllReduceSum(torch.autograd.Function):
    """Sum of per-rank partials into a REPLICATED tensor every rank goes on using for its share of the loss:
    the gradient of the total loss w.r.t. the replicated value is the sum of the ranks' partial gradients."""

    @staticmethod
    def forward(ctx, part, group):
        ctx.group = group
        out = part.contiguous().clone()
        dist.all_reduce(out, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


def all_reduce_sum(part, group=None):
    if not dist.is_initialized() or not _multi(dist.get_world_size(group)):
        return part
    return _AllReduceSum.apply(part, group)


def allreduce_replicated_grads(params, group=None):
    """Replicated parameters (MHCN's gating / attention weights, a replicated item table) receive a partial
    gradient on every rank (each rank back-propagates its own users' share of the loss): sum them."""
    if not dist.is_initialized() or not _multi(dist.get_world_size(group)):
        return
    for p in params:
        if p.grad is not None:
            dist.all_reduce(p.grad, group=group)


class _ShardedSymInfoNCE(torch.autograd.Function):
    """gcl.py:28-35 over row-sharded views on the HIP kernels, the single-process recipe of `functional._InfoNCEStats`
    (want_col) cut by rows: forward = two row-logsumexp launches (local anchors of one view against the all-gathered other
    view, one tile product each); backward = one launch per local table with the statistics of BOTH cross-entropies —
        d/d z1_i = sum_j ( w e^{s_ij - lse12_i} + w e^{s_ij - lse21_j} ) z2hat_j / tau - 2 w z2hat_i / tau,  w = 1 / (2 M)
    — which needs the other view's rows (gathered already) and its row logsumexps (an all-gather of [M] floats), so NO
    gradient flows back through the gathered tables: six tile products per step instead of the eight of two flash-forward /
    table-side-backward pairs, and no reduce-scatter of two [M, d] gradient tables."""

    @staticmethod
    def forward(ctx, z1, z2, inv_tau, group):
        multi = dist.is_initialized() and _multi(dist.get_world_size(group))
        world = dist.get_world_size(group) if multi else 1

        def gather(x):
            if not multi:
                return x
            full = torch.empty((x.shape[0] * world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            _all_gather(full, x.contiguous(), group, False)
            return full

        a, b = Fn._pad_dim(z1).contiguous(), Fn._pad_dim(z2).contiguous()
        sa, sb = Fn.row_inv_norm(a), Fn.row_inv_norm(b)
        a_f, b_f, sa_f, sb_f = gather(a), gather(b), gather(sa), gather(sb)
        eng = Fn._resolve_engine(unit_rows=True)
        lse12 = Fn.infonce_lse_raw(a, sa, b_f, sb_f, inv_tau, engine_flag=eng)
        lse21 = Fn.infonce_lse_raw(b, sb, a_f, sa_f, inv_tau, engine_flag=eng)
        m_local = a.shape[0]
        idx = torch.arange(m_local, device=a.device, dtype=torch.int64)
        pos = Fn.pos_logit_raw(a, sa, b, sb, idx, inv_tau)          # row i of both views lives on this rank
        ctx.save_for_backward(a, b, sa, sb, a_f, b_f, sa_f, sb_f, lse12, lse21, idx)
        ctx.inv_tau, ctx.eng, ctx.d, ctx.gather, ctx.m_total = inv_tau, eng, z1.shape[1], gather, m_local * world
        return ((lse12 - pos).sum() + (lse21 - pos).sum()) / (2 * ctx.m_total)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        a, b, sa, sb, a_f, b_f, sa_f, sb_f, lse12, lse21, idx = ctx.saved_tensors
        inv_tau, eng = ctx.inv_tau, ctx.eng
        m_local = a.shape[0]
        w = (g.float() / (2 * ctx.m_total)).reshape(1)
        w_local, w_full = w.expand(m_local).contiguous(), w.expand(ctx.m_total).contiguous()
        lse12_f, lse21_f = ctx.gather(lse12), ctx.gather(lse21)
        ga = Fn._infonce_bwd_raw(a, sa, b_f, sb_f, inv_tau, lse12, w_local, lse21_f, w_full, False, eng)
        gb = Fn._infonce_bwd_raw(b, sb, a_f, sa_f, inv_tau, lse21, w_local, lse12_f, w_full, False, eng)
        L = _lib.lib()
        stream = _lib.cur_stream(a.device)
        _lib.check(L.gcr_infonce_pos_bwd_f32(_lib.dptr(a), _lib.dptr(sa), _lib.dptr(b), _lib.dptr(sb), _lib.dptr(idx),
                                             _lib.dptr((-2.0 * w_local).contiguous()), m_local, m_local, a.shape[1],
                                             float(inv_tau), _lib.dptr(ga), _lib.dptr(gb), stream), "gcr_infonce_pos_bwd_f32")
        for x, sc, gx in ((a, sa, ga), (b, sb, gb)):
            _lib.check(L.gcr_normalize_bwd_f32(_lib.dptr(x), _lib.dptr(sc), _lib.dptr(gx), x.shape[0], x.shape[1],
                                               _lib.dptr(gx), stream), "gcr_normalize_bwd_f32")
        if ga.shape[1] != ctx.d:
            ga, gb = ga[:, :ctx.d].contiguous(), gb[:, :ctx.d].contiguous()
        return ga, gb, None, None


def sharded_info_nce_loss(z1_local, z2_local, temp=0.2, group=None, stats_fn=None):
    """gcl.py:28-35 over row-sharded views (BASELINE config 4): every rank holds the same row block of
    z1 and z2 ([M/world, d] each, equal sizes).  Both cross-entropies are row problems with LOCAL
    anchors against the ALL-GATHERED other view (CE(sim) from z1's rows, CE(sim.T) from z2's rows).
    Returns this rank's share of the loss: summing it over ranks (all-reduce) gives
    the single-process value; gradients are already the full-loss gradients of the local rows.
    On the HIP path (`_ShardedSymInfoNCE`) the backward uses both cross-entropies' row statistics in one launch per local
    table and nothing returns through the gathered tables.  `stats_fn` (the gloo choreography tests' CPU stand-in for
    `functional.infonce_stats`) selects the composition of two row problems instead, whose gathered tables' gradients
    return by reduce-scatter (`gather_items`)."""
    if stats_fn is None and z1_local.is_cuda:
        if z1_local.shape != z2_local.shape:
            raise ValueError("sharded_info_nce_loss needs the same row block of both views on every rank")
        return _ShardedSymInfoNCE.apply(z1_local, z2_local, 1.0 / float(temp), group)
    stats = stats_fn or Fn.infonce_stats
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    m_local = z1_local.shape[0]
    pos = torch.arange(m_local, device=z1_local.device, dtype=torch.int64) + rank * m_local
    z2_full, z1_full = gather_items(z2_local, group), gather_items(z1_local, group)
    lse12, pos12 = stats(z1_local, z2_full, pos, temp, True)[:2]
    lse21, pos21 = stats(z2_local, z1_full, pos, temp, True)[:2]
    m_total = m_local * world
    return ((lse12 - pos12).sum() + (lse21 - pos21).sum()) / (2 * m_total)


# ---------------------------------------------------------------------------------------------
# BASELINE config 5: MHCN's three U x U hypergraph channels, row-sharded by user (SURVEY §8e last sentence)
# ---------------------------------------------------------------------------------------------
def _hip_dual(graph, x_full):
    raw = torch.empty(graph.n_rows, x_full.shape[1], dtype=torch.float32, device=x_full.device)
    nrm = torch.empty_like(raw)
    inv = torch.empty(graph.n_rows, dtype=torch.float32, device=x_full.device)
    Fn.spmm_dual_into(graph, x_full, raw, nrm, inv)
    return raw, nrm, inv


def _hip_dual_acc(graph, x_full, acc):
    """(A x, acc + normalize(A x), 1 / |A x|) from one launch: the layer-list accumulation of mhcn.py:440-457 folded in."""
    raw = torch.empty(graph.n_rows, x_full.shape[1], dtype=torch.float32, device=x_full.device)
    out = torch.empty_like(raw)
    inv = torch.empty(graph.n_rows, dtype=torch.float32, device=x_full.device)
    Fn.spmm_dual_acc_into(graph, x_full, raw, acc.contiguous(), out, inv)
    return raw, out, inv


def _hip_spmm_t(graph, dz, out=None):
    gt = graph.t
    y = out if out is not None else torch.empty(gt.n_rows, dz.shape[1], dtype=torch.float32, device=dz.device)
    Fn.spmm_into(gt, dz.contiguous(), y=y)
    return y


class ShardedChannels:
    """Rank-local row blocks H_c[users_g, :] (CsrGraph [U_g, U_pad], columns = padded global user ids) of
    MHCN's channel operators H_s, H_j, H_p (univariate/mhcn.py:340-368).  Users are padded to a multiple of
    `world`; every rank owns `users_per_rank` rows."""

    def __init__(self, blocks, users_per_rank, rank, world, group=None):
        self.blocks = list(blocks)
        self.users_per_rank, self.rank, self.world, self.group = users_per_rank, rank, world, group
        self.users_padded = users_per_rank * world
        for b in self.blocks:
            if b.n_rows != users_per_rank or b.n_cols != self.users_padded:
                raise ValueError("channel block must be [users_per_rank, users_per_rank * world]")
        self.buffers = _Buffers()
        dev = self.blocks[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.blocks] if dev.type == "cuda" else None


class _ShardedChannelLayer(torch.autograd.Function):
    """One MHCN layer over the three channels (mhcn.py:440-448) on row-sharded operators:
        forward   all-gather X_c (its own [U, d] operand per channel) -> (raw_c, norm_c) = dual SpMM on H_c block
        backward  dZ_c = g_raw + normalize-backward(g_norm) -> partial = H_c block^T dZ_c [U, d] -> reduce-scatter
    The three all-gathers are issued back to back (async); channel c's SpMM starts as soon as ITS gather has
    landed, on its own HIP stream, so gather c+1 (and c+2) run beside SpMM c; the backward mirrors it with the
    reduce-scatter of channel c beside the transposed SpMM of channel c+1.
    With running sums `accs` (HIP path only) the second output of channel c is acc_c + norm_c from the same launch and the
    normalised copy is not kept (the backward rebuilds it from the raw rows: gcr_normalize_bwd_raw_f32)."""

    @staticmethod
    def forward(ctx, ch, dual_fn, spmm_t_fn, *tensors):
        n_c = len(ch.blocks)
        xs, accs = tensors[:n_c], tensors[n_c:]
        ctx.ch, ctx.spmm_t_fn = ch, spmm_t_fn
        ctx.with_acc = [a is not None for a in accs]
        world, dev, d = ch.world, xs[0].device, xs[0].shape[1]
        fulls, handles = [], []
        for c in range(n_c):
            if _multi(world):
                full = ch.buffers.get(f"x_full{c}", ch.users_padded, d, dev)
                handles.append(_all_gather(full, xs[c].contiguous(), ch.group, True))
            else:
                full = xs[c].contiguous()
                handles.append(None)
            fulls.append(full)
        outs, saved = [], []
        cur = torch.cuda.current_stream(dev) if ch.streams is not None else None
        for c in range(n_c):
            if ch.streams is not None:
                s = ch.streams[c]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    if handles[c] is not None:
                        handles[c].wait()                 # stream s waits for gather c only
                    raw, nrm, inv = _hip_dual_acc(ch.blocks[c], fulls[c], accs[c]) if ctx.with_acc[c] \
                        else dual_fn(ch.blocks[c], fulls[c])
                for t in (raw, nrm, inv):
                    t.record_stream(cur)
            else:
                if handles[c] is not None:
                    handles[c].wait()
                raw, nrm, inv = _hip_dual_acc(ch.blocks[c], fulls[c], accs[c]) if ctx.with_acc[c] \
                    else dual_fn(ch.blocks[c], fulls[c])
            outs += [raw, nrm]                           # (with an acc: nrm = acc + normalised rows)
            saved += [raw if ctx.with_acc[c] else nrm, inv]
        if ch.streams is not None:
            for s in ch.streams:
                cur.wait_stream(s)                       # join: the pooled gather buffers are free again, outputs ready
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)       # an unused output (the last layer's raw rows) arrives as None, not as zeros
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        ch = ctx.ch
        saved = ctx.saved_tensors
        world = ch.world
        n_c = len(ch.blocks)
        dxs, handles, parts = [], [], []
        daccs = [gs[2 * c + 1] if ctx.with_acc[c] else None for c in range(n_c)]   # acc + n: the sum's gradient passes through
        for c in range(n_c):
            g_raw, g_nrm = gs[2 * c], gs[2 * c + 1]
            nrm, inv = saved[2 * c], saved[2 * c + 1]
            dz = None
            if g_nrm is not None and nrm.is_cuda:
                dz = Fn.normalize_bwd_n(nrm, inv, g_nrm, g_raw, from_raw=ctx.with_acc[c])   # one pass
            elif g_nrm is not None:       # CPU stand-ins of the gloo choreography tests only
                dz = (g_nrm - nrm * (nrm * g_nrm).sum(1, keepdim=True)) * inv.unsqueeze(1)
                dz = dz if g_raw is None else dz + g_raw
            elif g_raw is not None:
                dz = g_raw
            if dz is None:
                dxs.append(None)
                handles.append(None)
                continue
            d = dz.shape[1]
            if _multi(world):
                buf = ch.buffers.get(f"g_part{c}", ch.users_padded, d, dz.device) if ctx.spmm_t_fn is _hip_spmm_t else None
                part = ctx.spmm_t_fn(ch.blocks[c], dz, buf) if buf is not None else ctx.spmm_t_fn(ch.blocks[c], dz)
                dx = torch.empty(ch.users_per_rank, d, dtype=torch.float32, device=dz.device)
                handles.append(_reduce_scatter(dx, part, ch.group, True))    # runs beside the next channel's SpMM
                parts.append(part)
            else:
                dx = ctx.spmm_t_fn(ch.blocks[c], dz)
                handles.append(None)
            dxs.append(dx)
        for h in handles:
            if h is not None:
                h.wait()
        return (None, None, None, *dxs, *daccs)


def sharded_channel_layer(ch: ShardedChannels, xs, dual_fn=_hip_dual, spmm_t_fn=_hip_spmm_t, accs=None):
    """[(raw_c, norm_c)] for the rank's user rows of every channel; xs: the rank's [U_g, d] operand rows.
    accs (HIP path: dual_fn is the default): one running sum per channel — the pairs become (raw_c, acc_c + norm_c)."""
    if accs is not None and dual_fn is not _hip_dual:
        raise ValueError("sharded_channel_layer: running sums need the HIP dual launch")
    accs = [None] * len(xs) if accs is None else list(accs)
    out = _ShardedChannelLayer.apply(ch, dual_fn, spmm_t_fn, *xs, *accs)
    return [(out[2 * c], out[2 * c + 1]) for c in range(len(xs))]
