"""Host-side mirror of the reference's data / encoder classes for the hot path, on top of the HIP
ops: same constructor arguments, attribute names and return values, so a reference script swaps
its class for this one and keeps its training loop.

  Interaction     ncl.py:46-88 (= directau.py:102-144, univariate/sept.py:109-152)
  LGCNEncoder     ncl.py:397-422 (= directau.py:269-293, selfcf.py:457-485 with normalised=True)
  LightGCN        lightgcn.py:12-27
  sept_encoder    univariate/sept.py:220-226
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import functional as Fn
from .graph import CsrGraph


def _device(device=None):
    return torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")


class Interaction:
    """ncl.py:46-88.  `train`/`test` are lists of (user, item, rating) with hashable raw ids.
    Dense ids follow sorted raw-id order (ncl.py:60-61).  Besides the reference's attributes
    (`user`, `item`, `id2user`, `id2item`, `user_num`, `item_num`, `training_data`,
    `training_set_u`, `test_set`) it carries the device-resident operator `norm_adj` (a CsrGraph:
    the raw 0/1 adjacency with duplicates kept, exactly what the reference calls norm_adj, Q1) and
    the sorted per-user training rows the device sampler rejects against."""

    def __init__(self, conf, train, test, device=None, normalised=False, id_order="sorted"):
        self.train, self.test = train, test
        self.device = _device(device)
        if id_order == "sorted":          # ncl.py:60-61, directau.py:116-117, sept.py:122-123
            users = sorted({t[0] for t in train})
            items = sorted({t[1] for t in train})
        elif id_order == "first_seen":    # selfcf.py:279-288, ssl4rec.py:69-75 (dict insertion order)
            users = list(dict.fromkeys(t[0] for t in train))
            items = list(dict.fromkeys(t[1] for t in train))
        else:
            raise ValueError("id_order must be 'sorted' or 'first_seen'")
        self.user = {u: k for k, u in enumerate(users)}
        self.item = {i: k for k, i in enumerate(items)}
        self.id2user = {k: u for u, k in self.user.items()}
        self.id2item = {k: i for i, k in self.item.items()}
        self.user_num, self.item_num = len(users), len(items)
        self.training_data = [(t[0], t[1]) for t in train]
        self.training_set_u = {u: set() for u in self.user}
        for t in train:
            self.training_set_u[t[0]].add(t[1])
        self.test_set = {}
        for t in test:
            self.test_set.setdefault(t[0], {})[t[1]] = 1
        self.uid = np.fromiter((self.user[t[0]] for t in train), dtype=np.int64, count=len(train))
        self.iid = np.fromiter((self.item[t[1]] for t in train), dtype=np.int64, count=len(train))
        build = CsrGraph.bipartite_sym_norm if normalised else CsrGraph.bipartite_raw
        self.norm_adj = build(self.uid, self.iid, self.user_num, self.item_num, self.device)
        # sorted, de-duplicated positives per user for the rejection sampler
        keys = np.unique(self.uid * max(self.item_num, 1) + self.iid)
        pu, pi = keys // max(self.item_num, 1), keys % max(self.item_num, 1)
        rowptr = np.zeros(self.user_num + 1, dtype=np.int64)
        np.cumsum(np.bincount(pu, minlength=self.user_num), out=rowptr[1:])
        self.user_rowptr = torch.from_numpy(rowptr).to(self.device)
        self.user_items_sorted = torch.from_numpy(pi.astype(np.int32)).to(self.device)
        self.uid_dev = torch.from_numpy(self.uid).to(self.device)
        self.iid_dev = torch.from_numpy(self.iid).to(self.device)

    def get_user_id(self, user):
        return self.user[user]

    def user_rated(self, user):
        return list(self.training_set_u[user]), []


class _StackedTable(torch.autograd.Function):
    """`torch.cat([user_emb, item_emb], 0)` (ncl.py:416) without the copy: the two parameters are the row blocks of ONE
    [U + I, d] buffer, so the concatenation is that buffer itself; the backward hands each parameter its block of the
    incoming gradient as a view (no split copy either)."""

    @staticmethod
    def forward(ctx, user_emb, item_emb, table):
        ctx.n_u = user_emb.shape[0]
        return table.view_as(table)

    @staticmethod
    def backward(ctx, g):
        return g[: ctx.n_u], g[ctx.n_u:], None


class LGCNEncoder(nn.Module):
    """ncl.py:397-422: `forward()` -> (user_emb [U, d], item_emb [I, d], all_emb list of K+1 [N, d]),
    final = mean of the K+1 layer outputs; K SpMMs with the mean fused into the epilogue.

    `embedding_dict["user_emb"]` / `["item_emb"]` are the reference's two parameters (same names, shapes, separate
    xavier_uniform_ initialisation); here they are views of one stacked [U + I, d] buffer (`self.table`), which is what
    the propagation reads — the reference concatenates them in every forward (ncl.py:416), a 282 MB copy at cfg3."""

    def __init__(self, data, emb_size, n_layers):
        super().__init__()
        self.data = data
        self.latent_size = emb_size
        self.layers = n_layers
        self.norm_adj = data.norm_adj
        init = nn.init.xavier_uniform_
        n_u, n_i = data.user_num, data.item_num
        self.table = torch.empty(n_u + n_i, emb_size, device=data.device)
        init(self.table[:n_u])
        init(self.table[n_u:])
        self.embedding_dict = nn.ParameterDict({
            "user_emb": nn.Parameter(self.table[:n_u]),
            "item_emb": nn.Parameter(self.table[n_u:]),
        })

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() re-create the parameters one by one: re-establish the stacked buffer afterwards
        super()._apply(fn, *args, **kwargs)
        u, i = self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]
        if self.table.device != u.device or self.table.dtype != u.dtype or u.data_ptr() != self.table.data_ptr():
            self.table = torch.cat([u.data, i.data], 0)
            u.data, i.data = self.table[: u.shape[0]], self.table[u.shape[0]:]
        return self

    def stacked(self):
        """The [U + I, d] embedding table as an autograd alias of the two parameters."""
        return _StackedTable.apply(self.embedding_dict["user_emb"], self.embedding_dict["item_emb"], self.table)

    def forward(self):
        emb = self.stacked()
        final, all_emb = Fn.lightgcn_propagate(self.norm_adj, emb, self.layers, combine="mean", return_layers=True)
        user_all, item_all = Fn.split_rows(final, self.data.user_num)
        return user_all, item_all, all_emb


class LightGCN(nn.Module):
    """lightgcn.py:12-27: `forward(edge_index)` -> (user_emb, item_emb) = SUM over layers 0..K of the
    LGConv propagation (no division, Q3).  The gcn_norm weights + CSR are prepared once per
    edge_index tensor and cached (the reference recomputes them in every LGConv call)."""

    def __init__(self, num_users, num_items, embedding_dim=64, num_layers=3):
        super().__init__()
        self.user_embedding = nn.Embedding(num_users, embedding_dim)
        self.item_embedding = nn.Embedding(num_items, embedding_dim)
        self.num_layers = num_layers
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)
        self._graph_key, self._graph = None, None

    def prepare(self, edge_index, symmetric=None):
        """Builds (or returns the cached) operator for `edge_index` int64 [2, nnz]."""
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version)
        if self._graph_key != key:
            n = self.user_embedding.num_embeddings + self.item_embedding.num_embeddings
            ei = edge_index.detach().cpu().numpy()
            if symmetric is None:  # lightgcn.py:38-39 always passes [[u; i+U], [i+U; u]]
                half = ei.shape[1] // 2
                symmetric = ei.shape[1] % 2 == 0 and np.array_equal(ei[0, :half], ei[1, half:]) and \
                    np.array_equal(ei[1, :half], ei[0, half:])
            self._graph = CsrGraph.from_edge_index_gcn_norm(ei, n, self.user_embedding.weight.device, symmetric=symmetric)
            self._graph_key = key
        return self._graph

    def forward(self, edge_index):
        graph = edge_index if isinstance(edge_index, CsrGraph) else self.prepare(edge_index)
        x = torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)
        x = Fn.lightgcn_propagate(graph, x, self.num_layers, combine="sum")
        nu = self.user_embedding.num_embeddings
        return Fn.split_rows(x, nu)


def sept_encoder(emb, adj: CsrGraph, n_layers: int, combine: str = "mean"):
    """emb_k = normalize(A emb_{k-1}) per layer (the NORMALISED rows feed the next layer), then
    combine='mean': univariate/sept.py:220-226; combine='sum': univariate/sept_social.py:370-385
    (`encoder` on the bipartite operator, `social_encoder` on the U x U social / sharing views)."""
    if combine not in ("mean", "sum"):
        raise ValueError("combine must be 'mean' or 'sum'")
    acc, e = emb, emb
    for _ in range(n_layers):
        e = Fn.spmm_l2norm(adj, e)
        acc = acc + e
    return acc / (n_layers + 1) if combine == "mean" else acc


def load_data(train_path, test_path, device=None):
    """lightgcn.py:30-40 / gcl.py:67-78: space-separated `user item rating` lines with integer ids;
    num_users / num_items = max id over train and test + 1; edge_index = [[u; i+U], [i+U; u]] int64
    [2, 2E].  Returns (edge_index, train (users, items), test (users, items), num_users, num_items);
    the reference returns pandas frames where this returns int64 arrays."""
    def read(path):
        rows = np.loadtxt(path, dtype=np.float64, ndmin=2)
        return rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)

    tu, ti = read(train_path)
    su, si = read(test_path)
    num_users = int(max(tu.max(), su.max())) + 1
    num_items = int(max(ti.max(), si.max())) + 1
    u = torch.from_numpy(tu)
    i = torch.from_numpy(ti) + num_users
    edge_index = torch.stack([torch.cat([u, i]), torch.cat([i, u])]).to(_device(device))
    return edge_index, (tu, ti), (su, si), num_users, num_items


class _MultiStreamSpMM(torch.autograd.Function):
    """The independent SpMMs of one layer as ONE autograd node: forward and backward fork onto the side streams and
    join on the caller's stream themselves, so autograd sees a node that lives on the caller's stream (per-operator
    nodes recorded under `torch.cuda.stream(s)` make the engine run each backward on its side stream and the
    parameters' AccumulateGrad then synchronises across streams — the warning GPUTEST r02 showed — and it is the
    form a hipGraph capture of the step needs)."""

    @staticmethod
    def forward(ctx, graphs, modes, streams, *xs):
        dev = xs[0].device
        cur = torch.cuda.current_stream(dev)
        xs = [x.contiguous() for x in xs]
        # outputs come from the caller's stream's pool; the join below orders every later reuse behind the side streams
        bufs = []
        for g, x, mode in zip(graphs, xs, modes):
            d = x.shape[1]
            y = torch.empty(g.n_rows, d, dtype=torch.float32, device=dev)
            raw = torch.empty_like(y) if mode == "dual" else None
            inv = torch.empty(g.n_rows, dtype=torch.float32, device=dev) if mode else None
            bufs.append((y, raw, inv))
        start = torch.cuda.Event()
        start.record(cur)
        for g, x, s, mode, (y, raw, inv) in zip(graphs, xs, streams, modes, bufs):
            s.wait_event(start)
            with torch.cuda.stream(s):
                if mode == "dual":
                    Fn.spmm_dual_into(g, x, raw, y, inv)
                else:
                    Fn.spmm_into(g, x, y=y, l2norm=bool(mode), inv_norm_out=inv)
        for s in streams[: len(graphs)]:
            cur.wait_stream(s)
        ctx.graphs, ctx.modes, ctx.streams = graphs, modes, streams
        ctx.layout = []
        saved, outs = [], []
        for mode, (y, raw, inv) in zip(modes, bufs):
            if mode == "dual":
                outs += [raw, y]
                saved += [y, inv]
            elif mode:
                outs.append(y)
                saved += [y, inv]
            else:
                outs.append(y)
        ctx.save_for_backward(*saved)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        graphs, modes, streams = ctx.graphs, ctx.modes, ctx.streams
        saved = list(ctx.saved_tensors)
        ref = next(g for g in gs if g is not None)
        dev = ref.device
        cur = torch.cuda.current_stream(dev)
        start = torch.cuda.Event()
        start.record(cur)
        dxs, gi, si = [], 0, 0
        used = []
        for g, s, mode in zip(graphs, streams, modes):
            if mode == "dual":
                g_raw, g_y = gs[gi], gs[gi + 1]
                gi += 2
            else:
                g_raw, g_y = (None, gs[gi]) if mode else (gs[gi], None)
                gi += 1
            y = inv = None
            if mode:
                y, inv = saved[si], saved[si + 1]
                si += 2
            if g_raw is None and g_y is None:
                dxs.append(None)
                continue
            gt = g.t
            dx = torch.empty(gt.n_rows, ref.shape[1], dtype=torch.float32, device=dev)
            s.wait_event(start)
            with torch.cuda.stream(s):
                dz = None
                if g_y is not None:       # through the row normalise: (g - y <y, g>) / max(||A x||, eps)
                    dz = (g_y - y * (y * g_y).sum(1, keepdim=True)) * inv.unsqueeze(1)
                if g_raw is not None:
                    dz = g_raw if dz is None else dz + g_raw
                dz = dz.contiguous()
                Fn.spmm_into(gt, dz, y=dx)
                dz.record_stream(s)
            used.append(s)
            dxs.append(dx)
        for s in used:
            cur.wait_stream(s)
        return (None, None, None, *dxs)


def multi_stream_spmm(graphs, xs, streams=None, l2norm=False):
    """BASELINE config 5 (univariate/mhcn.py:440-456): the per-layer SpMMs over independent operators
    (H_s, H_j, H_p, R^T, R) launched on separate HIP streams so that they fill the chip together and
    can hide each other's tails / a concurrent all-gather.  Returns the outputs in order; the caller's
    current stream waits for all of them (stream join) in the forward AND in the backward (one autograd node,
    `_MultiStreamSpMM`).

    l2norm: False -> A x (`spmm`); True -> normalize(A x) only (SEPT-style, `spmm_l2norm`);
    "dual" -> the pair (A x, normalize(A x)) per operator (`spmm_l2norm_dual`) — what MHCN's layer loop
    needs: mhcn.py:440-442 feeds the RAW product to the next layer and appends the normalised copy to
    the layer list.  A per-operator sequence of those values is accepted too."""
    graphs, xs = list(graphs), list(xs)
    if streams is None:
        streams = [torch.cuda.Stream() for _ in graphs]
    modes = list(l2norm) if isinstance(l2norm, (list, tuple)) else [l2norm] * len(graphs)
    for x in xs:
        if not x.is_cuda:
            raise RuntimeError("multi_stream_spmm operates on HIP device tensors only")
    flat = _MultiStreamSpMM.apply(graphs, modes, list(streams), *xs)
    outs, k = [], 0
    for mode in modes:
        if mode == "dual":
            outs.append((flat[k], flat[k + 1]))
            k += 2
        else:
            outs.append(flat[k])
            k += 1
    return outs
