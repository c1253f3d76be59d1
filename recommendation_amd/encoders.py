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


def encode_raw_ids(values):
    """Raw ids (a column of the `user item rating` records) -> uint64 keys [n, W] whose lexicographic unsigned order is
    the order Python's `sorted()` gives the ids (ncl.py:60-61): integers by value, strings by code point (= by UTF-8
    byte, packed big-endian eight bytes per word, NUL-padded: a prefix sorts first).  None: ids of another type (the
    caller then keeps the host dict path)."""
    if not (isinstance(values, np.ndarray) and values.dtype.kind in "iuUS"):
        # decided from the WHOLE column: np.asarray would silently turn a mixed int / str column into strings (ints then
        # sort as text), where the reference's sorted() raises a TypeError — such a column keeps the host dict path
        kinds = set(map(type, values))
        if not kinds or any(issubclass(k, bool) or isinstance(k, type(None)) for k in kinds):
            return None
        is_int = all(issubclass(k, (int, np.integer)) for k in kinds)
        is_str = all(issubclass(k, (str, np.str_)) for k in kinds)
        if not (is_int or is_str):
            return None
    arr = np.asarray(values)
    if arr.ndim != 1 or arr.shape[0] != len(values):
        return None
    if arr.dtype.kind in "iu":
        if arr.dtype.kind == "u" and arr.size and int(arr.max()) >= 1 << 63:
            return None                                     # would wrap in int64: host dict path
        return (arr.astype(np.int64).view(np.uint64) ^ np.uint64(1 << 63)).reshape(-1, 1)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind != "S":
        return None
    n, w = arr.size, max(arr.dtype.itemsize, 1)
    words = (w + 7) // 8
    buf = np.zeros((n, words * 8), dtype=np.uint8)
    buf[:, :w] = np.frombuffer(arr.tobytes(), dtype=np.uint8).reshape(n, w)
    return buf.view(">u8").astype(np.uint64)


def dense_ids_device(keys, device, order="sorted"):
    """(dense int64 [n] on the device, first_pos int64 [m] on the host) of the records' uint64 keys [n, W]:
    gcr_dense_ids_u64; ids wider than one word are folded word by word through their sorted ranks."""
    from . import _lib
    L = _lib.lib()
    n, words = keys.shape
    dev = torch.device(device)
    ws = torch.empty(int(L.gcr_dense_ids_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    n_unique = torch.zeros(1, dtype=torch.int64, device=dev)

    def call(k, mode):
        dense = torch.empty(n, dtype=torch.int64, device=dev)
        first = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
        _lib.check(L.gcr_dense_ids_u64(_lib.dptr(k), n, mode, _lib.dptr(dense), _lib.dptr(first), _lib.dptr(n_unique),
                                       _lib.dptr(ws), _lib.cur_stream(dev)), "gcr_dense_ids_u64")
        return dense, first

    mode = {"sorted": 0, "first_seen": 1}[order]
    rank = None
    for w in range(words):
        kw = torch.from_numpy(np.ascontiguousarray(keys[:, w]).view(np.int64)).to(dev)
        if words == 1:
            rank = kw
            break
        rw, _ = call(kw, 0)
        rank = rw if rank is None else call(rank * n + rw, 0)[0]
    dense, first = call(rank, mode)
    return dense, first[: int(n_unique.item())].cpu().numpy()


class _LazyMaps:
    """The Python-side views of the id maps (`user`, `item`, `id2user`, `id2item`, `training_data`, `training_set_u`:
    ncl.py:50-72), built from the device result only when something reads them — the training step never does."""

    def __getattr__(self, name):
        if name == "user":
            val = {u: k for k, u in enumerate(self._users_raw)}
        elif name == "item":
            val = {i: k for k, i in enumerate(self._items_raw)}
        elif name == "id2user":
            val = dict(enumerate(self._users_raw))
        elif name == "id2item":
            val = dict(enumerate(self._items_raw))
        elif name == "training_data":
            val = [(t[0], t[1]) for t in self.train]
        elif name == "training_set_u":
            val = {u: set() for u in self._users_raw}
            for t in self.train:
                val[t[0]].add(t[1])
        else:
            raise AttributeError(name)
        setattr(self, name, val)
        return val


class Interaction(_LazyMaps):
    """ncl.py:46-88.  `train`/`test` are lists of (user, item, rating) with hashable raw ids.
    Dense ids follow sorted raw-id order (ncl.py:60-61) or first appearance (selfcf.py:281-288).  Besides the reference's
    attributes (`user`, `item`, `id2user`, `id2item`, `user_num`, `item_num`, `training_data`,
    `training_set_u`, `test_set`) it carries the device-resident operator `norm_adj` (a CsrGraph:
    the raw 0/1 adjacency with duplicates kept, exactly what the reference calls norm_adj, Q1) and
    the sorted per-user training rows the device sampler rejects against.

    On a GPU device the id maps are built there (gcr_dense_ids_u64: radix sort + head flags + scan over the encoded id
    columns) — the Python dicts of the reference become lazily built views.
    reorder="spectral": the dense ids are additionally re-numbered for gather locality (reorder.py); `perm_user` /
    `perm_item` hold reference id -> id used here, and every id this object hands out is the re-numbered one.  With
    `reorder_guard` (default) the renumbering skips itself — `perm_user` stays None, `reorder_decision` says why — when
    the graph fits the caches, its spectrum shows no community direction, or the renumbered operator measures no faster."""

    def __init__(self, conf, train, test, device=None, normalised=False, id_order="sorted", reorder=None,
                 rows_per_cluster=None, reorder_guard=True):
        self.train, self.test = train, test
        self.device = _device(device)
        if id_order not in ("sorted", "first_seen"):
            raise ValueError("id_order must be 'sorted' or 'first_seen'")
        cols = list(zip(*train)) if len(train) else [(), ()]
        ku = encode_raw_ids(cols[0]) if self.device.type == "cuda" and len(train) else None
        ki = encode_raw_ids(cols[1]) if ku is not None else None
        if ku is not None and ki is not None:
            uid, first_u = dense_ids_device(ku, self.device, id_order)
            iid, first_i = dense_ids_device(ki, self.device, id_order)
            self._users_raw = [cols[0][p] for p in first_u]
            self._items_raw = [cols[1][p] for p in first_i]
        else:
            if id_order == "sorted":          # ncl.py:60-61, directau.py:116-117, sept.py:122-123
                self._users_raw = sorted({t[0] for t in train})
                self._items_raw = sorted({t[1] for t in train})
            else:                             # selfcf.py:279-288, ssl4rec.py:69-75 (dict insertion order)
                self._users_raw = list(dict.fromkeys(t[0] for t in train))
                self._items_raw = list(dict.fromkeys(t[1] for t in train))
            user, item = self.user, self.item
            uid = torch.from_numpy(np.fromiter((user[t[0]] for t in train), dtype=np.int64, count=len(train))).to(self.device)
            iid = torch.from_numpy(np.fromiter((item[t[1]] for t in train), dtype=np.int64, count=len(train))).to(self.device)
        self.user_num, self.item_num = len(self._users_raw), len(self._items_raw)
        self.perm_user = self.perm_item = None
        row_group = None
        if reorder is not None:
            if reorder != "spectral" or self.device.type != "cuda":
                raise ValueError("reorder must be None or 'spectral' (GPU device)")
            from .reorder import DEFAULT_ROWS_PER_CLUSTER, guarded_locality_permutation, locality_permutation
            rpc = rows_per_cluster or DEFAULT_ROWS_PER_CLUSTER
            if reorder_guard:
                # the renumbering must never slow a graph down: it skips itself (reason in `reorder_decision`, logged) on
                # a graph that fits the caches, whose spectrum shows no community direction, or that measures no faster
                pu, pi, row_group, self.reorder_decision = guarded_locality_permutation(
                    uid, iid, self.user_num, self.item_num, self.device, rpc)
                import logging
                logging.getLogger("recommendation_amd").info("reorder='spectral': %s (%s)", "applied" if pu is not None
                                                             else "skipped", self.reorder_decision["reason"])
            else:
                pu, pi, row_group = locality_permutation(uid, iid, self.user_num, self.item_num, self.device, rpc)
                self.reorder_decision = {"applied": True, "reason": "reorder_guard=False"}
            if pu is not None:
                uid, iid = pu[uid], pi[iid]
                self.perm_user, self.perm_item = pu.cpu().numpy(), pi.cpu().numpy()
                self._users_raw = [u for _, u in sorted(zip(self.perm_user.tolist(), self._users_raw))]
                self._items_raw = [i for _, i in sorted(zip(self.perm_item.tolist(), self._items_raw))]
        self.test_set = {}
        for t in test:
            self.test_set.setdefault(t[0], {})[t[1]] = 1
        self.uid_dev, self.iid_dev = uid.contiguous(), iid.contiguous()
        self.uid, self.iid = self.uid_dev.cpu().numpy(), self.iid_dev.cpu().numpy()
        build = CsrGraph.bipartite_sym_norm if normalised else CsrGraph.bipartite_raw
        self.norm_adj = build(self.uid_dev if self.device.type == "cuda" else self.uid,
                              self.iid_dev if self.device.type == "cuda" else self.iid,
                              self.user_num, self.item_num, self.device, row_group=row_group)
        # sorted, de-duplicated positives per user for the rejection sampler
        ni = max(self.item_num, 1)
        keys = torch.unique(self.uid_dev * ni + self.iid_dev)
        pu_, pi_ = keys // ni, keys % ni
        rowptr = torch.zeros(self.user_num + 1, dtype=torch.int64, device=self.device)
        rowptr[1:] = torch.cumsum(torch.bincount(pu_, minlength=self.user_num), 0)
        self.user_rowptr = rowptr
        self.user_items_sorted = pi_.to(torch.int32).contiguous()

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

    def _aliased(self):
        u, i = self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]
        t = self.table
        return t.device == u.device and t.dtype == u.dtype and t.shape[0] == u.shape[0] + i.shape[0] and \
            u.data_ptr() == t.data_ptr() and i.data_ptr() == t.data_ptr() + u.numel() * t.element_size() and \
            u.is_contiguous() and i.is_contiguous()

    def restack(self):
        """Re-establishes `table` as the storage of the two parameters when something replaced theirs — `.to()` / `.cuda()`,
        `copy.deepcopy` (a target-encoder copy, bgrl_g2l.py:548-style), `load_state_dict(assign=True)`, a bare
        `param.data = ...`: the parameters win (they are what the optimiser and the state dict see), the table is rebuilt
        from them and they become views of it again.  Cheap when nothing changed (two pointer comparisons)."""
        if not self._aliased():
            u, i = self.embedding_dict["user_emb"], self.embedding_dict["item_emb"]
            with torch.no_grad():
                self.table = torch.cat([u.data, i.data], 0)
                u.data, i.data = self.table[: u.shape[0]], self.table[u.shape[0]:]
        return self.table

    def _apply(self, fn, *args, **kwargs):
        # .to() / .cuda() / .float() re-create the parameters one by one: re-establish the stacked buffer afterwards
        super()._apply(fn, *args, **kwargs)
        self.restack()
        return self

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            # graph handles are shared, not copied (device-resident CSR + plan; immutable for the encoder)
            new.__dict__[k] = v if k in ("data", "norm_adj") else copy.deepcopy(v, memo)
        new.restack()
        return new

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.restack()
        return out

    def stacked(self):
        """The [U + I, d] embedding table as an autograd alias of the two parameters (re-stacked first if the parameters'
        storage was replaced behind the table's back: a stale table would train nothing, silently)."""
        return _StackedTable.apply(self.embedding_dict["user_emb"], self.embedding_dict["item_emb"], self.restack())

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


    def loss(self, edge_index, neg_i=None, loss_type="bpr", reg_weight=0.0):
        """One training step's loss as lightgcn.py:85-118 computes it on the FULL batch of training edges
        (pos_u / pos_i = the edge list the graph was built from, in the graph's user-major order):
        loss_type "bpr": -log(sigmoid(pos - neg)).mean() with `neg_i` [E] or [E, n_neg] (lightgcn.py:98-108);
        loss_type "bce": BCE-with-logits of the [E, I] scores against one-hot labels (lightgcn.py:109-113; neg_i unused);
        plus reg_weight * (|user_vecs|^2 + |pos_item_vecs|^2) (lightgcn.py:118)."""
        graph = edge_index if isinstance(edge_index, CsrGraph) else self.prepare(edge_index)
        nu = self.user_embedding.num_embeddings
        x = torch.cat([self.user_embedding.weight, self.item_embedding.weight], dim=0)
        final = Fn.lightgcn_propagate(graph, x, self.num_layers, combine="sum")
        n_edges = graph.user_major_edges(nu)[0].numel()
        if loss_type == "bpr":
            if neg_i is None:
                raise ValueError("loss_type 'bpr' needs neg_i")
            s = Fn.bpr_edge_sums(graph, final, nu, neg_i, Fn.BPR_LOG_SIGMOID)
            return s[0] / n_edges + reg_weight * (s[1] + s[2])
        if loss_type != "bce":
            raise ValueError("Unsupported loss_type")              # lightgcn.py:115
        loss = Fn.bce_edge_loss(graph, final, nu)
        if reg_weight:
            # |user_vecs|^2 + |pos_item_vecs|^2 over the edge list = sum over rows of (#edges of the row) |row|^2
            loss = loss + reg_weight * torch.dot(graph.row_degrees(), final.square().sum(1))
        return loss


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
    form a hipGraph capture of the step needs).
    Modes per operator: False (A x), True (normalize(A x)), "dual" ((A x, normalize(A x))), "dual_acc" ((A x, acc + normalize(A
    x)) with the running sum `acc` as a second input: the layer-list accumulation of mhcn.py:440-457 in the same launch; the
    normalised copy is not kept, the backward rebuilds it from A x and 1 / |A x|)."""

    @staticmethod
    def forward(ctx, graphs, modes, streams, *tensors):
        n = len(graphs)
        xs, accs = list(tensors[:n]), list(tensors[n:])
        dev = xs[0].device
        cur = torch.cuda.current_stream(dev)
        xs = [x.contiguous() for x in xs]
        accs = [None if a is None else a.contiguous() for a in accs]
        # outputs come from the caller's stream's pool; the join below orders every later reuse behind the side streams
        bufs = []
        for g, x, mode in zip(graphs, xs, modes):
            d = x.shape[1]
            y = torch.empty(g.n_rows, d, dtype=torch.float32, device=dev)
            raw = torch.empty_like(y) if mode in ("dual", "dual_acc") else None
            inv = torch.empty(g.n_rows, dtype=torch.float32, device=dev) if mode else None
            bufs.append((y, raw, inv))
        start = torch.cuda.Event()
        start.record(cur)
        for g, x, a, s, mode, (y, raw, inv) in zip(graphs, xs, accs, streams, modes, bufs):
            s.wait_event(start)
            with torch.cuda.stream(s):
                if mode == "dual_acc":
                    Fn.spmm_dual_acc_into(g, x, raw, a, y, inv)          # y = acc + normalize(A x)
                elif mode == "dual":
                    Fn.spmm_dual_into(g, x, raw, y, inv)
                else:
                    Fn.spmm_into(g, x, y=y, l2norm=bool(mode), inv_norm_out=inv)
        for s in streams[: len(graphs)]:
            cur.wait_stream(s)
        ctx.graphs, ctx.modes, ctx.streams = graphs, modes, streams
        saved, outs = [], []
        for mode, (y, raw, inv) in zip(modes, bufs):
            if mode == "dual_acc":
                outs += [raw, y]
                saved += [raw, inv]
            elif mode == "dual":
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
        dxs, daccs, gi, si = [], [], 0, 0
        used = []
        for g, s, mode in zip(graphs, streams, modes):
            if mode in ("dual", "dual_acc"):
                g_raw, g_y = gs[gi], gs[gi + 1]
                gi += 2
            else:
                g_raw, g_y = (None, gs[gi]) if mode else (gs[gi], None)
                gi += 1
            y = inv = None
            if mode:
                y, inv = saved[si], saved[si + 1]                 # (dual_acc: y is the RAW product)
                si += 2
            daccs.append(g_y if mode == "dual_acc" else None)    # acc + n: the running sum's gradient passes through
            if g_raw is None and g_y is None:
                dxs.append(None)
                continue
            gt = g.t
            dx = torch.empty(gt.n_rows, ref.shape[1], dtype=torch.float32, device=dev)
            s.wait_event(start)
            with torch.cuda.stream(s):
                # through the row normalise, one pass: (g - y <y, g>) / max(||A x||, eps) + g_raw
                dz = Fn.normalize_bwd_n(y, inv, g_y, g_raw, from_raw=(mode == "dual_acc")) if g_y is not None \
                    else g_raw.contiguous()
                Fn.spmm_into(gt, dz, y=dx)
                dz.record_stream(s)
            used.append(s)
            dxs.append(dx)
        for s in used:
            cur.wait_stream(s)
        return (None, None, None, *dxs, *daccs)


def multi_stream_spmm(graphs, xs, streams=None, l2norm=False, acc=None):
    """BASELINE config 5 (univariate/mhcn.py:440-456): the per-layer SpMMs over independent operators
    (H_s, H_j, H_p, R^T, R) launched on separate HIP streams so that they fill the chip together and
    can hide each other's tails / a concurrent all-gather.  Returns the outputs in order; the caller's
    current stream waits for all of them (stream join) in the forward AND in the backward (one autograd node,
    `_MultiStreamSpMM`).

    l2norm: False -> A x (`spmm`); True -> normalize(A x) only (SEPT-style, `spmm_l2norm`);
    "dual" -> the pair (A x, normalize(A x)) per operator (`spmm_l2norm_dual`) — what MHCN's layer loop
    needs: mhcn.py:440-442 feeds the RAW product to the next layer and appends the normalised copy to
    the layer list.  A per-operator sequence of those values is accepted too.
    acc (with "dual"): one running sum [n_rows, d] per operator — the pair becomes (A x, acc + normalize(A x)), the
    layer-list sum of mhcn.py:458-466 folded into the launch."""
    graphs, xs = list(graphs), list(xs)
    if streams is None:
        streams = [torch.cuda.Stream() for _ in graphs]
    modes = list(l2norm) if isinstance(l2norm, (list, tuple)) else [l2norm] * len(graphs)
    accs = [None] * len(graphs) if acc is None else list(acc)
    if len(accs) != len(graphs):
        raise ValueError("acc: one running sum per operator")
    for k, a in enumerate(accs):
        if a is not None:
            if modes[k] != "dual":
                raise ValueError("acc needs l2norm='dual'")
            modes[k] = "dual_acc"
    for x in xs:
        if not x.is_cuda:
            raise RuntimeError("multi_stream_spmm operates on HIP device tensors only")
    flat = _MultiStreamSpMM.apply(graphs, modes, list(streams), *xs, *accs)
    outs, k = [], 0
    for mode in modes:
        if mode in ("dual", "dual_acc"):
            outs.append((flat[k], flat[k + 1]))
            k += 2
        else:
            outs.append(flat[k])
            k += 1
    return outs
