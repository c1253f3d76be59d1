"""Prepared sparse operators for the LightGCN message pass (device-resident CSR + work plan).

Mirrors what the reference builds once per model and then reuses every step:
  * raw 0/1 bipartite adjacency, duplicates kept       ncl.py:74-85 (= directau.py:130-141,
                                                        sept.py:134-145, mhcn.py:245-256)   [Q1]
  * D^-1/2 (R + R^T) D^-1/2, duplicates summed          selfcf.py:291-306 + 240-255, ssl4rec.py:79-88
  * gcn_norm(add_self_loops=False) edge weights         lightgcn.py:17,25 (torch_geometric LGConv)
Layout in HBM: rowptr int64[N+1], col int32[nnz], val fp32[nnz] (absent for the all-ones raw
adjacency), plus the SpMM plan (32-byte partition descriptors) and, for split rows, an fp32
partial-sum workspace [n_slots, d].
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

# interleaved A/B on MI355X (scripts/perf_spmm_ab.py): 64 -> 0.713 ms, 128 -> 0.652, 256 -> 0.629,
# 512 -> 0.617, 1024 -> 0.631 per cfg2 layer; cfg4 7.97 / 7.22 / 6.87 / 6.70 / 6.67 ms
DEFAULT_NNZ_PER_PART = 512


def _np_i64(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int64)


class SpmmPlan:
    """Host-built partition list for gcr_spmm_csr_f32 (include/gcr.h)."""

    def __init__(self, rowptr_host: np.ndarray, device, nnz_per_part=DEFAULT_NNZ_PER_PART, row_group=None):
        """row_group (optional int array [n_rows]): a locality group per row (reorder.locality_permutation's labels on the
        re-numbered operator).  The partitions are then laid out so that each group runs on ONE XCD, group after group
        (reorder.xcd_grouped_order), and its rows stay in that XCD's L2 between the gathers that share them; without it
        the partitions keep row order and consecutive workgroups alternate over the XCDs."""
        L = _lib.lib()
        rowptr_host = np.ascontiguousarray(rowptr_host, dtype=np.int64)
        n_rows = rowptr_host.size - 1
        sizes = (ctypes.c_int64 * 3)()
        p = ctypes.addressof(sizes)
        _lib.check(L.gcr_spmm_plan_size_host(rowptr_host.ctypes.data, n_rows, nnz_per_part, p, p + 8, p + 16),
                   "gcr_spmm_plan_size_host")
        self.n_parts, self.n_long, self.n_slots = int(sizes[0]), int(sizes[1]), int(sizes[2])
        desc = np.empty((max(self.n_parts, 1), 4), dtype=np.int64)
        long_row = np.zeros(max(self.n_long, 1), dtype=np.int32)
        long_slot0 = np.zeros(self.n_long + 1, dtype=np.int32)
        _lib.check(L.gcr_spmm_plan_fill_host(rowptr_host.ctypes.data, n_rows, nnz_per_part, desc.ctypes.data,
                                             long_row.ctypes.data, long_slot0.ctypes.data), "gcr_spmm_plan_fill_host")
        self.nnz_per_part = nnz_per_part
        self.grouped = row_group is not None and self.n_parts > 0
        if self.grouped:
            from .reorder import xcd_grouped_order
            rg = np.asarray(row_group.detach().cpu().numpy() if isinstance(row_group, torch.Tensor) else row_group)
            if rg.shape != (n_rows,):
                raise ValueError("row_group must have one entry per row")
            order = xcd_grouped_order(desc[: self.n_parts], rg)
            pad = np.array([0, 0, 0, -1], dtype=np.int64)                 # an empty whole-row partition: writes nothing
            desc = np.where((order >= 0)[:, None], desc[np.maximum(order, 0)], pad[None, :])
            self.n_parts = int(order.size)
        self.desc_host = desc[: self.n_parts]
        self.desc = torch.from_numpy(desc).to(device)
        self.long_row = torch.from_numpy(long_row).to(device)
        self.long_slot0 = torch.from_numpy(long_slot0).to(device)


class CsrGraph:
    """A sparse operator A [n_rows, n_cols] resident on one GPU, ready for `functional.spmm`."""

    def __init__(self, rowptr, col, val, n_rows, n_cols, device, symmetric=False,
                 nnz_per_part=DEFAULT_NNZ_PER_PART, validate=True, transpose=None, row_group=None):
        rowptr_host = _np_i64(rowptr)
        if rowptr_host.size != n_rows + 1:
            raise ValueError("rowptr must have n_rows + 1 entries")
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.nnz = int(rowptr_host[-1])
        self.device = torch.device(device)
        self.symmetric = bool(symmetric)
        self.rowptr_host = rowptr_host
        self.rowptr = torch.from_numpy(rowptr_host).to(self.device)
        self.col = torch.as_tensor(col).to(device=self.device, dtype=torch.int32).contiguous()
        self.val = None if val is None else torch.as_tensor(val).to(device=self.device, dtype=torch.float32).contiguous()
        if self.col.numel() != self.nnz or (self.val is not None and self.val.numel() != self.nnz):
            raise ValueError("col / val length must equal rowptr[-1]")
        self.plan = SpmmPlan(rowptr_host, self.device, nnz_per_part, row_group=row_group)
        self._t = self if symmetric else transpose
        self._workspaces = {}
        if validate and self.device.type == "cuda":
            self.validate()

    # -- checks ---------------------------------------------------------------------------
    def validate(self):
        """Device-side structural check before any kernel trusts the indices."""
        errs = torch.zeros(1, dtype=torch.int64, device=self.device)
        _lib.check(_lib.lib().gcr_csr_validate(_lib.dptr(self.rowptr), _lib.dptr(self.col), self.n_rows, self.n_cols,
                                               self.nnz, _lib.dptr(errs), _lib.cur_stream(self.device)), "gcr_csr_validate")
        n = int(errs.item())
        if n:
            raise ValueError(f"CSR graph is malformed: {n} structural errors (rowptr order / column range)")

    # -- helpers --------------------------------------------------------------------------
    def workspace(self, d: int):
        """fp32 [n_slots, d] partial-sum buffer for split rows, one per (d, stream)."""
        if self.plan.n_slots == 0:
            return None
        key = (d, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._workspaces.get(key)
        if ws is None:
            ws = torch.empty(self.plan.n_slots, d, dtype=torch.float32, device=self.device)
            self._workspaces[key] = ws
        return ws

    @property
    def t(self) -> "CsrGraph":
        """Transposed operator (for the backward pass dX = A^T dY); `self` when symmetric."""
        if self._t is None and self.device.type == "cuda":
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device),
                                           self.rowptr[1:] - self.rowptr[:-1])
            rp, c, v, perm = coo_to_csr_device(self.col.to(torch.int64), rows, self.val, self.n_cols, self.n_rows,
                                               self.device, want_perm=True)
            self._t = CsrGraph(rp, c, v, self.n_cols, self.n_rows, self.device, symmetric=False,
                               nnz_per_part=self.plan.nnz_per_part, validate=False, transpose=self)
            self._t.perm_from_transpose = perm     # non-zero e of A^T is non-zero perm[e] of A (shared edge masks)
        if self._t is None:
            col_host = self.col.cpu().numpy().astype(np.int64)
            rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(self.rowptr_host))
            val_host = None if self.val is None else self.val.cpu().numpy()
            rp, c, v, _ = _coo_to_csr_host(col_host, rows, val_host, self.n_cols)
            self._t = CsrGraph(rp, c, v, self.n_cols, self.n_rows, self.device, symmetric=False,
                               nnz_per_part=self.plan.nnz_per_part, validate=False, transpose=self)
        return self._t

    def row_degrees(self):
        """Non-zeros per row, float32 [n_rows] (cached)."""
        dg = getattr(self, "_row_deg", None)
        if dg is None:
            dg = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.float32)
            self._row_deg = dg
        return dg

    def with_values(self, val):
        """A view of this operator with other per-non-zero values (float32 [nnz], same order): structure, work plan and
        workspace are shared, nothing is copied — e.g. the per-edge coefficients of `functional.bpr_edge_sums`."""
        import copy
        if tuple(val.shape) != (self.nnz,) or val.dtype != torch.float32 or val.device != self.col.device:
            raise ValueError("val must be float32 [nnz] on the graph's device")
        g = copy.copy(self)
        g.val = val.contiguous()
        g._t = None
        return g

    def user_major_edges(self, n_users):
        """Bipartite symmetric operators ([U + I] x [U + I], users first; `bipartite_sym_norm`, `from_edge_index_gcn_norm`
        of lightgcn.py:36-39's edge_index): (u_idx, i_idx) int64 [E] of the non-zeros of the user rows in CSR order — the
        training pairs when the graph was built from them — cached."""
        c = getattr(self, "_um_edges", None)
        if c is None or c[0] != int(n_users):
            n_users = int(n_users)
            rp = self.rowptr[: n_users + 1]
            e = int(rp[-1])
            if 2 * e != self.nnz:
                raise ValueError("not a symmetric bipartite operator with the users first: nnz != 2 * nnz(user rows)")
            u = torch.repeat_interleave(torch.arange(n_users, device=self.device), rp[1:] - rp[:-1])
            i = self.col[:e].to(torch.int64) - n_users
            c = (n_users, u, i)
            self._um_edges = c
        return c[1], c[2]

    def negatives_user_block(self, n_users, n_items, n_neg=1):
        """The CSR skeleton [U x I] of `n_neg` negatives per training edge in user-major order: the user rows' pointer of this
        (symmetric bipartite) operator times n_neg, with its work plan — built once, cached; the caller fills `col` / `val`
        with a step's negatives and coefficients (functional.bpr_edge_sums' backward)."""
        key = (int(n_users), int(n_items), int(n_neg))
        cache = self.__dict__.setdefault("_neg_blocks", {})
        blk = cache.get(key)
        if blk is None:
            rp = self.rowptr_host[: n_users + 1] * int(n_neg)
            nnz = int(rp[-1])
            blk = CsrGraph(rp, torch.zeros(nnz, dtype=torch.int32, device=self.device),
                           torch.zeros(nnz, dtype=torch.float32, device=self.device), n_users, n_items, self.device,
                           symmetric=False, nnz_per_part=self.plan.nnz_per_part, validate=False)
            cache[key] = blk
        return blk

    def mirror_perm(self):
        """Symmetric operators only: int64 [nnz], mirror[e] = position of the non-zero (c, r) for the non-zero
        e = (r, c) — the nnz -> transpose-nnz map that lets a per-non-zero edge mask be handed to the
        backward pass in A^T's order (`functional.spmm(keep_bits_t=...)`).  Duplicate pairs (the raw
        multigraph of ncl.py:74-85) are matched copy by copy.  One-off index plumbing, cached."""
        if not self.symmetric:
            raise ValueError("mirror_perm() is for symmetric graphs; an asymmetric graph has graph.t.perm_from_transpose")
        m = getattr(self, "_mirror", None)
        if m is None:
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), self.rowptr[1:] - self.rowptr[:-1])
            cols = self.col.to(torch.int64)
            by_rc = torch.argsort(rows * self.n_cols + cols, stable=True)     # k-th copy of (r, c)
            by_cr = torch.argsort(cols * self.n_rows + rows, stable=True)     # k-th copy whose transpose is (r, c)
            m = torch.empty_like(by_rc)
            m[by_rc] = by_cr
            self._mirror = m
        return m

    # -- constructors ---------------------------------------------------------------------
    @classmethod
    def from_coo(cls, row, col, val, n_rows, n_cols, device, coalesce=False, symmetric=False, **kw):
        """Stable sort by row (COO order kept inside a row, duplicates kept — torch.sparse.mm on the
        uncoalesced COO of ncl.py:203-209 sums them), or (row, col)-sorted with duplicates summed
        when `coalesce` (scipy `tmp + tmp.T`, selfcf.py:297; `.coalesce()`, sept.py:50).
        On a GPU device the sort / merge runs in gcr_coo_to_csr; the numpy path only serves
        CPU-resident graphs (host logic, gloo tests)."""
        if torch.device(device).type == "cuda":
            rp, c, v, _ = coo_to_csr_device(row, col, val, n_rows, n_cols, device, coalesce=coalesce)
            return cls(rp, c, v, n_rows, n_cols, device, symmetric=symmetric, **kw)
        row, col = _np_i64(row), _np_i64(col)
        if val is not None and isinstance(val, torch.Tensor):
            val = val.detach().cpu().numpy()
        if coalesce:
            rp, c, v = _coalesce_host(row, col, val, n_rows, n_cols)
        else:
            rp, c, v, _ = _coo_to_csr_host(row, col, val, n_rows)
        return cls(rp, c, v, n_rows, n_cols, device, symmetric=symmetric, **kw)

    @classmethod
    def bipartite_raw(cls, uid, iid, num_users, num_items, device, **kw):
        """ncl.py:74-85: rows/cols [(u, i+U), (i+U, u)] per interaction, value 1, duplicates kept (Q1)."""
        n = num_users + num_items
        if torch.device(device).type == "cuda":
            u, i = _dev_i64(uid, device), _dev_i64(iid, device) + num_users
            row = torch.stack([u, i], 1).reshape(-1)      # (u, i+U) then (i+U, u) per interaction
            col = torch.stack([i, u], 1).reshape(-1)
            return cls.from_coo(row, col, None, n, n, device, symmetric=True, **kw)
        uid, iid = _np_i64(uid), _np_i64(iid) + num_users
        row = np.empty(2 * uid.size, dtype=np.int64)
        col = np.empty_like(row)
        row[0::2], row[1::2] = uid, iid
        col[0::2], col[1::2] = iid, uid
        return cls.from_coo(row, col, None, n, n, device, symmetric=True, **kw)

    @classmethod
    def bipartite_sym_norm(cls, uid, iid, num_users, num_items, device, **kw):
        """selfcf.py:291-306 + 240-255 (ssl4rec.py:79-88): A = R~ + R~^T with duplicate interactions
        summed, then D^-1/2 A D^-1/2 with 1/sqrt(0) -> 0, float32."""
        n = num_users + num_items
        if torch.device(device).type == "cuda":
            u, i = _dev_i64(uid, device), _dev_i64(iid, device) + num_users
            rp, c, v, _ = coo_to_csr_device(torch.cat([u, i]), torch.cat([i, u]), None, n, n, device, coalesce=True)
            return cls(rp, c, sym_norm_device(rp, c, v, n), n, n, device, symmetric=True, **kw)
        uid, iid = _np_i64(uid), _np_i64(iid) + num_users
        rp, c, v = _coalesce_host(np.concatenate([uid, iid]), np.concatenate([iid, uid]), None, n, n)
        rows = np.repeat(np.arange(n), np.diff(rp))
        rowsum = np.zeros(n, dtype=np.float32)
        np.add.at(rowsum, rows, v)
        with np.errstate(divide="ignore"):
            dinv = np.power(rowsum, np.float32(-0.5), dtype=np.float32)
        dinv[np.isinf(dinv)] = 0.0
        return cls(rp, c, (dinv[rows] * v * dinv[c]).astype(np.float32), n, n, device, symmetric=True, **kw)

    @classmethod
    def row_normalised(cls, row, col, val, n_rows, n_cols, device, **kw):
        """Graph.normalize_graph_mat on a NON-square matrix (selfcf.py:250-254 = ncl.py:37-41):
        D^-1 A with 1/0 -> 0, duplicates summed first (scipy CSR) — MHCN's R and H operators
        (univariate/mhcn.py:340-368,401-402).  GPU device only."""
        rp, c, v, _ = coo_to_csr_device(row, col, val, n_rows, n_cols, device, coalesce=True)
        rinv = torch.empty(n_rows, dtype=torch.float32, device=rp.device)
        out = torch.empty_like(v)
        _lib.check(_lib.lib().gcr_csr_row_norm_f32(_lib.dptr(rp), _lib.dptr(c), _lib.dptr(v), n_rows, _lib.dptr(rinv),
                                                   _lib.dptr(out), _lib.cur_stream(rp.device)), "gcr_csr_row_norm_f32")
        return cls(rp, c, out, n_rows, n_cols, device, symmetric=False, **kw)

    @classmethod
    def from_edge_index_gcn_norm(cls, edge_index, num_nodes, device, symmetric=False, **kw):
        """lightgcn.py:25 `LGConv()(x, edge_index)`: deg[v] = #edges with target v; w_e =
        deg^-1/2[src] deg^-1/2[dst] (inf -> 0); out[dst] += w_e x[src]  => CSR over rows = dst."""
        if torch.device(device).type == "cuda":
            ei = _dev_i64(edge_index, device)
            rp, c, _, _ = coo_to_csr_device(ei[1].contiguous(), ei[0].contiguous(), None, num_nodes, num_nodes, device)
            return cls(rp, c, sym_norm_device(rp, c, None, num_nodes), num_nodes, num_nodes, device,
                       symmetric=symmetric, **kw)
        ei = _np_i64(edge_index)
        src, dst = ei[0], ei[1]
        deg = np.bincount(dst, minlength=num_nodes).astype(np.float32)
        with np.errstate(divide="ignore"):
            dis = np.power(deg, np.float32(-0.5), dtype=np.float32)
        dis[np.isinf(dis)] = 0.0
        return cls.from_coo(dst, src, (dis[src] * dis[dst]).astype(np.float32), num_nodes, num_nodes, device,
                            symmetric=symmetric, **kw)


def _dev_i64(a, device):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    return torch.as_tensor(a).to(device=device, dtype=torch.int64).contiguous()


def coo_to_csr_device(row, col, val, n_rows, n_cols, device, coalesce=False, want_perm=False):
    """gcr_coo_to_csr: returns (rowptr int64 [n_rows+1], col int32 [nnz'], val fp32 [nnz'] or None,
    perm int64 [nnz'] or None) as device tensors.  Raises on ids outside the matrix."""
    L = _lib.lib()
    row, col = _dev_i64(row, device), _dev_i64(col, device)
    nnz = row.numel()
    if col.numel() != nnz:
        raise ValueError("row / col length mismatch")
    if val is not None:
        val = torch.as_tensor(val).to(device=device, dtype=torch.float32).contiguous()
    rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=device)
    col_out = torch.empty(max(nnz, 1), dtype=torch.int32, device=device)
    keep_val = val is not None or coalesce
    val_out = torch.empty(max(nnz, 1), dtype=torch.float32, device=device) if keep_val else None
    perm = torch.empty(max(nnz, 1), dtype=torch.int64, device=device) if (want_perm and not coalesce) else None
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    ws = torch.empty(int(L.gcr_coo_to_csr_workspace_bytes(nnz)), dtype=torch.uint8, device=device)
    _lib.check(L.gcr_coo_to_csr(_lib.dptr(row), _lib.dptr(col), _lib.dptr(val), nnz, n_rows, n_cols, int(coalesce),
                                _lib.dptr(rowptr), _lib.dptr(col_out), _lib.dptr(val_out), _lib.dptr(perm),
                                _lib.dptr(meta[0:1]), _lib.dptr(meta[1:2]), _lib.dptr(ws), _lib.cur_stream(device)),
               "gcr_coo_to_csr")
    n_out, n_err = (int(v) for v in meta.tolist())
    if n_err:
        raise ValueError(f"{n_err} COO entries have a row / column outside [{n_rows}, {n_cols}]")
    return (rowptr, col_out[:n_out].contiguous(), None if val_out is None else val_out[:n_out].contiguous(),
            None if perm is None else perm[:n_out].contiguous())


def sym_norm_device(rowptr, col, val, n):
    """gcr_csr_sym_norm_f32 on a square symmetric operator: D^-1/2 A D^-1/2 (val None = ones)."""
    dinv = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    out = torch.empty(col.numel(), dtype=torch.float32, device=rowptr.device)
    _lib.check(_lib.lib().gcr_csr_sym_norm_f32(_lib.dptr(rowptr), _lib.dptr(col), _lib.dptr(val), n, n, None, None,
                                               _lib.dptr(dinv), None, _lib.dptr(out), _lib.cur_stream(rowptr.device)),
               "gcr_csr_sym_norm_f32")
    return out


def _coo_to_csr_host(row, col, val, n_rows):
    order = np.argsort(row, kind="stable")
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=n_rows), out=rowptr[1:])
    v = None if val is None else np.asarray(val, dtype=np.float32)[order]
    return rowptr, col[order].astype(np.int32), v, order


def _coalesce_host(row, col, val, n_rows, n_cols):
    key = row * np.int64(n_cols) + col
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    first = np.ones(key_s.size, dtype=bool)
    first[1:] = key_s[1:] != key_s[:-1]
    seg = np.cumsum(first) - 1
    v_in = np.ones(row.size, dtype=np.float32) if val is None else np.asarray(val, dtype=np.float32)
    v = np.zeros(int(first.sum()), dtype=np.float32)
    np.add.at(v, seg, v_in[order])
    r, c = row[order][first], col[order][first]
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n_rows), out=rowptr[1:])
    return rowptr, c.astype(np.int32), v
