"""Sparse-matrix algebra on the device for the one-off operator construction around the hot path, first of all
MHCN's motif adjacency (univariate/mhcn.py:340-368 `build_hyper_adj_mats`).

A `Sp` is a coalesced CSR on the GPU (rowptr int64, col int32 ascending inside a row, val float32, no stored
zeros).  Products are expand-sort-compress: gcr_spgemm_expand_f32 emits one COO entry per (A non-zero, B row
entry) pair, gcr_coo_to_csr(coalesce = 1) sorts them and sums duplicates (rocPRIM radix sort + reduce-by-key);
element-wise products with a sparse mask are a binary search per entry (gcr_csr_lookup_f32); sums are a
concatenation + the same coalescing sort.  torch only concatenates, prefix-sums and filters index vectors.
The motif counts are small integers, exact in float32, so the STRUCTURE of the result is bit-exact with scipy's
(tests/golden/mhcn.npz holds the reference's own H_s / H_j / H_p).
"""
from __future__ import annotations

import torch

from . import _lib
from .graph import CsrGraph, coo_to_csr_device


class Sp:
    """Coalesced CSR [n_rows, n_cols] on the device."""

    def __init__(self, rowptr, col, val, n_rows, n_cols):
        self.rowptr, self.col, self.val, self.n_rows, self.n_cols = rowptr, col, val, int(n_rows), int(n_cols)

    @property
    def nnz(self):
        return int(self.col.numel())

    @property
    def device(self):
        return self.rowptr.device

    def row_of(self):
        r = getattr(self, "_row_of", None)
        if r is None:
            r = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device, dtype=torch.int32),
                                        self.rowptr[1:] - self.rowptr[:-1])
            self._row_of = r
        return r

    @classmethod
    def from_coo(cls, row, col, val, n_rows, n_cols, device):
        """Sorted by (row, col), duplicates summed, exact zeros dropped (what scipy's sparse arithmetic stores)."""
        rp, c, v, _ = coo_to_csr_device(row, col, val, n_rows, n_cols, device, coalesce=True)
        out = cls(rp, c, v, n_rows, n_cols)
        return out._drop_zeros()

    def _drop_zeros(self):
        keep = self.val != 0
        if bool(keep.all()):
            return self
        rows = self.row_of()[keep].to(torch.int64)
        rp = torch.zeros(self.n_rows + 1, dtype=torch.int64, device=self.device)
        rp[1:] = torch.cumsum(torch.bincount(rows, minlength=self.n_rows), 0)
        return Sp(rp, self.col[keep].contiguous(), self.val[keep].contiguous(), self.n_rows, self.n_cols)

    # -- algebra ----------------------------------------------------------------------------------
    @property
    def T(self):
        return Sp.from_coo(self.col.to(torch.int64), self.row_of().to(torch.int64), self.val, self.n_cols, self.n_rows,
                           self.device)

    def __matmul__(self, other):
        """A @ B (scipy `.dot`)."""
        if self.n_cols != other.n_rows:
            raise ValueError("shape mismatch")
        L = _lib.lib()
        dev = self.device
        if self.nnz == 0 or other.nnz == 0:
            return Sp.from_coo(torch.zeros(0, dtype=torch.int64, device=dev), torch.zeros(0, dtype=torch.int64, device=dev),
                               torch.zeros(0, device=dev), self.n_rows, other.n_cols, dev)
        blen = (other.rowptr[1:] - other.rowptr[:-1])[self.col.to(torch.int64)]
        offset = torch.zeros(self.nnz + 1, dtype=torch.int64, device=dev)
        offset[1:] = torch.cumsum(blen, 0)
        total = int(offset[-1])
        out_row = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        out_col = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
        out_val = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
        _lib.check(L.gcr_spgemm_expand_f32(_lib.dptr(self.rowptr), _lib.dptr(self.col), _lib.dptr(self.val), self.n_rows, self.nnz,
                                           _lib.dptr(self.row_of()), _lib.dptr(other.rowptr), _lib.dptr(other.col),
                                           _lib.dptr(other.val), _lib.dptr(offset), _lib.dptr(out_row), _lib.dptr(out_col),
                                           _lib.dptr(out_val), _lib.cur_stream(dev)), "gcr_spgemm_expand_f32")
        return Sp.from_coo(out_row[:total], out_col[:total], out_val[:total], self.n_rows, other.n_cols, dev)

    def __mul__(self, mask):
        """Element-wise product with another sparse matrix (scipy `.multiply`)."""
        if (self.n_rows, self.n_cols) != (mask.n_rows, mask.n_cols):
            raise ValueError("shape mismatch")
        if self.nnz == 0 or mask.nnz == 0:
            return Sp(torch.zeros(self.n_rows + 1, dtype=torch.int64, device=self.device), self.col[:0], self.val[:0],
                      self.n_rows, self.n_cols)
        m = torch.empty(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().gcr_csr_lookup_f32(_lib.dptr(self.row_of()), _lib.dptr(self.col), self.nnz, _lib.dptr(mask.rowptr),
                                                 _lib.dptr(mask.col), _lib.dptr(mask.val), _lib.dptr(m),
                                                 _lib.cur_stream(self.device)), "gcr_csr_lookup_f32")
        return Sp(self.rowptr, self.col, self.val * m[: self.nnz], self.n_rows, self.n_cols)._drop_zeros()

    def __add__(self, other):
        return add([self, other])

    def __sub__(self, other):
        return add([self, other], [1.0, -1.0])

    def greater(self, thr):
        """Entries > thr kept with their values (`H.multiply(H > thr)`)."""
        return Sp(self.rowptr, self.col, torch.where(self.val > thr, self.val, torch.zeros_like(self.val)), self.n_rows,
                  self.n_cols)._drop_zeros()

    def to_dense(self):
        out = torch.zeros(self.n_rows, self.n_cols, device=self.device)
        out[self.row_of().to(torch.int64), self.col.to(torch.int64)] = self.val
        return out

    def row_normalised_graph(self, **kw) -> CsrGraph:
        """`H.multiply(1.0 / H.sum(axis=1))` (mhcn.py:361-367) as a ready operator (gcr_csr_row_norm_f32)."""
        return CsrGraph.row_normalised(self.row_of().to(torch.int64), self.col.to(torch.int64), self.val, self.n_rows,
                                       self.n_cols, self.device, **kw)


def add(mats, signs=None):
    signs = signs or [1.0] * len(mats)
    dev = mats[0].device
    rows = torch.cat([m.row_of().to(torch.int64) for m in mats])
    cols = torch.cat([m.col.to(torch.int64) for m in mats])
    vals = torch.cat([m.val * s for m, s in zip(mats, signs)])
    return Sp.from_coo(rows, cols, vals, mats[0].n_rows, mats[0].n_cols, dev)


def motif_adjacency(s_row, s_col, y_row, y_col, n_users, n_items, device):
    """univariate/mhcn.py:340-368 before the row normalisation: (A1 + ... + A7, A8 + A9, A10 * (A10 > 3)) from the
    directed social pairs S and the interaction pairs Y (unit entries; a repeated pair sums, exactly as the
    `sp.csr_matrix((entries, (row, col)))` of mhcn.py:115-122,258-262 does)."""
    ones = lambda n: torch.ones(n, device=device)      # noqa: E731
    s_row, s_col = torch.as_tensor(s_row, device=device), torch.as_tensor(s_col, device=device)
    y_row, y_col = torch.as_tensor(y_row, device=device), torch.as_tensor(y_col, device=device)
    S = Sp.from_coo(s_row, s_col, ones(s_row.numel()), n_users, n_users, device)
    Y = Sp.from_coo(y_row, y_col, ones(y_row.numel()), n_users, n_items, device)
    B = S * S.T
    U = S - B
    UT, UU, BU, UB, BB = U.T, U @ U, B @ U, U @ B, B @ B
    UUT, UTU = U @ UT, UT @ U
    C1 = UU * UT
    A1 = C1 + C1.T
    C2 = add([BU * UT, UB * UT, UU * B])
    A2 = C2 + C2.T
    C3 = add([BB * U, BU * B, UB * B])
    A3 = C3 + C3.T
    A4 = BB * B
    C5 = add([UU * U, UUT * U, UTU * U])
    A5 = C5 + C5.T
    A6 = add([UB * U, (B @ UT) * UT, UTU * B])
    A7 = add([(UT @ B) * UT, BU * U, UUT * B])
    YY = Y @ Y.T
    A8 = YY * B
    A9 = YY * U
    A9 = A9 + A9.T
    A10 = add([YY, A8, A9], [1.0, -1.0, -1.0])
    return add([A1, A2, A3, A4, A5, A6, A7]), A8 + A9, A10.greater(3.0), Y


def build_hyper_graphs(s_row, s_col, y_row, y_col, n_users, n_items, device):
    """(H_s, H_j, H_p, R): the row-normalised channel operators of MHCN and the row-normalised interaction
    matrix (mhcn.py:340-368,401) as CsrGraph handles for mhcn.MHCNEncoder."""
    hs, hj, hp, Y = motif_adjacency(s_row, s_col, y_row, y_col, n_users, n_items, device)
    return hs.row_normalised_graph(), hj.row_normalised_graph(), hp.row_normalised_graph(), Y.row_normalised_graph()
