"""TEST INFRASTRUCTURE ONLY — ctypes front-end of oracle/oracle.c (liboracle.so).

Importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Never the product.
"""
from __future__ import annotations

import ctypes
import os
import shutil
import subprocess
from ctypes import c_double, c_float, c_int, c_int64, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle.c")
LIB = os.path.join(HERE, "liboracle.so")
_lib = None


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which("gcc")
    if gcc is None:
        raise RuntimeError("gcc not found: cannot build the C oracle")
    # no -march=native: the .so is built in one container and timed on another host
    cmd = [gcc, "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", SRC, "-o", LIB + ".tmp", "-lm"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + res.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = ctypes.CDLL(LIB)
        P = c_void_p
        L.orc_num_threads.restype = c_int
        L.orc_spmm_csr_f32.argtypes = [P, P, P, c_int64, P, c_int, P]
        L.orc_spmm_coo_f32.argtypes = [P, P, P, c_int64, c_int64, P, c_int, P]
        L.orc_lightgcn_propagate_f32.argtypes = [P, P, P, c_int64, P, c_int, c_int, c_float, P, P]
        L.orc_row_lse_f32.argtypes = [P, c_int64, P, c_int64, c_int, P, c_float, c_int, P, P, P]
        L.orc_bpr_loss_f32.argtypes = [P, P, c_int, P, P, P, c_int64, c_int, c_int]
        L.orc_bpr_loss_f32.restype = c_double
        L.orc_neg_sample.argtypes = [P, P, P, c_int64, c_int, c_int64, c_uint64, c_uint64, c_int, P]
        L.orc_edge_keep_mask.argtypes = [c_int64, c_float, c_uint64, P]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


def num_threads():
    return int(lib().orc_num_threads())


def spmm_csr(rowptr, col, val, x):
    rowptr, col, val, x = _c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32), _c(x, np.float32)
    y = np.empty((rowptr.size - 1, x.shape[1]), dtype=np.float32)
    lib().orc_spmm_csr_f32(_p(rowptr), _p(col), _p(val), rowptr.size - 1, _p(x), x.shape[1], _p(y))
    return y


def spmm_coo(row, col, val, n_rows, x):
    row, col, val, x = _c(row, np.int64), _c(col, np.int64), _c(val, np.float32), _c(x, np.float32)
    y = np.empty((n_rows, x.shape[1]), dtype=np.float32)
    lib().orc_spmm_coo_f32(_p(row), _p(col), _p(val), row.size, n_rows, _p(x), x.shape[1], _p(y))
    return y


def lightgcn_propagate(rowptr, col, val, x0, n_layers, combine="mean"):
    rowptr, col, val, x0 = _c(rowptr, np.int64), _c(col, np.int32), _c(val, np.float32), _c(x0, np.float32)
    n, d = x0.shape
    out = np.empty_like(x0)
    work = np.empty((2 * n, d), dtype=np.float32)
    scale = 1.0 / (n_layers + 1) if combine == "mean" else 1.0
    lib().orc_lightgcn_propagate_f32(_p(rowptr), _p(col), _p(val), n, _p(x0), d, n_layers, scale, _p(out), _p(work))
    return out


def row_lse(a, b, pos, inv_tau, normalize=True):
    a, b = _c(a, np.float32), _c(b, np.float32)
    pos = _c(pos, np.int64)
    m, d = a.shape
    n = b.shape[0]
    scratch = np.empty(((m + n), d), dtype=np.float32) if normalize else None
    lse = np.empty(m, dtype=np.float64)
    pl = np.zeros(m, dtype=np.float64)
    lib().orc_row_lse_f32(_p(a), m, _p(b), n, d, _p(pos), inv_tau, int(normalize), _p(scratch), _p(lse), _p(pl))
    return lse, pl


def bpr_loss(user_tab, item_tab, u, i, j, variant):
    user_tab, item_tab = _c(user_tab, np.float32), _c(item_tab, np.float32)
    u, i, j = _c(u, np.int64), _c(i, np.int64), _c(j, np.int64)
    n_neg = 1 if j.ndim == 1 else j.shape[1]
    return float(lib().orc_bpr_loss_f32(_p(user_tab), _p(item_tab), user_tab.shape[1], _p(u), _p(i), _p(j), u.size,
                                        n_neg, variant))


def neg_sample(user_rowptr, user_items_sorted, u_idx, n_negs, num_items, seed, offset, max_trials):
    rp, it, u = _c(user_rowptr, np.int64), _c(user_items_sorted, np.int32), _c(u_idx, np.int64)
    out = np.empty(u.size * n_negs, dtype=np.int64)
    lib().orc_neg_sample(_p(rp), _p(it), _p(u), u.size, n_negs, num_items, seed, offset, max_trials, _p(out))
    return out


def edge_keep_mask(nnz, pe, seed):
    keep = np.empty(nnz, dtype=np.uint8)
    lib().orc_edge_keep_mask(nnz, pe, seed, _p(keep))
    return keep.astype(bool)
