"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/gcr.h declares (and the ctypes table binds exactly those), and the host-side SpMM
planner covers every non-zero exactly once.  No device compute here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from recommendation_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build()
    return _lib.lib()


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gcr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from recommendation_amd import _lib
    names = header_symbols()
    assert names, "no declarations parsed from include/gcr.h"
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gcr.h but not exported by libgcr.so"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.gcr_version() >= 100 and lib.gcr_arch() == b"gfx950"


def test_missing_library_fails_loudly(monkeypatch):
    from recommendation_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgcr.so")
    with pytest.raises(_lib.GcrError):
        _lib.lib()


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from recommendation_amd import CsrGraph, functional, _lib
    g = CsrGraph(np.array([0, 1, 2]), np.array([1, 0]), None, 2, 2, "cpu", symmetric=True)
    with pytest.raises(_lib.GcrError):
        functional.spmm(g, torch.ones(2, 64))


@pytest.mark.parametrize("L", [64, 256])
def test_spmm_plan_covers_every_nonzero_once(lib, L):
    from recommendation_amd.graph import SpmmPlan
    rng = np.random.default_rng(L)
    degs = rng.poisson(7, 5000)
    degs[rng.integers(0, 5000, 40)] = 0
    degs[[5, 1000, 4999]] = [10 * L + 3, L + 1, 3 * L]
    degs[[17, 18]] = [L, L]
    rowptr = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    plan = SpmmPlan(rowptr, "cpu", L)
    desc = plan.desc_host
    assert desc[0, 0] == 0 and desc[-1, 1] == rowptr[-1]
    assert np.array_equal(desc[1:, 0], desc[:-1, 1])            # contiguous cover of [0, nnz)
    assert (desc[:, 1] - desc[:, 0]).max() <= L
    row0, nrows, slot = desc[:, 2] & 0xFFFFFFFF, desc[:, 2] >> 32, desc[:, 3]
    whole = slot < 0
    assert nrows[whole].max() <= 64 and (nrows[~whole] == 1).all()
    # whole-row partitions: consecutive rows, exact nnz range, every short row exactly once
    seen = np.zeros(5000, dtype=np.int64)
    for r0, nr, a, b in zip(row0[whole], nrows[whole], desc[whole, 0], desc[whole, 1]):
        assert rowptr[r0] == a and rowptr[r0 + nr] == b
        seen[r0:r0 + nr] += 1
    long_rows = plan.long_row.numpy()[: plan.n_long]
    assert sorted(long_rows.tolist()) == sorted(np.nonzero(degs > L)[0].tolist())
    seen[long_rows] += 1
    assert (seen == 1).all()
    # split rows: slots are consecutive per row, chunks tile the row
    s0 = plan.long_slot0.numpy()
    for k, r in enumerate(long_rows):
        parts = np.nonzero((~whole) & (row0 == r))[0]
        assert slot[parts].tolist() == list(range(s0[k], s0[k + 1]))
        assert desc[parts[0], 0] == rowptr[r] and desc[parts[-1], 1] == rowptr[r + 1]
    assert plan.n_slots == s0[plan.n_long]


def test_plan_rejects_bad_rowptr(lib):
    from recommendation_amd import _lib
    from recommendation_amd.graph import SpmmPlan
    with pytest.raises(_lib.GcrError):
        SpmmPlan(np.array([0, 5, 3], dtype=np.int64), "cpu", 256)
