"""gloo tests (world sizes 2, 4 and 8) of the row-sharded propagation choreography (distributed.py) on CPU
tensors: the per-rank SpMM is injected from the CPU oracle (tests may use the oracle; the
product default is the HIP kernel), so what is checked here is the partition, the collectives
and the autograd wiring:  N-rank result == 1-rank result on the same synthetic graph."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_np as O

N_USERS, N_ITEMS, N_EDGES, D, K = 120, 53, 1500, 16, 3   # 53 items: not divisible by 2 -> padding path


def oracle_spmm(graph, x, acc_in=None, acc_scale=1.0, want_y=True):
    y = O.spmm_csr(graph.rowptr_host, graph.col.numpy(), graph.val.numpy(), x.detach().numpy())
    y = torch.from_numpy(y.astype(np.float32))
    acc = None if acc_in is None else (acc_in + y) * acc_scale
    return (y if want_y else None), acc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommendation_amd import distributed as gd
        u, i = O.synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=1)
        per_u = N_USERS // world
        lo, hi = rank * per_u, (rank + 1) * per_u
        sel = (u >= lo) & (u < hi)
        deg_u = np.bincount(u, minlength=N_USERS)[lo:hi]
        deg_i_local = torch.from_numpy(np.bincount(i[sel], minlength=N_ITEMS))
        dist.all_reduce(deg_i_local)                       # global item degrees, as the bench does
        g = gd.ShardedBipartiteGraph.from_local_interactions(u[sel] - lo, i[sel], per_u, N_ITEMS, deg_u,
                                                             deg_i_local.numpy(), rank, world, "cpu", validate=False)
        rng = np.random.default_rng(0)
        x_all = rng.standard_normal((N_USERS + g.items_padded, D)).astype(np.float32)
        x_all[N_USERS + N_ITEMS:] = 0                     # padding items
        w_all = rng.standard_normal((N_USERS + g.items_padded, D)).astype(np.float32)
        ipr = g.items_per_rank
        xu = torch.from_numpy(x_all[lo:hi]).requires_grad_(True)
        xi = torch.from_numpy(x_all[N_USERS + rank * ipr: N_USERS + (rank + 1) * ipr]).requires_grad_(True)
        fu, fi = gd.sharded_lightgcn_propagate(g, xu, xi, K, combine="mean", spmm=oracle_spmm, overlap=True)
        items_full = gd.gather_items(fi)                   # replicate for the loss; backward = reduce-scatter
        wu = torch.from_numpy(w_all[lo:hi])
        wi = torch.from_numpy(w_all[N_USERS:])
        # every rank holds the full item table, so weight the item term by 1/world to make the
        # summed per-rank losses equal the single-process loss
        loss = (fu * wu).sum() + (items_full * wi).sum() / world
        loss.backward()
        out[rank] = dict(fu=fu.detach().numpy(), fi=fi.detach().numpy(), gu=xu.grad.numpy(), gi=xi.grad.numpy(),
                         lo=lo, hi=hi, ipr=ipr, pad=g.items_padded)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_propagation_equals_single_process(world):
    from recommendation_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = {r: out[r] for r in range(world)}
    pad = res[0]["pad"]
    u, i = O.synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=1)
    rowptr, col, val = O.norm_adj_csr(u, i, N_USERS, pad)          # global operator incl. padding items
    rng = np.random.default_rng(0)
    x_all = rng.standard_normal((N_USERS + pad, D)).astype(np.float32)
    x_all[N_USERS + N_ITEMS:] = 0
    w_all = rng.standard_normal((N_USERS + pad, D)).astype(np.float32)
    ref, _ = O.lgcn_encoder_forward(rowptr, col, val, x_all, K, combine="mean")
    gacc, g = w_all.astype(np.float64), w_all.astype(np.float64)
    for _ in range(K):
        g = O.spmm_backward(rowptr, col, val, g, N_USERS + pad)
        gacc = gacc + g
    gref = gacc / (K + 1)
    tol = dict(rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    for r in range(world):
        lo, hi, ipr = res[r]["lo"], res[r]["hi"], res[r]["ipr"]
        np.testing.assert_allclose(res[r]["fu"], ref[lo:hi], **tol)
        np.testing.assert_allclose(res[r]["fi"], ref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **tol)
        gt = dict(rtol=2e-5, atol=2e-5 * np.abs(gref).max())
        np.testing.assert_allclose(res[r]["gu"], gref[lo:hi], **gt)
        np.testing.assert_allclose(res[r]["gi"], gref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **gt)


def _dense_stats(a, b, pos, temp, normalize=True):
    """torch CPU stand-in for functional.infonce_stats (dense; only to exercise the sharding)."""
    import torch.nn.functional as F
    an, bn = (F.normalize(a, dim=1), F.normalize(b, dim=1)) if normalize else (a, b)
    s = an @ bn.T / temp
    return torch.logsumexp(s, 1), s[torch.arange(a.shape[0]), pos]


def _nce_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommendation_amd import distributed as gd
        g = torch.Generator().manual_seed(0)
        z1 = torch.randn(64, 16, generator=g)
        z2 = z1 + 0.4 * torch.randn(64, 16, generator=g)
        m = 64 // world
        a = z1[rank * m:(rank + 1) * m].clone().requires_grad_(True)
        b = z2[rank * m:(rank + 1) * m].clone().requires_grad_(True)
        share = gd.sharded_info_nce_loss(a, b, 0.2, stats_fn=_dense_stats)
        share.backward()
        total = share.detach().clone()
        dist.all_reduce(total)
        out[rank] = dict(loss=float(total), ga=a.grad.numpy(), gb=b.grad.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_info_nce_loss_equals_single_process(world):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_nce_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = {r: out[r] for r in range(world)}
    g = torch.Generator().manual_seed(0)
    z1 = torch.randn(64, 16, generator=g)
    z2 = z1 + 0.4 * torch.randn(64, 16, generator=g)
    ref = O.info_nce_loss(z1.numpy(), z2.numpy(), 0.2)
    assert all(res[r]["loss"] == pytest.approx(ref, rel=1e-5) for r in range(world))
    w = np.full(64, 0.5 / 64)
    g1, g2 = O.infonce_grads(z1.numpy(), z2.numpy(), np.arange(64), 5.0, True, w, w)
    m = 64 // world
    for r in range(world):
        np.testing.assert_allclose(res[r]["ga"], g1[r * m:(r + 1) * m], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(res[r]["gb"], g2[r * m:(r + 1) * m], rtol=1e-4, atol=1e-6)


# --------------------------------------------------------------------------- BASELINE config 5 (MHCN channels)
class _CpuGraph:
    """Stand-in operator for the gloo choreography test: a dense float32 matrix with the CsrGraph attributes
    the sharded encoder reads.  The SpMM arithmetic itself is covered by the GPU tests."""

    def __init__(self, row, col, val, n_rows, n_cols):
        m = torch.zeros(n_rows, n_cols)
        m.index_put_((torch.from_numpy(row), torch.from_numpy(col)), torch.from_numpy(val), accumulate=True)
        self.m, self.n_rows, self.n_cols, self.device = m, n_rows, n_cols, torch.device("cpu")

    @property
    def t(self):
        g = _CpuGraph.__new__(_CpuGraph)
        g.m, g.n_rows, g.n_cols, g.device = self.m.T.contiguous(), self.n_cols, self.n_rows, self.device
        return g


class _CpuOps:
    @staticmethod
    def spmm(graph, x):
        return graph.m @ x

    @staticmethod
    def dual(graph, x):
        z = graph.m @ x
        return z, torch.nn.functional.normalize(z, p=2, dim=1)

    @staticmethod
    def channel_dual(graph, x_full):
        z = graph.m @ x_full
        nrm = z.norm(dim=1).clamp_min(1e-12)
        return z, z / nrm.unsqueeze(1), 1.0 / nrm

    @staticmethod
    def channel_spmm_t(graph, dz):
        return graph.m.T @ dz


def _mhcn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mhcn_sharded_common as C
        out[rank] = C.run_rank(rank, world, torch.device("cpu"), _CpuGraph, _CpuOps)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_mhcn_channels_equal_single_process(world):
    """Config 5: users row-sharded over the ranks, one all-gather per channel operand, reduce-scatter of the
    channel gradients, all-reduce of the item-side partial sums: values and gradients == single process."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mhcn_sharded_common as C
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_mhcn_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = {r: out[r] for r in range(world)}
    C.check(res, world, 2e-5)
