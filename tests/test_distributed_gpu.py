"""Two ranks SHARING the one GPU of the test box (gloo carries the collectives; RCCL refuses two
ranks on one device) run the row-sharded propagation and the sharded symmetric InfoNCE with the
real HIP kernels — the product `spmm` / `infonce_stats` defaults, nothing injected — and must
reproduce the single-process oracle result, values and gradients.  The N-GPU RCCL run itself is
the driver's (bench.py --gpus N); this is its rehearsal with real kernels at world_size 2."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

N_USERS, N_ITEMS, N_EDGES, D, K = 3000, 1111, 60000, 64, 3      # 1111 items: padding shard path
PE = 0.3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _inputs(pad):
    rng = np.random.default_rng(0)
    x_all = rng.standard_normal((N_USERS + pad, D)).astype(np.float32)
    x_all[N_USERS + N_ITEMS:] = 0
    w_all = rng.standard_normal((N_USERS + pad, D)).astype(np.float32)
    return x_all, w_all


def _worker(rank, world, port, out, backend="gloo", overlap=True):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # gloo: both ranks share cuda:0 (the one-GPU test box; RCCL refuses two ranks on one device);
    # nccl: one rank per GPU over RCCL — the production configuration, run wherever >= 2 GPUs are visible
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recommendation_amd import distributed as gd
        u, i = O.synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=1)
        per_u = N_USERS // world
        lo, hi = rank * per_u, (rank + 1) * per_u
        sel = (u >= lo) & (u < hi)
        deg_u = np.bincount(u, minlength=N_USERS)[lo:hi]
        deg_i = torch.from_numpy(np.bincount(i[sel], minlength=N_ITEMS)).to(dev)       # RCCL reduces device tensors
        dist.all_reduce(deg_i)
        g = gd.ShardedBipartiteGraph.from_local_interactions(u[sel] - lo, i[sel], per_u, N_ITEMS, deg_u, deg_i.cpu().numpy(),
                                                             rank, world, dev)
        x_all, w_all = _inputs(g.items_padded)
        ipr = g.items_per_rank
        xu = torch.from_numpy(x_all[lo:hi]).to(dev).requires_grad_(True)
        xi = torch.from_numpy(x_all[N_USERS + rank * ipr: N_USERS + (rank + 1) * ipr]).to(dev).requires_grad_(True)
        fu, fi = gd.sharded_lightgcn_propagate(g, xu, xi, K, combine="mean", overlap=overlap)
        items_full = gd.gather_items(fi)
        wu = torch.from_numpy(w_all[lo:hi]).to(dev)
        wi = torch.from_numpy(w_all[N_USERS:]).to(dev)
        ((fu * wu).sum() + (items_full * wi).sum() / world).backward()
        res = dict(fu=fu.detach().cpu().numpy(), fi=fi.detach().cpu().numpy(), gu=xu.grad.cpu().numpy(),
                   gi=xi.grad.cpu().numpy(), lo=lo, hi=hi, ipr=ipr, pad=g.items_padded)
        # an edge-dropped view (buir-style: independent draws per stored non-zero, kept values / (1 - pe))
        view = gd.ShardedEdgeDrop(g, PE, seed=11, rescale=True)
        xu2, xi2 = xu.detach().clone().requires_grad_(True), xi.detach().clone().requires_grad_(True)
        mu, mi = gd.sharded_lightgcn_propagate(g, xu2, xi2, K, combine="mean", view=view, overlap=overlap)
        ((mu * wu).sum() + (gd.gather_items(mi) * wi).sum() / world).backward()
        host = lambda b: (b.rowptr_host, b.col.cpu().numpy(), b.val.cpu().numpy())
        res.update(mu=mu.detach().cpu().numpy(), mi=mi.detach().cpu().numpy(), mgu=xu2.grad.cpu().numpy(),
                   mgi=xi2.grad.cpu().numpy(), seeds=view.seeds, ui=host(g.r_ui), iu=host(g.r_iu))
        # sharded symmetric InfoNCE on this rank's user rows of two noisy views
        gen = torch.Generator().manual_seed(5)
        z1 = torch.randn(N_USERS, D, generator=gen)
        z2 = z1 + 0.4 * torch.randn(N_USERS, D, generator=gen)
        a = z1[lo:hi].to(dev).requires_grad_(True)
        b = z2[lo:hi].to(dev).requires_grad_(True)
        share = gd.sharded_info_nce_loss(a, b, 0.2)
        share.backward()
        total = share.detach().clone()
        dist.all_reduce(total)
        res.update(nce=float(total), ga=a.grad.cpu().numpy(), gb=b.grad.cpu().numpy())
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend,overlap", [("gloo", True), ("gloo", False), ("nccl", True), ("nccl", False)])
def test_two_ranks_one_gpu_match_single_process(backend, overlap):
    """overlap=True is the two-stream, double-buffered schedule (distributed._propagate_two_streams), False the
    sequential one; backend "nccl" = one rank per GPU over RCCL, skipped where fewer than two GPUs are visible."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: fewer than 2 GPUs visible")
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out, backend, overlap)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        res = {r: out[r] for r in range(world)}
    pad = res[0]["pad"]
    u, i = O.synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=1)
    rowptr, col, val = O.norm_adj_csr(u, i, N_USERS, pad)
    x_all, w_all = _inputs(pad)
    ref, _ = O.lgcn_encoder_forward(rowptr, col, val, x_all, K, combine="mean")
    gacc = g = w_all.astype(np.float64)
    for _ in range(K):
        g = O.spmm_backward(rowptr, col, val, g, N_USERS + pad)
        gacc = gacc + g
    gref = gacc / (K + 1)
    tol = dict(rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    gt = dict(rtol=1e-5, atol=1e-5 * np.abs(gref).max())
    for r in range(world):
        lo, hi, ipr = res[r]["lo"], res[r]["hi"], res[r]["ipr"]
        np.testing.assert_allclose(res[r]["fu"], ref[lo:hi], **tol)
        np.testing.assert_allclose(res[r]["fi"], ref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **tol)
        np.testing.assert_allclose(res[r]["gu"], gref[lo:hi], **gt)
        np.testing.assert_allclose(res[r]["gi"], gref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **gt)
    # the edge-dropped view: assemble the global masked (non-symmetric) operator from every rank's
    # blocks and the oracle's Philox keep masks, then propagate / back-propagate in float64
    import scipy.sparse as sp
    n = N_USERS + pad
    rows, cols, vals = [], [], []
    for r in range(world):
        lo = res[r]["lo"]
        s_ui, s_iu = res[r]["seeds"]
        (rp, c, v), (rpt, ct, vt) = res[r]["ui"], res[r]["iu"]
        keep = O.edge_keep_mask(c.size, PE, seed=s_ui)
        ru = np.repeat(np.arange(rp.size - 1), np.diff(rp))
        rows.append(lo + ru[keep]); cols.append(N_USERS + c[keep]); vals.append(v[keep] / (1 - PE))
        keep_t = O.edge_keep_mask(ct.size, PE, seed=s_iu)
        ri = np.repeat(np.arange(rpt.size - 1), np.diff(rpt))
        rows.append(N_USERS + ri[keep_t]); cols.append(lo + ct[keep_t]); vals.append(vt[keep_t] / (1 - PE))
    a_m = sp.csr_matrix((np.concatenate(vals).astype(np.float64), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    assert abs(a_m.nnz / (2 * sum(res[r]["ui"][1].size for r in range(world))) - (1 - PE)) < 0.02
    z = acc = x_all.astype(np.float64)
    gz = gacc = w_all.astype(np.float64)
    for _ in range(K):
        z = a_m @ z
        acc = acc + z
        gz = a_m.T @ gz
        gacc = gacc + gz
    mref, mgref = acc / (K + 1), gacc / (K + 1)
    tol = dict(rtol=1e-5, atol=1e-5 * np.abs(mref).max())
    gt = dict(rtol=1e-5, atol=1e-5 * np.abs(mgref).max())
    for r in range(world):
        lo, hi, ipr = res[r]["lo"], res[r]["hi"], res[r]["ipr"]
        np.testing.assert_allclose(res[r]["mu"], mref[lo:hi], **tol)
        np.testing.assert_allclose(res[r]["mi"], mref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **tol)
        np.testing.assert_allclose(res[r]["mgu"], mgref[lo:hi], **gt)
        np.testing.assert_allclose(res[r]["mgi"], mgref[N_USERS + r * ipr: N_USERS + (r + 1) * ipr], **gt)
    gen = torch.Generator().manual_seed(5)
    z1 = torch.randn(N_USERS, D, generator=gen)
    z2 = z1 + 0.4 * torch.randn(N_USERS, D, generator=gen)
    nce = O.info_nce_loss(z1.numpy(), z2.numpy(), 0.2)
    w = np.full(N_USERS, 0.5 / N_USERS)
    g1, g2 = O.infonce_grads(z1.numpy(), z2.numpy(), np.arange(N_USERS), 5.0, True, w, w)
    per = N_USERS // world
    for r in range(world):
        assert res[r]["nce"] == pytest.approx(nce, rel=1e-5)
        np.testing.assert_allclose(res[r]["ga"], g1[r * per:(r + 1) * per], rtol=1e-4, atol=1e-5 * np.abs(g1).max())
        np.testing.assert_allclose(res[r]["gb"], g2[r * per:(r + 1) * per], rtol=1e-4, atol=1e-5 * np.abs(g2).max())


# --------------------------------------------------------------------------- BASELINE config 5 (MHCN channels)
def _mhcn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mhcn_sharded_common as C
        import recommendation_amd as ra
        from recommendation_amd.mhcn import HipOps
        dev = torch.device("cuda", 0)

        def make_graph(row, col, val, n_rows, n_cols):
            return ra.CsrGraph.from_coo(row, col, val, n_rows, n_cols, dev)

        out[rank] = C.run_rank(rank, world, dev, make_graph, HipOps)
    finally:
        dist.destroy_process_group()


def test_sharded_mhcn_two_ranks_one_gpu():
    """Config 5 with the real HIP kernels (dual-output SpMM on the row blocks of H_s / H_j / H_p, three streams,
    per-channel all-gather / reduce-scatter over gloo): values and gradients == the single-process float64
    restatement of univariate/mhcn.py:422-466."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mhcn_sharded_common as C
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_mhcn_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        res = {r: out[r] for r in range(world)}
    C.check(res, world, 2e-5)


# --------------------------------------------------------------------------- the real `nccl` (RCCL) backend, one rank
def _nccl_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import mhcn_sharded_common as C
        import recommendation_amd as ra
        from recommendation_amd import distributed as gd
        from recommendation_amd.mhcn import HipOps
        u, i = O.synthetic_interactions(N_USERS, N_ITEMS, N_EDGES, seed=1)
        deg_u, deg_i = np.bincount(u, minlength=N_USERS), np.bincount(i, minlength=N_ITEMS)
        g = gd.ShardedBipartiteGraph.from_local_interactions(u, i, N_USERS, N_ITEMS, deg_u, deg_i, 0, 1, dev)
        x_all, w_all = _inputs(g.items_padded)
        res = {}
        for force in (False, True):
            gd.FORCE_COLLECTIVES = force
            xu = torch.from_numpy(x_all[:N_USERS]).to(dev).requires_grad_(True)
            xi = torch.from_numpy(x_all[N_USERS:]).to(dev).requires_grad_(True)
            for overlap in (True, False):
                xu.grad = xi.grad = None
                view = gd.ShardedEdgeDrop(g, PE, seed=11, rescale=True)
                fu, fi = gd.sharded_lightgcn_propagate(g, xu, xi, K, combine="mean", overlap=overlap, view=view)
                nce = gd.sharded_info_nce_loss(fu, fu * 1.1 + 0.01, 0.2) + gd.sharded_info_nce_loss(fi, fi + 0.02, 0.2)
                items_full = gd.gather_items(fi)
                loss = (fu * torch.from_numpy(w_all[:N_USERS]).to(dev)).sum() + \
                    (items_full * torch.from_numpy(w_all[N_USERS:]).to(dev)).sum() + nce
                loss.backward()
                res[(force, overlap)] = [t.detach().cpu().numpy() for t in (fu, fi, xu.grad, xi.grad)] + [float(loss)]
            # config 5 through the same backend: per-channel all-gather / reduce-scatter / all-reduce calls
            res[("mhcn", force)] = C.run_rank(0, 1, dev, lambda r, c, v, nr, nc: ra.CsrGraph.from_coo(r, c, v, nr, nc, dev), HipOps)
        gd.FORCE_COLLECTIVES = False
        out["res"] = res
        out["backend"] = dist.get_backend()
    finally:
        dist.destroy_process_group()


def test_single_rank_rccl_runs_every_collective():
    """The one-GPU box cannot host two RCCL ranks, but one rank can run every collective call of distributed.py on the
    real `nccl` backend (FORCE_COLLECTIVES: all_gather_into_tensor, reduce_scatter_tensor, all_reduce, async handles,
    the per-channel streams) — shapes, dtypes, handle semantics and stream ordering as the 8-GPU run will see them.
    Results must equal the collective-free world-1 path."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mhcn_sharded_common as C
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        p = ctx.Process(target=_nccl_worker, args=(_free_port(), out))
        p.start()
        p.join(300)
        assert p.exitcode == 0
        res, backend = out["res"], out["backend"]
    assert backend == "nccl"
    base = res[(False, True)]
    for key in ((False, False), (True, True), (True, False)):
        for a, b in zip(res[key][:4], base[:4]):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6 * np.abs(b).max())
        assert res[key][4] == pytest.approx(base[4], rel=1e-6)
    C.check({0: res[("mhcn", True)]}, 1, 2e-5)
    C.check({0: res[("mhcn", False)]}, 1, 2e-5)
