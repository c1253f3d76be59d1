"""The host-side mirror of the reference interface (encoders, sampler, augmentation) driving the HIP
kernels end to end on the GPU, checked against the goldens / the oracle / a dense CPU PyTorch
restatement of one full NCL training step (ncl.py:310-329)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu


def _triples(golden):
    g = golden("graph_build.npz")
    return [[u, i, 1.0] for u, i in zip(g["train_user"].tolist(), g["train_item"].tolist())], g


def test_interaction_and_lgcn_encoder_match_reference(golden):
    from recommendation_amd.encoders import Interaction, LGCNEncoder
    train, g = _triples(golden)
    data = Interaction({}, train, [], device="cuda")
    assert [data.id2user[k] for k in range(data.user_num)] == g["sorted_user_ids"].tolist()
    assert [data.id2item[k] for k in range(data.item_num)] == g["sorted_item_ids"].tolist()
    p = golden("propagation.npz")
    for k in (1, 2, 3):
        enc = LGCNEncoder(data, 64, k)
        with torch.no_grad():
            enc.embedding_dict["user_emb"].copy_(torch.from_numpy(p["x0"][: data.user_num]))
            enc.embedding_dict["item_emb"].copy_(torch.from_numpy(p["x0"][data.user_num:]))
        ue, ie, all_emb = enc()
        final = torch.cat([ue, ie])
        ref = p[f"raw_mean_K{k}"]
        assert len(all_emb) == k + 1
        np.testing.assert_allclose(final.detach().cpu().numpy(), ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
        (final * torch.from_numpy(p["w"]).cuda()).sum().backward()
        gref = p[f"raw_grad_K{k}"]
        got = torch.cat([enc.embedding_dict["user_emb"].grad, enc.embedding_dict["item_emb"].grad]).cpu().numpy()
        np.testing.assert_allclose(got, gref, rtol=2e-5, atol=2e-5 * np.abs(gref).max())


def test_lightgcn_module_cfg1():
    """lightgcn.py:12-27 wiring at BASELINE cfg1 size (K = 2, d = 64) against the oracle restatement."""
    from recommendation_amd.encoders import LightGCN
    u, i = O.synthetic_interactions(943, 1682, 80000, seed=20250919)
    ei = torch.from_numpy(O.build_edge_index(u, i, 943))
    model = LightGCN(943, 1682, embedding_dim=64, num_layers=2).cuda()
    ei_dev = ei.cuda()
    ue, ie = model(ei_dev)
    ru, ri = O.lightgcn_forward(ei.numpy(), model.user_embedding.weight.detach().cpu().numpy(),
                                model.item_embedding.weight.detach().cpu().numpy(), 2)
    np.testing.assert_allclose(ue.detach().cpu().numpy(), ru, rtol=1e-5, atol=1e-5 * np.abs(ru).max())
    np.testing.assert_allclose(ie.detach().cpu().numpy(), ri, rtol=1e-5, atol=1e-5 * np.abs(ri).max())
    assert model.prepare(ei_dev) is model.prepare(ei_dev)              # operator cached per edge_index
    (ue.sum() + ie.sum()).backward()
    assert model.user_embedding.weight.grad is not None and torch.isfinite(model.user_embedding.weight.grad).all()


def test_next_batch_pairwise_contract(golden):
    from recommendation_amd.encoders import Interaction
    from recommendation_amd.sampler import next_batch_pairwise, randint_negatives
    train, _ = _triples(golden)
    data = Interaction({}, train, [], device="cuda")
    seen = []
    for u, i, j in next_batch_pairwise(data, 64, n_negs=2, seed=3, epoch=1):
        assert u.dtype == torch.int64 and j.numel() == 2 * u.numel()
        uu, jj = u.repeat_interleave(2).cpu().numpy(), j.cpu().numpy()
        for a, b in zip(uu, jj):
            assert data.id2item[int(b)] not in data.training_set_u[data.id2user[int(a)]]
        seen += list(zip(u.cpu().tolist(), i.cpu().tolist()))
    assert sorted(seen) == sorted(zip(data.uid.tolist(), data.iid.tolist()))      # one epoch = every pair once
    again = [b[2] for b in next_batch_pairwise(data, 64, n_negs=2, seed=3, epoch=1)]
    first = [b[2] for b in next_batch_pairwise(data, 64, n_negs=2, seed=3, epoch=1)]
    assert all(torch.equal(a, b) for a, b in zip(again, first))                   # reproducible
    r = randint_negatives(10000, 30, n_neg=3, seed=1, step=2)
    assert r.shape == (10000, 3) and int(r.min()) >= 0 and int(r.max()) < 30


def test_edge_removing_contract(golden):
    from recommendation_amd.sampler import EdgeRemoving
    g = golden("graph_build.npz")
    ei = torch.from_numpy(g["gcl_edge_index"]).cuda()
    big = ei.repeat(1, 200)
    aug = EdgeRemoving(pe=0.3, seed=4)
    v1, v2 = aug(big), aug(big)
    k1 = v1.materialize()
    assert abs(k1.shape[1] / big.shape[1] - 0.7) < 0.01 and k1.shape[0] == 2
    assert not torch.equal(v1.keep_bits, v2.keep_bits)                # two calls -> two views
    m = v1.keep_mask()
    assert torch.equal(k1, big[:, m])                                 # a column subset, order kept
    a = golden("augment.npz")
    assert abs(int(a["gcl_kept"]) / int(a["gcl_nnz"]) - 0.7) < 0.1     # the reference's own keep-rate


def test_full_ncl_training_step_against_dense_cpu(golden):
    """One NCL step (ncl.py:310-329 minus the faiss k-means, whose centroids are injected):
    encoder forward, BPR, l2 reg, structure contrast, prototype contrast, backward — HIP path vs a
    dense CPU PyTorch restatement on the same inputs."""
    from recommendation_amd import losses as Ls
    from recommendation_amd.encoders import Interaction, LGCNEncoder
    train, g = _triples(golden)
    c = golden("contrast.npz")
    data = Interaction({}, train, [], device="cuda")
    nu, ni = data.user_num, data.item_num
    n_layers, hyper_layers, tau, ssl_reg, alpha, proto_reg, reg, bsz = 3, 1, 0.1, 1e-3, 1.5, 1e-3, 1e-4, 32
    uidx, iidx = c["ncl_uidx"], c["ncl_iidx"]
    jidx = np.random.default_rng(0).integers(0, ni, bsz)
    x0 = c["ncl_x0"]
    enc = LGCNEncoder(data, 64, n_layers)
    with torch.no_grad():
        enc.embedding_dict["user_emb"].copy_(torch.from_numpy(x0[:nu]))
        enc.embedding_dict["item_emb"].copy_(torch.from_numpy(x0[nu:]))

    def step(user_w, item_w, spmm, loss_mod, dev):
        ue, ie, emb_list = spmm(user_w, item_w)
        ut, it, jt = (torch.as_tensor(v, device=dev) for v in (uidx, iidx, jidx))
        u_e, p_e, n_e = ue[ut], ie[it], ie[jt]
        rec = loss_mod["bpr"](u_e, p_e, n_e)
        l2 = loss_mod["l2"](reg, u_e, p_e, n_e) / bsz
        ssl = loss_mod["ssl"](emb_list[2 * hyper_layers], emb_list[0], ut, it)
        proto = loss_mod["proto"](emb_list[0], ut, it)
        return rec + l2 + ssl + proto, (rec, l2, ssl, proto)

    cents = {k: torch.from_numpy(c[k]) for k in ("ncl_ucent", "ncl_icent", "ncl_u2c", "ncl_i2c")}
    hip = {
        "bpr": Ls.bpr_loss, "l2": Ls.l2_reg_loss,
        "ssl": lambda ctx, ini, u, i: Ls.ssl_layer_loss(ctx, ini, u, i, nu, tau, ssl_reg, alpha),
        "proto": lambda ini, u, i: Ls.ProtoNCE_loss(ini, u, i, nu, cents["ncl_ucent"].cuda(), cents["ncl_u2c"].cuda(),
                                                   cents["ncl_icent"].cuda(), cents["ncl_i2c"].cuda(), tau, proto_reg, bsz),
    }
    total, parts = step(enc.embedding_dict["user_emb"], enc.embedding_dict["item_emb"], lambda a, b: enc(), hip, "cuda")
    total.backward()

    # dense CPU restatement (float64)
    rows, cols = torch.from_numpy(g["coo_row"]), torch.from_numpy(g["coo_col"])
    A = torch.zeros(nu + ni, nu + ni, dtype=torch.float64).index_put_((rows, cols), torch.ones(rows.numel(), dtype=torch.float64), accumulate=True)
    uw = torch.from_numpy(x0[:nu]).double().requires_grad_(True)
    iw = torch.from_numpy(x0[nu:]).double().requires_grad_(True)

    def dense_enc(a, b):
        emb = torch.cat([a, b])
        all_emb = [emb]
        for _ in range(n_layers):
            emb = A @ emb
            all_emb.append(emb)
        fin = torch.stack(all_emb).mean(0)
        return fin[:nu], fin[nu:], all_emb

    def d_infonce(v1, v2, t):
        v1, v2 = F.normalize(v1, dim=1), F.normalize(v2, dim=1)
        return -torch.diag(F.log_softmax(v1 @ v2.T / t, dim=1)).mean()

    def d_ssl(ctx, ini, u, i):
        out = 0
        for (cx, i0, idx, w) in ((ctx[:nu], ini[:nu], u, 1.0), (ctx[nu:], ini[nu:], i, alpha)):
            a, p_ = F.normalize(cx[idx]), F.normalize(i0[idx])
            out = out + w * -torch.log(torch.exp((a * p_).sum(1) / tau) / torch.exp(a @ F.normalize(i0).T / tau).sum(1)).sum()
        return ssl_reg * out

    dense = {
        "bpr": lambda u, p_, n_: torch.mean(-torch.log(10e-6 + torch.sigmoid((u * p_).sum(1) - (u * n_).sum(1)))),
        "l2": lambda r, *xs: r * sum(torch.norm(x, p=2) / x.shape[0] for x in xs),
        "ssl": d_ssl,
        "proto": lambda ini, u, i: proto_reg * bsz * (
            d_infonce(ini[:nu][u], cents["ncl_ucent"].double()[cents["ncl_u2c"][u]], tau)
            + d_infonce(ini[nu:][i], cents["ncl_icent"].double()[cents["ncl_i2c"][i]], tau)),
    }
    ref_total, ref_parts = step(uw, iw, dense_enc, dense, "cpu")
    ref_total.backward()
    for got, ref in zip(parts, ref_parts):
        assert float(got.detach()) == pytest.approx(float(ref.detach()), rel=1e-5)
    assert float(total.detach()) == pytest.approx(float(ref_total.detach()), rel=1e-5)
    for got, ref in ((enc.embedding_dict["user_emb"].grad, uw.grad), (enc.embedding_dict["item_emb"].grad, iw.grad)):
        ref = ref.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * np.abs(ref).max())


def test_first_seen_ids_and_load_data(golden, tmp_path):
    from recommendation_amd.encoders import Interaction, load_data
    train, g = _triples(golden)
    data = Interaction({}, train, [], device="cuda", normalised=True, id_order="first_seen")   # selfcf.py:258-306
    assert [data.id2user[k] for k in range(data.user_num)] == g["seen_user_ids"].tolist()
    assert [data.id2item[k] for k in range(data.item_num)] == g["seen_item_ids"].tolist()
    assert np.array_equal(data.norm_adj.rowptr.cpu().numpy(), g["norm_indptr"])
    np.testing.assert_allclose(data.norm_adj.val.cpu().numpy(), g["norm_data"], rtol=3e-7)
    tr, te = tmp_path / "train.txt", tmp_path / "test.txt"
    tr.write_text("".join(f"{a} {b} 1\n" for a, b in zip(g["gcl_user"], g["gcl_item"])))
    te.write_text("".join(f"{a} {b} 1\n" for a, b in zip(g["gcl_test_user"], g["gcl_test_item"])))
    ei, _, _, nu, ni = load_data(str(tr), str(te), device="cuda")
    assert (nu, ni) == (int(g["gcl_num_users"]), int(g["gcl_num_items"]))
    assert np.array_equal(ei.cpu().numpy(), g["gcl_edge_index"])          # gcl.load_data's own output


def test_multi_stream_spmm_mhcn_pattern():
    """config 5: three U x U channel operators + R^T / R launched on separate streams (mhcn.py:440-456)
    give the same results as serial launches, forward and backward."""
    import recommendation_amd as ra
    from recommendation_amd.encoders import multi_stream_spmm
    rng = np.random.default_rng(0)
    n_u, n_i, d = 3000, 800, 64
    graphs, xs, refs = [], [], []
    for k, (nr, nc, nnz) in enumerate([(n_u, n_u, 40000), (n_u, n_u, 25000), (n_u, n_u, 60000), (n_i, n_u, 30000), (n_u, n_i, 30000)]):
        row, col = rng.integers(0, nr, nnz), rng.integers(0, nc, nnz)
        val = rng.random(nnz).astype(np.float32)
        graphs.append(ra.CsrGraph.from_coo(row, col, val, nr, nc, "cuda"))
        x = rng.standard_normal((nc, d)).astype(np.float32)
        xs.append(torch.from_numpy(x).cuda().requires_grad_(True))
        rp, c, v, _ = O.coo_to_csr_stable(row, col, val, nr)
        refs.append(O.row_l2_normalize(O.spmm_csr(rp, c, v, x)))
    outs = multi_stream_spmm(graphs, xs, l2norm=True)
    for o, r in zip(outs, refs):
        np.testing.assert_allclose(o.detach().cpu().numpy(), r, rtol=1e-5, atol=1e-5)
    sum(o.sum() for o in outs).backward()
    serial = [ra.functional.spmm_l2norm(g, x.detach().clone().requires_grad_(True)) for g, x in zip(graphs, xs)]
    for o, s in zip(outs, serial):
        assert torch.equal(o, s)
    assert all(x.grad is not None and torch.isfinite(x.grad).all() for x in xs)


def test_ncl_model_trains_end_to_end():
    """NCLModel(conf, train, test).train() (ncl.py:282-394 protocol) on a block-structured toy set:
    every stage on the HIP path, metrics dict with the reference's keys, and it learns (Recall@20 far
    above the 20/120 chance level)."""
    from recommendation_amd.ncl import NCLModel
    rng = np.random.default_rng(0)
    n_u, n_i, groups = 300, 120, 6
    pairs = set()
    while len(pairs) < 7000:
        u = int(rng.integers(0, n_u))
        g = u % groups
        i = int(rng.integers(0, n_i // groups)) * groups + g if rng.random() < 0.9 else int(rng.integers(0, n_i))
        pairs.add((u, i))
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    train = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[:6000]]
    test = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[6000:]]
    conf = {"model": {"name": "NCL", "type": "graph"}, "embedding.size": 64, "batch.size": 512, "learning.rate": 0.005,
            "reg.lambda": 1e-4, "max.epoch": 8, "item.ranking.topN": [10, 20],
            "NCL": {"n_layers": 2, "tau": 0.1, "ssl_reg": 1e-4, "proto_reg": 1e-4, "alpha": 1.0, "num_clusters": 20,
                    "hyper_layers": 1}}
    model = NCLModel(conf, train, test, device="cuda", seed=1)
    metrics = model.train()
    assert set(metrics) == {"Hit Ratio", "Precision", "Recall", "NDCG"}
    # the dict keeps the last cut-off's values (Top 20), like the reference's comprehension; chance level is
    # 20 / 120 = 0.17.  Two layers at lr 0.005: Recall@20 0.72-0.84 over repeated runs (scripts/exp/ncl_stability.py).
    # With three layers of the raw (un-normalised, Q1) adjacency at lr 0.01 the trajectory is chaotic — the 1e-7
    # run-to-run noise of the float atomics grows into anything between 0.19 and 0.82 — so that is not a test.
    print("NCL end-to-end metrics:", metrics)
    assert metrics["Recall"] > 0.5, metrics
    assert model.user_2cluster.shape == (model.data.user_num,) and int(model.user_2cluster.max()) < 20
    scores = model.predict("u0")
    assert scores.shape == (model.data.item_num,)


def test_propagation_is_hipgraph_capturable():
    """INTEGRATION.md's contract — nothing is allocated or synchronised inside a libgcr call — means a launch-bound
    inner loop can be captured in a hipGraph: K-layer propagation at MovieLens-100K size (cfg1: three ~10 us launches)
    captured once and replayed on new inputs equals the eager result bit for bit."""
    import recommendation_amd as ra
    from recommendation_amd import functional as Fn
    n_u, n_i = 943, 1682
    u, i = O.synthetic_interactions(n_u, n_i, 80000, seed=3)
    g = ra.CsrGraph.bipartite_sym_norm(u, i, n_u, n_i, "cuda")
    x = torch.randn(n_u + n_i, 64, device="cuda")
    with torch.no_grad():
        Fn.lightgcn_propagate(g, x, 2, "sum")                       # warm-up: split-row workspace, allocator pools
    torch.cuda.synchronize()
    static_x = x.clone()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        Fn.lightgcn_propagate(g, static_x, 2, "sum")                # the capture stream's own workspace
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=s):
            static_out = Fn.lightgcn_propagate(g, static_x, 2, "sum")
    torch.cuda.current_stream().wait_stream(s)
    for seed in (1, 2):
        xn = torch.randn(n_u + n_i, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
        static_x.copy_(xn)
        graph.replay()
        torch.cuda.synchronize()
        with torch.no_grad():
            ref = Fn.lightgcn_propagate(g, xn, 2, "sum")
        assert torch.equal(static_out, ref)
