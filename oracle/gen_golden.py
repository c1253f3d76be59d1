#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — generate tests/golden/*.npz by running the reference itself.

Runs ONLY in the build container (it needs /root/reference, which never travels to the GPU
box); the fixtures it writes are plain data: seeded inputs + the outputs the reference's own
functions produced on them with CPU PyTorch fp32.

How each reference function is reached:
  * modules that import as-is here: gcl.py, directau.py, selfcf.py, univariate/sept.py,
    univariate/buir.py (directau.Interaction / LGCNEncoder are textually ncl.py's);
  * ncl.py and ssl4rec.py stop at `from numba import jit` / `import faiss` (ordinary
    ModuleNotFoundError; both packages stay absent).  Their loss functions do not use either
    package, so the named FunctionDef nodes are taken out of the reference file with `ast` at
    run time and executed unchanged against torch — no stand-in libraries, no source copied;
  * lightgcn.py needs torch_geometric (absent) and its loss block lives in the body of the epoch loop
    (lightgcn.py:95-118), like gcl.py's (gcl.py:216-223): `load_stmts` takes that range of statements out of
    the enclosing function's AST and executes them unchanged in a namespace holding the fixture's tensors
    (nothing re-typed); LGConv itself stays "parity unpinned".

Usage:  python oracle/gen_golden.py   (writes tests/golden/)
"""
import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "univariate"))


def load_defs(path, names, methods=()):
    """exec the named top-level functions (and `Class.method`s, as plain functions) of a reference file."""
    tree = ast.parse(open(path).read())
    wanted = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            wanted.append(node)
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and f"{node.name}.{sub.name}" in methods:
                    wanted.append(sub)
    ns = {"torch": torch, "F": F, "nn": torch.nn, "np": np, "device": torch.device("cpu")}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), path, "exec"), ns)
    return ns


def load_stmts(path, func, first, last):
    """Code object of the statements of top-level function (or `Class.method`) `func` in a reference file that lie in
    the line range [first, last] — a slice of ONE statement list, however deeply nested in loops / try / with blocks
    (the loss blocks of lightgcn.py / gcl.py are bodies of training loops).  Executed unchanged with `exec(code, ns)`."""
    tree = ast.parse(open(path).read())
    target = None
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == func:
            target = node
        if isinstance(node, ast.ClassDef):
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and f"{node.name}.{sub.name}" == func:
                    target = sub
    if target is None:
        raise KeyError(f"{func} not found in {path}")

    def find(body):
        inside = [st for st in body if st.lineno >= first and st.end_lineno <= last]
        if inside and inside[0].lineno == first and inside[-1].end_lineno == last:
            return inside
        for st in body:
            if st.lineno <= first and st.end_lineno >= last:
                for field in ("body", "orelse", "finalbody", "handlers"):
                    sub = getattr(st, field, None)
                    if sub:
                        got = find([h for h in sub if isinstance(h, ast.stmt)] +
                                   [x for h in sub if isinstance(h, ast.ExceptHandler) for x in h.body])
                        if got:
                            return got
        return None

    stmts = find(target.body)
    if not stmts:
        raise KeyError(f"{path}:{first}-{last} is not a run of whole statements of one block of {func}")
    return compile(ast.Module(body=stmts, type_ignores=[]), f"{path}:{first}-{last}", "exec")


def seeded_triples(rng, n_users, n_items, n_inter, n_dup):
    raw_u = rng.choice(np.arange(1000, 1000 + 3 * n_users), n_users, replace=False)
    raw_i = rng.choice(np.arange(5000, 5000 + 3 * n_items), n_items, replace=False)
    u = raw_u[rng.integers(0, n_users, n_inter)]
    i = raw_i[rng.integers(0, n_items, n_inter)]
    # every user and every item at least once, plus explicit duplicate pairs
    u = np.concatenate([raw_u, raw_u[rng.integers(0, n_users, n_items)], u, u[:n_dup]])
    i = np.concatenate([raw_i[rng.integers(0, n_items, n_users)], raw_i, i, i[:n_dup]])
    p = rng.permutation(u.size)
    return [[str(a), str(b), 1.0] for a, b in zip(u[p], i[p])]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    rng = np.random.default_rng(20250919)
    import directau, selfcf, sept, gcl  # noqa: E401  (reference modules, imported as-is)

    # ---------------------------------------------------------------- 1. graph build
    train = seeded_triples(rng, 50, 30, 400, 12)
    test = [[train[k][0], train[(k * 7) % len(train)][1], 1.0] for k in range(0, 60, 3)] + [["999999", "888888", 1.0]]
    tr_u = np.array([t[0] for t in train])
    tr_i = np.array([t[1] for t in train])

    d_int = directau.Interaction({}, train, test)          # == ncl.py:46-88
    coo = d_int.norm_adj
    s_int = selfcf.Interaction({}, train, test)
    s_adj = s_int.norm_adj.tocsr()
    s_adj.sort_indices()

    # gcl.load_data wants integer ids in space-separated files
    int_u = rng.integers(0, 40, 300)
    int_i = rng.integers(0, 25, 300)
    tmp = tempfile.mkdtemp()
    with open(os.path.join(tmp, "train.txt"), "w") as f:
        f.writelines(f"{a} {b} 1\n" for a, b in zip(int_u, int_i))
    with open(os.path.join(tmp, "test.txt"), "w") as f:
        f.writelines(f"{a} {b} 1\n" for a, b in zip(int_u[:20] + 1, int_i[:20] + 2))
    edge_index, _, _, g_nu, g_ni = gcl.load_data(os.path.join(tmp, "train.txt"), os.path.join(tmp, "test.txt"))

    np.savez_compressed(
        os.path.join(OUT, "graph_build.npz"),
        train_user=tr_u, train_item=tr_i,
        sorted_user_ids=np.array([d_int.id2user[k] for k in range(d_int.user_num)]),
        sorted_item_ids=np.array([d_int.id2item[k] for k in range(d_int.item_num)]),
        coo_row=coo.row.astype(np.int64), coo_col=coo.col.astype(np.int64), coo_data=coo.data.astype(np.float32),
        seen_user_ids=np.array([s_int.id2user[k] for k in range(s_int.user_num)]),
        seen_item_ids=np.array([s_int.id2item[k] for k in range(s_int.item_num)]),
        norm_indptr=s_adj.indptr.astype(np.int64), norm_indices=s_adj.indices.astype(np.int64),
        norm_data=s_adj.data.astype(np.float32),
        gcl_user=int_u, gcl_item=int_i, gcl_test_user=int_u[:20] + 1, gcl_test_item=int_i[:20] + 2,
        gcl_edge_index=edge_index.numpy(), gcl_num_users=g_nu, gcl_num_items=g_ni,
    )

    # ---------------------------------------------------------------- 2. propagation
    n = d_int.user_num + d_int.item_num
    d = 64
    x0 = torch.empty(n, d)
    torch.nn.init.xavier_uniform_(x0, generator=torch.Generator().manual_seed(0))
    wgt = torch.randn(n, d, generator=torch.Generator().manual_seed(1))
    out = {"x0": x0.numpy(), "w": wgt.numpy()}
    for k_layers in (1, 2, 3):
        enc = directau.LGCNEncoder(d_int, d, k_layers)
        with torch.no_grad():
            enc.embedding_dict["user_emb"].copy_(x0[: d_int.user_num])
            enc.embedding_dict["item_emb"].copy_(x0[d_int.user_num:])
        ue, ie, all_emb = enc()
        final = torch.cat([ue, ie])
        (final * wgt).sum().backward()
        out[f"raw_mean_K{k_layers}"] = final.detach().numpy()
        out[f"raw_last_K{k_layers}"] = all_emb[-1].detach().numpy()
        out[f"raw_grad_K{k_layers}"] = torch.cat([enc.embedding_dict["user_emb"].grad, enc.embedding_dict["item_emb"].grad]).numpy()

    ns = s_int.user_num + s_int.item_num
    xs = torch.empty(ns, d)
    torch.nn.init.xavier_uniform_(xs, generator=torch.Generator().manual_seed(2))
    out["xs"] = xs.numpy()
    for k_layers in (2, 3):
        enc = selfcf.LGCN_Encoder(s_int, d, k_layers)
        with torch.no_grad():
            enc.embedding_dict["user_emb"].copy_(xs[: s_int.user_num])
            enc.embedding_dict["item_emb"].copy_(xs[s_int.user_num:])
        ue, ie = enc()
        final = torch.cat([ue, ie])
        (final * wgt[:ns]).sum().backward()
        out[f"norm_mean_K{k_layers}"] = final.detach().numpy()
        out[f"norm_grad_K{k_layers}"] = torch.cat([enc.embedding_dict["user_emb"].grad, enc.embedding_dict["item_emb"].grad]).numpy()

    # sept.SEPT.encoder (sept.py:220-226) on the coalesced raw adjacency it builds (sept.py:42-50)
    sp_int = sept.Interaction({}, train, test)
    adj_t = sept.TFGraphInterface.convert_sparse_mat_to_tensor_inputs(sp_int.norm_adj)
    fake = types.SimpleNamespace(n_layers=2)
    xr = x0.clone().requires_grad_(True)
    fin = sept.SEPT.encoder(fake, xr, adj_t)
    (fin * wgt).sum().backward()
    out["sept_mean_K2"] = fin.detach().numpy()
    out["sept_grad_K2"] = xr.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "propagation.npz"), **out)

    # ---------------------------------------------------------------- 3. contrast
    ncl = load_defs(os.path.join(REF, "ncl.py"), {"InfoNCE", "bpr_loss", "l2_reg_loss"},
                    {"NCLModel.ssl_layer_loss", "NCLModel.ProtoNCE_loss"})
    s4r = load_defs(os.path.join(REF, "ssl4rec.py"), {"batch_softmax_loss", "InfoNCE", "l2_reg_loss"})
    out = {}
    g = torch.Generator().manual_seed(3)
    for m in (1, 7, 257, 1000):
        z1 = torch.randn(m, d, generator=g)
        z2 = (z1 + 0.5 * torch.randn(m, d, generator=g))
        out[f"z1_{m}"], out[f"z2_{m}"] = z1.numpy(), z2.numpy()
        for temp in (0.1, 0.2, 0.5):
            a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
            loss = gcl.info_nce_loss(a, b, temp)
            loss.backward()
            out[f"gcl_loss_{m}_{temp}"] = loss.item()
            if m in (7, 257) and temp == 0.2:
                out[f"gcl_g1_{m}"], out[f"gcl_g2_{m}"] = a.grad.numpy(), b.grad.numpy()
        for b_cos in (True, False):
            a, b = (0.3 * z1).clone().requires_grad_(True), (0.3 * z2).clone().requires_grad_(True)
            loss = ncl["InfoNCE"](a, b, 0.2, b_cos)
            loss.backward()
            out[f"ncl_infonce_{m}_{int(b_cos)}"] = loss.item()
            if m == 257:
                out[f"ncl_infonce_g1_{m}_{int(b_cos)}"] = a.grad.numpy()
                out[f"ncl_infonce_g2_{m}_{int(b_cos)}"] = b.grad.numpy()
            out[f"s4r_infonce_{m}_{int(b_cos)}"] = s4r["InfoNCE"](0.3 * z1, 0.3 * z2, 0.2, b_cos).item()
        out[f"s4r_bsl_{m}"] = s4r["batch_softmax_loss"](z1, z2, 0.2).item()

    # NCL structure / prototype losses on the toy graph (centroids + assignments injected)
    nu, ni = d_int.user_num, d_int.item_num
    ctx = torch.randn(n, d, generator=g) * 0.1 + x0
    bsz = 32
    uidx = torch.randint(0, nu, (bsz,), generator=g).tolist()
    iidx = torch.randint(0, ni, (bsz,), generator=g).tolist()
    kc = 5
    self_ = types.SimpleNamespace(
        data=types.SimpleNamespace(user_num=nu, item_num=ni), ssl_temp=0.1, ssl_reg=1e-6, alpha=1.5,
        proto_reg=8e-8, batch_size=bsz,
        user_centroids=torch.randn(kc, d, generator=g), item_centroids=torch.randn(kc, d, generator=g),
        user_2cluster=torch.randint(0, kc, (nu,), generator=g), item_2cluster=torch.randint(0, kc, (ni,), generator=g))
    ctx_r, x0_r = ctx.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    ssl = ncl["ssl_layer_loss"](self_, ctx_r, x0_r, uidx, iidx)
    ssl.backward()
    x0_p = x0.clone().requires_grad_(True)
    # ProtoNCE_loss calls the module-level InfoNCE: it is in the same exec namespace
    proto = ncl["ProtoNCE_loss"](self_, x0_p, uidx, iidx)
    proto.backward()
    out.update(ncl_ctx=ctx.numpy(), ncl_x0=x0.numpy(), ncl_uidx=np.array(uidx), ncl_iidx=np.array(iidx),
               ncl_num_users=nu, ncl_ssl_temp=0.1, ncl_ssl_reg=1e-6, ncl_alpha=1.5, ncl_proto_reg=8e-8, ncl_bsz=bsz,
               ncl_ucent=self_.user_centroids.numpy(), ncl_icent=self_.item_centroids.numpy(),
               ncl_u2c=self_.user_2cluster.numpy(), ncl_i2c=self_.item_2cluster.numpy(),
               ncl_ssl=ssl.item(), ncl_ssl_gctx=ctx_r.grad.numpy(), ncl_ssl_gx0=x0_r.grad.numpy(),
               ncl_proto=proto.item(), ncl_proto_gx0=x0_p.grad.numpy())
    np.savez_compressed(os.path.join(OUT, "contrast.npz"), **out)

    # ---------------------------------------------------------------- 4. BPR / regularisers
    out = {}
    ut = torch.randn(40, d, generator=g) * 0.3
    it = torch.randn(25, d, generator=g) * 0.3
    bq = 300
    ui = torch.randint(0, 40, (bq,), generator=g)
    pi = torch.randint(0, 25, (bq,), generator=g)
    ni1 = torch.randint(0, 25, (bq,), generator=g)
    ni3 = torch.randint(0, 25, (bq, 3), generator=g)
    out.update(user_tab=ut.numpy(), item_tab=it.numpy(), u_idx=ui.numpy(), i_idx=pi.numpy(), j_idx=ni1.numpy(), j_idx3=ni3.numpy())

    def run(fn_loss, name, neg_idx):
        a, b = ut.clone().requires_grad_(True), it.clone().requires_grad_(True)
        loss = fn_loss(a, b, neg_idx)
        loss.backward()
        out[f"{name}_loss"], out[f"{name}_gu"], out[f"{name}_gi"] = loss.item(), a.grad.numpy(), b.grad.numpy()

    run(lambda a, b, nj: ncl["bpr_loss"](a[ui], b[pi], b[nj]), "ncl_bpr", ni1)          # ncl.py:116-120
    run(lambda a, b, nj: sept.bpr_loss(a[ui], b[pi], b[nj]), "sept_bpr", ni1)             # sept.py:34-38
    run(lambda a, b, nj: ncl["l2_reg_loss"](1e-4, a[ui], b[pi], b[nj]), "ncl_l2reg", ni1)  # ncl.py:122-123
    out["directau_l2reg_loss"] = directau.l2_reg_loss(1e-4, ut[ui], it[pi], it[ni1]).item()

    # lightgcn.py:95-118 (gathers, BPR / BCE branch, regulariser): the statements of train_model's epoch loop, lifted
    lgcn_code = load_stmts(os.path.join(REF, "lightgcn.py"), "train_model", 95, 118)

    def lightgcn_block(loss_type):
        def fn(a, b, nj, reg_weight=1e-4):
            ns_ = {"torch": torch, "F": F, "user_emb": a, "item_emb": b, "pos_u": ui, "pos_i": pi, "neg_i": nj,
                   "config": {"n_neg": 1 if nj.dim() == 1 else nj.shape[1], "loss_type": loss_type, "reg_weight": reg_weight}}
            exec(lgcn_code, ns_)
            return ns_["loss"]
        return fn

    run(lightgcn_block("bpr"), "lgcn_block_n1", ni1)
    run(lightgcn_block("bpr"), "lgcn_block_n3", ni3)
    # loss_type == "bce" (lightgcn.py:109-113): all-pairs scores [B, I], one-hot labels, BCE-with-logits + the regulariser
    run(lightgcn_block("bce"), "lgcn_bce", ni1)
    # ... the same block at the magnitudes of TRAINED embeddings (scores of several units, both signs) and with
    # repeated (user, item) pairs in the batch
    ut_w, it_w = ut.clone(), it.clone()
    ut, it = ut_w * 4.0, it_w * 3.0
    out.update(user_tab_big=ut.numpy(), item_tab_big=it.numpy())
    run(lightgcn_block("bce"), "lgcn_bce_big", ni1)
    ut, it = ut_w, it_w

    # gcl.py:216-223 (gathers, BPR, regulariser, total) lifted from the batch loop of GCLTuner.run; the contrast term
    # of :223 enters as a zero (it has its own fixtures)
    gcl_code = load_stmts(os.path.join(REF, "gcl.py"), "GCLTuner.run", 216, 223)

    def gcl_block(a, b, nj, reg_weight=1e-4):
        ns_ = {"torch": torch, "F": F, "device": torch.device("cpu"), "user_z1": a, "item_z1": b, "users": ui,
               "pos_items": pi, "neg_items": nj, "ssl_loss": torch.zeros(()), "config": {"reg_weight": reg_weight}}
        exec(gcl_code, ns_)
        return ns_["total_loss"]

    run(gcl_block, "gcl_block", ni1)
    np.savez_compressed(os.path.join(OUT, "bpr.npz"), **out)

    # ---------------------------------------------------------------- 5. augmentation properties
    torch.manual_seed(5)
    np.random.seed(5)
    ei = edge_index
    kept = gcl.EdgeRemoving(0.3)(ei)
    dropped = sept.GraphAugmentor.edge_dropout(sp_int.norm_adj.tocsr(), 0.25)
    nnz_unique = len(sp_int.norm_adj.tocsr().nonzero()[0])
    np.savez_compressed(os.path.join(OUT, "augment.npz"),
                        gcl_pe=0.3, gcl_nnz=ei.shape[1], gcl_kept=kept.shape[1],
                        sept_rate=0.25, sept_nnz=nnz_unique, sept_kept=dropped.nnz,
                        sept_vals_all_one=bool((dropped.data == 1.0).all()))
    print("golden fixtures written to", os.path.abspath(OUT))
    for fn in sorted(os.listdir(OUT)):
        print("  ", fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


def gen_eval():
    """tests/golden/eval.json: ncl.py's own ranking_evaluation / Metric (ncl.py:133-177) on a seeded
    (test_set, rec_list) pair — the strings the reference prints."""
    import json
    import math
    tree = ast.parse(open(os.path.join(REF, "ncl.py")).read())
    wanted = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "Metric")
              or (isinstance(n, ast.FunctionDef) and n.name == "ranking_evaluation")]
    ns = {"np": np, "math": math}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), "ncl.py", "exec"), ns)
    rng = np.random.default_rng(11)
    users = [f"u{k}" for k in range(40)]
    origin = {u: {f"i{int(j)}": 1 for j in rng.choice(200, int(rng.integers(1, 9)), replace=False)} for u in users}
    res = {u: [(f"i{int(j)}", float(s)) for j, s in zip(rng.choice(200, 50, replace=False), -np.sort(-rng.random(50)))]
           for u in users}
    for u in users[:25]:        # make sure there are hits at different ranks
        k = list(origin[u])[0]
        pos = int(rng.integers(0, 50))
        res[u][pos] = (k, res[u][pos][1])
    out = ns["ranking_evaluation"](origin, res, [10, 20, 30, 50])
    with open(os.path.join(OUT, "eval.json"), "w") as f:
        json.dump({"origin": origin, "res": res, "N": [10, 20, 30, 50], "lines": out}, f)
    print("wrote eval.json:", "".join(out[:5]).replace("\n", " | "))


def gen_rownorm():
    """tests/golden/rownorm.npz: selfcf.Graph.normalize_graph_mat on a rectangular matrix
    (selfcf.py:250-254; the same method body as ncl.py:37-41 and the one MHCN uses for R / H)."""
    import scipy.sparse as sp
    import selfcf
    rng = np.random.default_rng(5)
    n_rows, n_cols, nnz = 60, 35, 500
    row, col = rng.integers(0, n_rows, nnz), rng.integers(0, n_cols, nnz)
    row[row == 7] = 8                                   # an empty row -> 1/0 -> 0
    val = rng.integers(1, 4, nnz).astype(np.float32)
    a = sp.csr_matrix((val, (row, col)), shape=(n_rows, n_cols), dtype=np.float32)
    out = selfcf.Graph.normalize_graph_mat(a).tocsr()
    out.sort_indices()
    np.savez_compressed(os.path.join(OUT, "rownorm.npz"), row=row, col=col, val=val, n_rows=n_rows, n_cols=n_cols,
                        indptr=out.indptr.astype(np.int64), indices=out.indices.astype(np.int64),
                        data=out.data.astype(np.float32))
    print("wrote rownorm.npz", out.nnz)


def gen_grace():
    """tests/golden/grace.npz: univariate/grace.py's DualBranchContrast(InfoNCE(tau), 'L2L', intraview_negs)
    (:196-224, :380-419, :448-502) run as-is on seeded views.  grace.py imports torch_geometric at module
    level (not installed), so the pure-torch classes are lifted out of its AST and executed unchanged.
    Two variants per case: extra_neg_mask=None — the call the model makes; add_extra_mask then REPLACES the
    sampler's negative mask by 1 - pos_mask, so the anchor's own row counts as a negative — and
    extra_neg_mask=ones, which keeps the sampler's mask (diagonal of the intra-view block excluded)."""
    import torch
    import torch.nn.functional as F
    from abc import ABC, abstractmethod
    tree = ast.parse(open(os.path.join(REF, "univariate", "grace.py")).read())
    names = {"_similarity", "Loss", "InfoNCE", "Sampler", "SameScaleSampler", "CrossScaleSampler", "get_sampler",
             "add_extra_mask", "DualBranchContrast"}
    wanted = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in wanted} == names
    ns = {"torch": torch, "F": F, "ABC": ABC, "abstractmethod": abstractmethod}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), "grace.py", "exec"), ns)
    rng = np.random.default_rng(21)
    out = {}
    for m in (7, 257):
        h1 = rng.standard_normal((m, 64)).astype(np.float32)
        h2 = (h1 + 0.5 * rng.standard_normal((m, 64))).astype(np.float32)
        out[f"h1_{m}"], out[f"h2_{m}"] = h1, h2
        for tau in (0.2, 0.5):
            for intra in (0, 1):
                for keep_sampler_mask in (0, 1):
                    if keep_sampler_mask and not intra:
                        continue
                    t1, t2 = torch.from_numpy(h1).requires_grad_(True), torch.from_numpy(h2).requires_grad_(True)
                    model = ns["DualBranchContrast"](loss=ns["InfoNCE"](tau=tau), mode="L2L", intraview_negs=bool(intra))
                    extra = torch.ones(m, 2 * m) if keep_sampler_mask else None
                    loss = model(h1=t1, h2=t2, extra_neg_mask=extra)
                    loss.backward()
                    key = f"{m}_{tau}_{intra}_{keep_sampler_mask}"
                    out[f"loss_{key}"] = np.float32(loss.item())
                    out[f"g1_{key}"], out[f"g2_{key}"] = t1.grad.numpy(), t2.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "grace.npz"), **out)
    print("wrote grace.npz", {k: float(v) for k, v in out.items() if k.startswith("loss_257_0.2")})


def _lift_class_methods(path, class_name, method_names, extra_ns=None, top_level=()):
    """Methods of one reference class (and optionally top-level defs) as a throw-away class whose
    bodies are the reference's own AST nodes, executed unchanged (the module itself stops at an import of
    an absent package: tensorflow / torch_geometric)."""
    tree = ast.parse(open(path).read())
    methods, tops = [], []
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == class_name:
            methods = [sub for sub in node.body if isinstance(sub, ast.FunctionDef) and sub.name in method_names]
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in top_level:
            tops.append(node)
    assert {m.name for m in methods} == set(method_names), (class_name, [m.name for m in methods])
    import scipy.sparse as sp
    ns = {"torch": torch, "F": F, "nn": torch.nn, "np": np, "sp": sp, "device": torch.device("cpu")}
    ns.update(extra_ns or {})
    exec(compile(ast.Module(body=tops, type_ignores=[]), path, "exec"), ns)
    fn_ns = dict(ns)
    exec(compile(ast.Module(body=methods, type_ignores=[]), path, "exec"), fn_ns)
    cls = type("Lifted" + class_name, (), {m.name: fn_ns[m.name] for m in methods})
    return cls, ns


def _csr_dict(prefix, m):
    m = m.tocsr().astype(np.float32)
    m.sum_duplicates()
    m.sort_indices()
    return {f"{prefix}_indptr": m.indptr.astype(np.int64), f"{prefix}_indices": m.indices.astype(np.int64),
            f"{prefix}_data": m.data.astype(np.float32), f"{prefix}_shape": np.array(m.shape, dtype=np.int64)}


def gen_mhcn():
    """tests/golden/mhcn.npz: univariate/mhcn.py's own MHCN.build_hyper_adj_mats (:340-368), the channel layer
    loop of MHCN.forward (:422-478: raw product fed forward, normalised copy summed) with self_gating /
    channel_attention (:404-420) and hierarchical_self_supervision (:480-506), lifted from the AST (the module
    imports tensorflow, which is absent) and run unchanged on a seeded social + interaction graph.  The
    torch.randperm draws of the self-supervision are recorded so that the test can replay them."""
    import scipy.sparse as sp
    path = os.path.join(REF, "univariate", "mhcn.py")
    names = {"build_hyper_adj_mats", "self_gating", "self_supervised_gating", "channel_attention", "forward",
             "hierarchical_self_supervision", "sparse_mx_to_torch_sparse_tensor"}
    cls, ns = _lift_class_methods(path, "MHCN", names, top_level=("TFGraphInterface", "Graph"))
    rng = np.random.default_rng(77)
    n_u, n_i, d, n_layers = 60, 45, 64, 2
    # directed social relations with a good share of reciprocal pairs, so that every motif has members
    s_r, s_c = rng.integers(0, n_u, 420), rng.integers(0, n_u, 420)
    keep = s_r != s_c
    s_r, s_c = s_r[keep], s_c[keep]
    s_r, s_c = np.concatenate([s_r, s_c[:150]]), np.concatenate([s_c, s_r[:150]])
    pairs = np.unique(np.stack([s_r, s_c], 1), axis=0)
    s_r, s_c = pairs[:, 0], pairs[:, 1]
    S = sp.csr_matrix((np.ones(len(s_r), dtype=np.float32), (s_r, s_c)), shape=(n_u, n_u), dtype=np.float32)
    y_r = np.concatenate([np.arange(n_u), rng.integers(0, n_u, 500)])
    y_c = np.concatenate([rng.integers(0, n_i, n_u), rng.integers(0, n_i, 500)])
    ypairs = np.unique(np.stack([y_r, y_c], 1), axis=0)
    y_r, y_c = ypairs[:, 0], ypairs[:, 1]
    Y = sp.csr_matrix((np.ones(len(y_r), dtype=np.float32), (y_r, y_c)), shape=(n_u, n_i), dtype=np.float32)

    m = cls()
    m.social_data = types.SimpleNamespace(get_social_mat=lambda: S)
    m.data = types.SimpleNamespace(interaction_mat=Y, user_num=n_u, item_num=n_i)
    H = m.build_hyper_adj_mats()
    H = [sp.csr_matrix(h) for h in H]
    R = ns["Graph"].normalize_graph_mat(Y)
    g = torch.Generator().manual_seed(7)

    def xav(*shape):
        t = torch.empty(*shape)
        torch.nn.init.xavier_uniform_(t, generator=g)
        return t

    m.n_layers, m.n_channel, m.emb_size, m.ss_rate = n_layers, 4, d, 0.01
    m.user_embeddings = xav(n_u, d).requires_grad_(True)
    m.item_embeddings = xav(n_i, d).requires_grad_(True)
    m.gating_weights = {str(c + 1): xav(d, d) for c in range(4)}
    m.gating_bias = {str(c + 1): 0.05 * torch.randn(1, d, generator=g) for c in range(4)}
    m.sgating_weights = {str(c + 1): xav(d, d) for c in range(4)}
    m.sgating_bias = {str(c + 1): 0.05 * torch.randn(1, d, generator=g) for c in range(4)}
    m.attention = xav(1, d)
    m.attention_mat = xav(d, d)
    m.H_s, m.H_j, m.H_p = (m.sparse_mx_to_torch_sparse_tensor(h) for h in H)
    m.R = m.sparse_mx_to_torch_sparse_tensor(R)

    # record the permutations the self-supervision draws (torch.randperm(n, device=cpu))
    perms = []
    real_randperm = torch.randperm

    def recording_randperm(n, **kw):
        p = real_randperm(n, generator=g)
        perms.append(p.numpy().copy())
        return p

    u_idx = torch.from_numpy(rng.integers(0, n_u, 32))
    v_idx = torch.from_numpy(rng.integers(0, n_i, 32))
    j_idx = torch.from_numpy(rng.integers(0, n_i, 32))
    torch.randperm = recording_randperm
    try:
        bu, bp, bn, ss_loss, fu, fi = m.forward(u_idx, v_idx, j_idx)
    finally:
        torch.randperm = real_randperm
    wu = torch.randn(n_u, d, generator=g)
    wi = torch.randn(n_i, d, generator=g)
    ((fu * wu).sum() + (fi * wi).sum() + ss_loss).backward()
    out = {"S_row": s_r.astype(np.int64), "S_col": s_c.astype(np.int64), "Y_row": y_r.astype(np.int64),
           "Y_col": y_c.astype(np.int64), "n_users": n_u, "n_items": n_i, "n_layers": n_layers, "ss_rate": 0.01,
           "user_emb": m.user_embeddings.detach().numpy(), "item_emb": m.item_embeddings.detach().numpy(),
           "attention": m.attention.numpy(), "attention_mat": m.attention_mat.numpy(),
           "u_idx": u_idx.numpy(), "v_idx": v_idx.numpy(), "j_idx": j_idx.numpy(),
           "perms": np.stack(perms), "wu": wu.numpy(), "wi": wi.numpy(),
           "final_user": fu.detach().numpy(), "final_item": fi.detach().numpy(), "ss_loss": np.float32(ss_loss.item()),
           "batch_user": bu.detach().numpy(), "batch_pos": bp.detach().numpy(), "batch_neg": bn.detach().numpy(),
           "grad_user": m.user_embeddings.grad.numpy(), "grad_item": m.item_embeddings.grad.numpy()}
    for c in range(4):
        out[f"gw{c + 1}"], out[f"gb{c + 1}"] = m.gating_weights[str(c + 1)].numpy(), m.gating_bias[str(c + 1)].numpy()
        out[f"sgw{c + 1}"], out[f"sgb{c + 1}"] = m.sgating_weights[str(c + 1)].numpy(), m.sgating_bias[str(c + 1)].numpy()
    for name, h in zip(("H_s", "H_j", "H_p"), H):
        out.update(_csr_dict(name, h))
    out.update(_csr_dict("R", sp.csr_matrix(R)))
    np.savez_compressed(os.path.join(OUT, "mhcn.npz"), **out)
    print("wrote mhcn.npz: nnz", [h.nnz for h in H], "ss_loss", float(ss_loss), "perms", len(perms))


def gen_sept_social():
    """tests/golden/sept_social.npz: univariate/sept_social.py's SEPT.encoder / social_encoder (:370-385: SUM of the
    row-normalised layers), get_social_related_views (:361-368), label_prediction / generate_pesudo_labels /
    neighbor_discrimination (:393-420), lifted from the AST (the module imports tensorflow) and run unchanged."""
    import scipy.sparse as sp
    from scipy.sparse import eye
    path = os.path.join(REF, "univariate", "sept_social.py")
    names = {"encoder", "social_encoder", "get_social_related_views", "label_prediction", "sampling",
             "generate_pesudo_labels", "neighbor_discrimination"}
    cls, ns = _lift_class_methods(path, "SEPT", names, extra_ns={"eye": eye}, top_level=("TFGraphInterface", "Graph"))
    rng = np.random.default_rng(99)
    n_u, n_i, d, n_layers = 50, 40, 64, 2
    s_r, s_c = rng.integers(0, n_u, 300), rng.integers(0, n_u, 300)
    keep = s_r != s_c
    s_r, s_c = s_r[keep], s_c[keep]
    s_r, s_c = np.concatenate([s_r, s_c[:120]]), np.concatenate([s_c, s_r[:120]])
    pairs = np.unique(np.stack([s_r, s_c], 1), axis=0)
    s_r, s_c = pairs[:, 0], pairs[:, 1]
    S = sp.csr_matrix((np.ones(len(s_r), dtype=np.float32), (s_r, s_c)), shape=(n_u, n_u), dtype=np.float32)
    bi = S.multiply(S)                                   # Relation.get_birectional_social_mat (:141-144)
    y_r = np.concatenate([np.arange(n_u), rng.integers(0, n_u, 400)])
    y_c = np.concatenate([rng.integers(0, n_i, n_u), rng.integers(0, n_i, 400)])
    ypairs = np.unique(np.stack([y_r, y_c], 1), axis=0)
    y_r, y_c = ypairs[:, 0], ypairs[:, 1]
    Y = sp.csr_matrix((np.ones(len(y_r), dtype=np.float32), (y_r, y_c)), shape=(n_u, n_i), dtype=np.float32)
    n = n_u + n_i
    adj = sp.bmat([[None, Y], [Y.T, None]], format="csr", dtype=np.float32)
    norm_adj = ns["Graph"].normalize_graph_mat(adj)

    m = cls()
    m.data = types.SimpleNamespace(user_num=n_u, item_num=n_i)
    m.social_data = types.SimpleNamespace(normalize_graph_mat=ns["Graph"].normalize_graph_mat)
    m.instance_cnt = 5
    social_mat, sharing_mat = m.get_social_related_views(bi, Y)
    to_t = ns["TFGraphInterface"].convert_sparse_mat_to_tensor
    g = torch.Generator().manual_seed(13)
    ego = torch.empty(n, d)
    torch.nn.init.xavier_uniform_(ego, generator=g)
    ego.requires_grad_(True)
    rec_u, rec_i = m.encoder(ego, to_t(norm_adj), n_layers)
    wgt = torch.randn(n, d, generator=g)
    (torch.cat([rec_u, rec_i]) * wgt).sum().backward()
    enc_grad = ego.grad.clone()
    users = ego.detach()[:n_u].clone().requires_grad_(True)
    friend = m.social_encoder(users, to_t(social_mat), n_layers)
    sharing = m.social_encoder(users, to_t(sharing_mat), n_layers)
    u_idx = torch.from_numpy(rng.integers(0, n_u, 64))
    m.aug_user_embeddings = (rec_u.detach() + 0.1 * torch.randn(n_u, d, generator=g))
    social_pred = m.label_prediction(friend.detach(), u_idx)
    sharing_pred = m.label_prediction(sharing.detach(), u_idx)
    f_pos = m.generate_pesudo_labels(sharing_pred, social_pred)
    emb_in = friend.detach().clone().requires_grad_(True)
    loss = m.neighbor_discrimination(f_pos, emb_in, u_idx)
    loss.backward()
    out = {"n_users": n_u, "n_items": n_i, "n_layers": n_layers, "ego": ego.detach().numpy(), "w": wgt.numpy(),
           "rec_user": rec_u.detach().numpy(), "rec_item": rec_i.detach().numpy(), "enc_grad": enc_grad.numpy(),
           "friend_view": friend.detach().numpy(), "sharing_view": sharing.detach().numpy(),
           "u_idx": u_idx.numpy(), "aug_user": m.aug_user_embeddings.numpy(), "positive": f_pos.numpy(),
           "nd_loss": np.float32(loss.item()), "nd_grad": emb_in.grad.numpy(),
           "S_row": s_r.astype(np.int64), "S_col": s_c.astype(np.int64),
           "Y_row": y_r.astype(np.int64), "Y_col": y_c.astype(np.int64)}
    out.update(_csr_dict("norm_adj", sp.csr_matrix(norm_adj)))
    out.update(_csr_dict("social", sp.csr_matrix(social_mat)))
    out.update(_csr_dict("sharing", sp.csr_matrix(sharing_mat)))
    np.savez_compressed(os.path.join(OUT, "sept_social.npz"), **out)
    print("wrote sept_social.npz: nd_loss", float(loss), "positives", tuple(f_pos.shape))


def gen_buir():
    """tests/golden/buir.npz: univariate/buir.py's LGCN_Encoder.sparse_dropout (:300-309) and the forward that
    consumes it (:311-326), the module imported as-is; the Bernoulli draw (`torch.rand(nnz)`) is recorded so
    that the test can hand the same keep mask to the HIP path."""
    import buir
    rng = np.random.default_rng(31)
    train = seeded_triples(rng, 40, 30, 350, 0)
    data = buir.Interaction({}, train, [])
    d, n_layers, rate = 64, 2, 0.3
    torch.manual_seed(17)
    enc = buir.LGCN_Encoder(data, d, n_layers, rate, drop_flag=True)
    adj = enc.sparse_norm_adj.coalesce() if not enc.sparse_norm_adj.is_coalesced() else enc.sparse_norm_adj
    nnz = enc.sparse_norm_adj._nnz()
    draws = []
    real_rand = torch.rand

    def recording_rand(*a, **kw):
        r = real_rand(*a, **kw)
        draws.append(r.numpy().copy())
        return r

    torch.rand = recording_rand
    try:
        dropped = enc.sparse_dropout(enc.sparse_norm_adj, rate, nnz)
    finally:
        torch.rand = real_rand
    x = torch.cat([enc.embedding_dict["user_emb"], enc.embedding_dict["item_emb"]], 0).detach()
    xr = x.clone().requires_grad_(True)
    layers = [xr]
    e = xr
    for _ in range(n_layers):                      # buir.py:315-320 on the dropped operator
        e = torch.sparse.mm(dropped, e)
        layers.append(e)
    final = torch.stack(layers, dim=1).mean(dim=1)
    wgt = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    (final * wgt).sum().backward()
    idx = enc.sparse_norm_adj._indices().numpy()
    dc = dropped.coalesce()
    np.savez_compressed(os.path.join(OUT, "buir.npz"), rate=rate, n_layers=n_layers, n_users=data.user_num,
                        n_items=data.item_num, adj_row=idx[0].astype(np.int64), adj_col=idx[1].astype(np.int64),
                        adj_val=enc.sparse_norm_adj._values().numpy(), rand=draws[0],
                        keep=np.floor(1 - rate + draws[0]).astype(bool),
                        dropped_row=dc.indices()[0].numpy(), dropped_col=dc.indices()[1].numpy(),
                        dropped_val=dc.values().numpy(), x=x.numpy(), w=wgt.numpy(),
                        final=final.detach().numpy(), grad=xr.grad.numpy())
    print("wrote buir.npz: nnz", nnz, "kept", int(np.floor(1 - rate + draws[0]).sum()))


def gen_featmask():
    """tests/golden/featmask.npz: univariate/grace.py's drop_feature (:261-267, what FeatureMasking.augment calls),
    lifted from the AST (the module imports torch_geometric) and run unchanged under a fixed torch seed."""
    ns = load_defs(os.path.join(REF, "univariate", "grace.py"), {"drop_feature"})
    g = torch.Generator().manual_seed(4)
    x = torch.randn(37, 64, generator=g)
    out = {}
    for k, pf in enumerate((0.0, 0.3, 0.7)):
        torch.manual_seed(100 + k)
        y = ns["drop_feature"](x, pf)
        out[f"pf{k}"], out[f"y{k}"] = np.float32(pf), y.numpy()
    assert torch.equal(x, torch.randn(37, 64, generator=torch.Generator().manual_seed(4)))      # input untouched
    np.savez_compressed(os.path.join(OUT, "featmask.npz"), x=x.numpy(), **out)
    print("wrote featmask.npz: dropped columns", [int((out[f"y{k}"] == 0).all(0).sum()) for k in range(3)])


if __name__ == "__main__":
    if "--out" in sys.argv:                        # write somewhere else (e.g. to diff a regeneration against tests/golden)
        OUT = sys.argv[sys.argv.index("--out") + 1]
        os.makedirs(OUT, exist_ok=True)
    if "--featmask" in sys.argv:
        gen_featmask()
    elif "--mhcn" in sys.argv:
        gen_mhcn()
    elif "--sept-social" in sys.argv:
        gen_sept_social()
    elif "--buir" in sys.argv:
        gen_buir()
    elif "--grace" in sys.argv:
        gen_grace()
    elif "--rownorm" in sys.argv:
        gen_rownorm()
    elif "--eval" in sys.argv:
        gen_eval()
    else:
        main()
