"""Shared by the CPU (gloo) and GPU two-rank tests of the row-sharded MHCN layer loop (BASELINE config 5):
a seeded synthetic problem, its single-process float64 reference (torch autograd on dense operators — a
restatement of oracle_np.mhcn_layer_loop, i.e. univariate/mhcn.py:422-466), and the per-rank partition."""
import numpy as np
import torch

N_USERS, N_ITEMS, D, LAYERS = 101, 37, 16, 2          # 101 users: not divisible by 2 -> padded user rows


def problem(seed=0):
    rng = np.random.default_rng(seed)

    def rand_rownorm(n_r, n_c, nnz):
        m = np.zeros((n_r, n_c))
        m[rng.integers(0, n_r, nnz), rng.integers(0, n_c, nnz)] = rng.random(nnz) + 0.1
        rs = m.sum(1, keepdims=True)
        return np.divide(m, rs, out=np.zeros_like(m), where=rs > 0)

    H = [rand_rownorm(N_USERS, N_USERS, 900), rand_rownorm(N_USERS, N_USERS, 700), rand_rownorm(N_USERS, N_USERS, 400)]
    R = rand_rownorm(N_USERS, N_ITEMS, 600)
    p = {"user": rng.standard_normal((N_USERS, D)) * 0.3, "item": rng.standard_normal((N_ITEMS, D)) * 0.3,
         "att": rng.standard_normal((1, D)) * 0.3, "att_mat": rng.standard_normal((D, D)) * 0.3,
         "wu": rng.standard_normal((N_USERS, D)), "wi": rng.standard_normal((N_ITEMS, D))}
    for c in range(4):
        p[f"gw{c}"] = rng.standard_normal((D, D)) * 0.3
        p[f"gb{c}"] = rng.standard_normal((1, D)) * 0.1
    return H, R, p


def reference(H, R, p):
    """Single-process float64 result: final user / item embeddings and the gradients of
    sum(final_user * wu) + sum(final_item * wi) w.r.t. every parameter."""
    t = {k: torch.tensor(v, dtype=torch.float64, requires_grad=k not in ("wu", "wi")) for k, v in p.items()}
    Ht = [torch.tensor(h) for h in H]
    Rt = torch.tensor(R)
    norm = lambda x: torch.nn.functional.normalize(x, p=2, dim=1)      # noqa: E731

    def gate(c):
        return t["user"] * torch.sigmoid(t["user"] @ t[f"gw{c}"] + t[f"gb{c}"])

    def attend(*e):
        w = torch.softmax(torch.stack([(t["att"] * (x @ t["att_mat"])).sum(1) for x in e]), 0)
        return sum(w[k].unsqueeze(1) * x for k, x in enumerate(e))

    c = [gate(0), gate(1), gate(2)]
    simple, items = gate(3), t["item"]
    sums = [c[0], c[1], c[2], simple, items]
    for _ in range(LAYERS):
        mixed = attend(*c) + simple / 2
        for k in range(3):
            c[k] = Ht[k] @ c[k]
            sums[k] = sums[k] + norm(c[k])
        new_items = Rt.T @ mixed
        sums[4] = sums[4] + norm(new_items)
        simple = Rt @ items
        sums[3] = sums[3] + norm(simple)
        items = new_items
    fu = attend(sums[0], sums[1], sums[2]) + sums[3] / 2
    fi = sums[4]
    ((fu * t["wu"]).sum() + (fi * t["wi"]).sum()).backward()
    grads = {k: v.grad.numpy() for k, v in t.items() if v.grad is not None}
    return fu.detach().numpy(), fi.detach().numpy(), grads


def coo_block(m, lo, hi, n_cols_pad):
    """Rows [lo, hi) of dense m as COO over n_cols_pad columns (local row ids)."""
    blk = m[lo:hi]
    r, c = np.nonzero(blk)
    return r.astype(np.int64), c.astype(np.int64), blk[r, c].astype(np.float32), n_cols_pad


def load_params(enc, p, lo, hi, device):
    n_loc = enc.user_num
    with torch.no_grad():
        u = np.zeros((n_loc, D), dtype=np.float32)
        u[: hi - lo] = p["user"][lo:hi]
        enc.user_embeddings.copy_(torch.from_numpy(u).to(device))
        enc.item_embeddings.copy_(torch.from_numpy(p["item"].astype(np.float32)).to(device))
        enc.attention.copy_(torch.from_numpy(p["att"].astype(np.float32)).to(device))
        enc.attention_mat.copy_(torch.from_numpy(p["att_mat"].astype(np.float32)).to(device))
        for c in range(4):
            enc.gating_weights[str(c + 1)].copy_(torch.from_numpy(p[f"gw{c}"].astype(np.float32)).to(device))
            enc.gating_bias[str(c + 1)].copy_(torch.from_numpy(p[f"gb{c}"].astype(np.float32)).to(device))


def run_rank(rank, world, device, make_graph, ops, group=None):
    """Builds rank's blocks with make_graph(row, col, val, n_rows, n_cols), runs propagate + backward of this
    rank's share of the loss, returns numpy results."""
    from recommendation_amd import distributed as gd
    from recommendation_amd.mhcn import ShardedMHCNEncoder
    H, R, p = problem()
    per_u = (N_USERS + world - 1) // world
    lo, hi = rank * per_u, min((rank + 1) * per_u, N_USERS)
    u_pad = per_u * world
    blocks = []
    for h in H:
        hp = np.zeros((u_pad, u_pad))
        hp[:N_USERS, :N_USERS] = h
        r, c, v, _ = coo_block(hp, rank * per_u, (rank + 1) * per_u, u_pad)
        blocks.append(make_graph(r, c, v, per_u, u_pad))
    rp = np.zeros((u_pad, N_ITEMS))
    rp[:N_USERS] = R
    r, c, v, _ = coo_block(rp, rank * per_u, (rank + 1) * per_u, N_ITEMS)
    r_local = make_graph(r, c, v, per_u, N_ITEMS)
    ch = gd.ShardedChannels(blocks, per_u, rank, world, group)
    enc = ShardedMHCNEncoder(ch, r_local, D, LAYERS, ops=ops)
    load_params(enc, p, lo, hi, device)
    fu, fi = enc.propagate()
    wu = np.zeros((per_u, D), dtype=np.float32)
    wu[: hi - lo] = p["wu"][lo:hi]
    # items are replicated: weight their term by 1 / world so that the ranks' shares sum to the full loss
    loss = (fu * torch.from_numpy(wu).to(device)).sum() + (fi * torch.from_numpy(p["wi"].astype(np.float32)).to(device)).sum() / world
    loss.backward()
    enc.allreduce_grads()
    out = {"fu": fu.detach().cpu().numpy()[: hi - lo], "fi": fi.detach().cpu().numpy(), "lo": lo, "hi": hi,
           "g_user": enc.user_embeddings.grad.cpu().numpy()[: hi - lo], "g_item": enc.item_embeddings.grad.cpu().numpy(),
           "g_att": enc.attention.grad.cpu().numpy(), "g_gw0": enc.gating_weights["1"].grad.cpu().numpy(),
           "g_gb3": enc.gating_bias["4"].grad.cpu().numpy()}
    return out


def check(results, world, rtol):
    H, R, p = problem()
    fu, fi, g = reference(H, R, p)
    for r in range(world):
        res = results[r]
        lo, hi = res["lo"], res["hi"]
        tol = dict(rtol=rtol, atol=rtol * np.abs(fu).max())
        np.testing.assert_allclose(res["fu"], fu[lo:hi], **tol)
        np.testing.assert_allclose(res["fi"], fi, **tol)
        np.testing.assert_allclose(res["g_user"], g["user"][lo:hi], rtol=rtol, atol=rtol * np.abs(g["user"]).max())
        np.testing.assert_allclose(res["g_item"], g["item"], rtol=rtol, atol=rtol * np.abs(g["item"]).max())
        np.testing.assert_allclose(res["g_att"], g["att"], rtol=rtol, atol=rtol * np.abs(g["att"]).max())
        np.testing.assert_allclose(res["g_gw0"], g["gw0"], rtol=rtol, atol=rtol * np.abs(g["gw0"]).max())
        np.testing.assert_allclose(res["g_gb3"], g["gb3"], rtol=rtol, atol=rtol * np.abs(g["gb3"]).max())
