"""Differentiable ops over libgcr (HIP kernels behind the C ABI of include/gcr.h).

Each op names the reference expression it stands in for.  Tensors must live on the GPU; a
missing/failed HIP library raises (there is deliberately no eager PyTorch fallback here).
"""
from __future__ import annotations

import torch

from . import _lib
from .graph import CsrGraph

SPMM_ROW_L2NORM = 1

# bench.py sets this to a list to receive a (start, end) HIP event pair per gcr_spmm_csr_f32
# launch, recorded on the stream the kernel is launched on
EVENT_SINK = None


def _check_dense(x, n_rows, name):
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != n_rows:
        raise ValueError(f"{name} must be float32 [{n_rows}, d], got {tuple(x.shape)} {x.dtype}")
    if not (1 <= x.shape[1] <= 256):
        raise ValueError("embedding dim must be in [1, 256]")


def spmm_into(graph: CsrGraph, x, *, y=None, acc_in=None, acc_out=None, acc_scale=1.0, val_scale=1.0,
              keep_bits=None, l2norm=False, inv_norm_out=None, acc_in2=None, acc_in2_scale=1.0, col_active_bits=None):
    """Raw launch of gcr_spmm_csr_acc2_f32: y = epilogue(val_scale * A[keep] x); optional fused layer
    combine acc_out = (acc_in + acc_in2_scale * acc_in2 + y) * acc_scale (lightgcn.py:26, ncl.py:421) and row L2
    normalise (sept.py:224).  col_active_bits: an int32 bitmap over the rows of x, clear = that row is zero (its
    non-zeros are skipped; `active_rows_bitmap`).  Outputs are caller-allocated; nothing is recorded for autograd."""
    _lib.require_cuda(x, y, acc_in, acc_out, keep_bits, inv_norm_out, acc_in2, col_active_bits)
    if col_active_bits is not None and (col_active_bits.dtype != torch.int32 or col_active_bits.numel() * 32 < graph.n_cols
                                        or keep_bits is not None):
        raise ValueError("col_active_bits must be an int32 bitmap with >= n_cols bits (and excludes keep_bits)")
    x = x.contiguous()
    _check_dense(x, graph.n_cols, "x")
    d = x.shape[1]
    for t, nm in ((y, "y"), (acc_in, "acc_in"), (acc_out, "acc_out"), (acc_in2, "acc_in2")):
        if t is not None:
            _check_dense(t, graph.n_rows, nm)
            if t.shape[1] != d or not t.is_contiguous():
                raise ValueError(f"{nm} must be contiguous [{graph.n_rows}, {d}]")
    if y is None and acc_out is None:
        raise ValueError("need y and/or acc_out")
    if acc_in2 is not None and acc_out is None:
        raise ValueError("acc_in2 needs acc_out")
    if keep_bits is not None and (keep_bits.dtype != torch.int32 or keep_bits.numel() * 32 < graph.nnz):
        raise ValueError("keep_bits must be an int32 bitmap with >= nnz bits")
    if inv_norm_out is not None and (inv_norm_out.dtype != torch.float32 or inv_norm_out.numel() != graph.n_rows):
        raise ValueError("inv_norm_out must be float32 [n_rows]")
    p = graph.plan
    ws = graph.workspace(d)
    sink = EVENT_SINK
    if sink is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rc = _lib.lib().gcr_spmm_csr_acc2_f32(
        _lib.dptr(p.desc), p.n_parts, _lib.dptr(p.long_row), _lib.dptr(p.long_slot0), p.n_long,
        _lib.dptr(graph.rowptr), _lib.dptr(graph.col), _lib.dptr(graph.val), _lib.dptr(keep_bits), float(val_scale),
        _lib.dptr(x), d, _lib.dptr(y), _lib.dptr(acc_in), _lib.dptr(acc_in2), float(acc_in2_scale), _lib.dptr(acc_out),
        float(acc_scale), SPMM_ROW_L2NORM if l2norm else 0, _lib.dptr(inv_norm_out), _lib.dptr(ws),
        graph.n_rows, graph.n_cols, _lib.dptr(col_active_bits), _lib.cur_stream(x.device))
    _lib.check(rc, "gcr_spmm_csr_acc2_f32")
    if sink is not None:
        ev1.record()
        sink.append((ev0, ev1))
    return y if y is not None else acc_out


def active_rows_bitmap(idx, n_rows):
    """int32 bitmap with bit r set for every r in `idx` (int64 ids; out-of-range ids ignored): the `col_active_bits` of a
    launch whose input is zero outside those rows (gcr_bitmap_set)."""
    idx = _as_index(idx, idx.device if isinstance(idx, torch.Tensor) else None).reshape(-1)
    _lib.require_cuda(idx)
    bits = torch.zeros((int(n_rows) + 31) // 32, dtype=torch.int32, device=idx.device)
    _lib.check(_lib.lib().gcr_bitmap_set(_lib.dptr(idx), idx.numel(), int(n_rows), _lib.dptr(bits), _lib.cur_stream(idx.device)),
               "gcr_bitmap_set")
    return bits


class _SpMM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, graph, keep_bits, keep_bits_t, val_scale):
        ctx.graph, ctx.keep_bits_t, ctx.val_scale = graph, keep_bits_t, val_scale
        y = torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device)
        return spmm_into(graph, x, y=y, keep_bits=keep_bits, val_scale=val_scale)

    @staticmethod
    def backward(ctx, dy):
        gt = ctx.graph.t
        dx = torch.empty(gt.n_rows, dy.shape[1], dtype=torch.float32, device=dy.device)
        spmm_into(gt, dy.contiguous(), y=dx, keep_bits=ctx.keep_bits_t, val_scale=ctx.val_scale)
        return dx, None, None, None, None


def spmm(graph: CsrGraph, x, keep_bits=None, keep_bits_t=None, val_scale=1.0, mask_symmetric=False):
    """Drop-in for `torch.sparse.mm(A, x)` (ncl.py:419 and the 13 other call sites of SURVEY §2.3
    S1), differentiable w.r.t. x: backward is the same kernel on A^T.  `keep_bits` is an edge-
    dropout bitmap in A's non-zero order, `keep_bits_t` the SAME mask in A^T's non-zero order.

    The backward computes (A o M)^T dy, so it needs the transposed mask even when A itself is
    symmetric: gcl.py:22-25 / buir.py:300-309 draw every stored (directed) non-zero independently, so
    M is not symmetric although A is.  For a symmetric graph `graph.mirror_perm()` maps each
    non-zero (r, c) to the position of (c, r): `edge_mask_bits(nnz, pe, seed, dev, edge_id=mirror)`
    (or `mirror_bits`) is the transposed mask.  Pass mask_symmetric=True only when (r, c) and (c, r)
    really share one draw (an undirected-edge mask); then keep_bits serves both directions."""
    if keep_bits is not None and keep_bits_t is None:
        if mask_symmetric and graph.symmetric:
            keep_bits_t = keep_bits
        elif x.requires_grad and torch.is_grad_enabled():
            raise ValueError("backward through a masked graph needs keep_bits_t (the mask in A^T's non-zero order); "
                             "for a symmetric graph build it with graph.mirror_perm(), or pass mask_symmetric=True "
                             "if the mask itself is symmetric")
    return _SpMM.apply(x, graph, keep_bits, keep_bits_t, float(val_scale))


def mirror_bits(keep_bits, mirror, nnz):
    """The bitmap `keep_bits` re-ordered by `mirror` (int64 [nnz]): bit e of the result = bit mirror[e] of
    the input — the transposed mask of a symmetric graph for masks that are not functions of a counter
    (e.g. a recorded draw).  Small one-off index plumbing (torch)."""
    shifts = torch.arange(32, device=keep_bits.device, dtype=torch.int32)
    flat = ((keep_bits.unsqueeze(1) >> shifts) & 1).reshape(-1)[:nnz]
    return pack_bits(flat[mirror].bool())


def pack_bits(keep_bool):
    """bool [nnz] -> little-endian int32 bitmap (bit e of word e // 32)."""
    n = keep_bool.numel()
    pad = (-n) % 32
    b = torch.cat([keep_bool.to(torch.int64), torch.zeros(pad, dtype=torch.int64, device=keep_bool.device)]).view(-1, 32)
    w = (b << torch.arange(32, device=b.device, dtype=torch.int64)).sum(1)
    return (w & 0xFFFFFFFF).to(torch.int64).where(w < 2 ** 31, w - 2 ** 32).to(torch.int32)


class _Propagate(torch.autograd.Function):
    """final = c * sum_{k=0..K} A^k x0 (c = 1 or 1/(K+1)), optionally also every layer output.
    Linear in x0, so backward is the same recurrence on A^T (Horner form):
        h_K = g_K + c g_final ; h_k = g_k + c g_final + A^T h_{k+1} ; dx0 = h_0."""

    @staticmethod
    def forward(ctx, x0, graph, n_layers, scale, want_layers):
        ctx.graph, ctx.n_layers, ctx.scale, ctx.want_layers = graph, n_layers, scale, want_layers
        x0 = x0.contiguous()
        if not want_layers and n_layers > 0:
            # Horner form  sum_k A^k x0 = x0 + A (x0 + A (x0 + ...)): every layer reads x0 and writes ONE
            # [N, d] array (no separate layer output + running sum): one N x d write less per layer
            z = x0
            for k in range(n_layers):
                nxt = torch.empty_like(x0)
                spmm_into(graph, z, acc_in=x0, acc_out=nxt, acc_scale=scale if k == n_layers - 1 else 1.0)
                z = nxt
            return z
        cur, acc = x0, x0
        layers = []
        for k in range(n_layers):
            last = k == n_layers - 1
            need_y = want_layers or not last
            y = torch.empty_like(x0) if need_y else None
            acc_out = torch.empty_like(x0) if k == 0 else acc  # in place after the first layer
            spmm_into(graph, cur, y=y, acc_in=acc, acc_out=acc_out, acc_scale=scale if last else 1.0)
            acc = acc_out
            if need_y:
                cur = y
                layers.append(y)
        if n_layers == 0:
            acc = x0 * scale
        return (acc, *layers) if want_layers else acc

    @staticmethod
    def backward(ctx, g_final, *g_layers):
        gt, K, c = ctx.graph.t, ctx.n_layers, ctx.scale
        g = g_final.contiguous()
        if K == 0:
            return g * c, None, None, None, None
        # the recurrence on h' = h / c: h'_K = g + g_K / c, h'_k = g + g_k / c + A^T h'_{k+1}, dx0 = c h'_0 — the scale
        # rides on the last launch's epilogue instead of a pass over g_final
        gl = [x.contiguous() if x is not None else None for x in g_layers] if ctx.want_layers else [None] * K
        inv_c = 1.0 / c
        # the last layer's own gradient is part of the first launch's INPUT, so it is the one addend that needs a pass;
        # every other g_k rides on an epilogue as the second addend (gcr_spmm_csr_acc2_f32)
        h = g if gl[K - 1] is None else torch.add(g, gl[K - 1], alpha=inv_c)
        for k in range(K - 1, 0, -1):
            out = torch.empty_like(g)
            spmm_into(gt, h, acc_in=g, acc_in2=gl[k - 1], acc_in2_scale=inv_c, acc_out=out)
            h = out
        dx0 = torch.empty_like(g)
        spmm_into(gt, h, acc_in=g, acc_out=dx0, acc_scale=c)
        return dx0, None, None, None, None


def lightgcn_propagate(graph: CsrGraph, x0, n_layers: int, combine: str = "mean", return_layers: bool = False):
    """K-layer LightGCN message pass with the layer combine fused into the SpMM epilogue.
    combine='mean' -> ncl.py:415-422 / selfcf.py:475-485 (mean of the K+1 layer outputs);
    combine='sum'  -> lightgcn.py:21-27 (x += out, no division, Q3).
    Returns final [N, d] (and the list [x0, A x0, ..., A^K x0] when return_layers)."""
    if combine not in ("mean", "sum"):
        raise ValueError("combine must be 'mean' or 'sum'")
    _lib.require_cuda(x0)
    _check_dense(x0, graph.n_cols, "x0")
    if graph.n_rows != graph.n_cols:
        raise ValueError("propagation needs a square operator")
    scale = 1.0 / (n_layers + 1) if combine == "mean" else 1.0
    out = _Propagate.apply(x0, graph, int(n_layers), scale, bool(return_layers))
    if return_layers:
        return out[0], [x0, *out[1:]]
    return out


def normalize_bwd_n(nrm, inv, g_n, g_raw=None, out=None, from_raw=False):
    """d(A x) of `F.normalize(A x)` from the saved normalised rows: (g_n - nrm <nrm, g_n>) * inv (+ g_raw) in ONE pass
    (gcr_normalize_bwd_n_f32; sept.py:223-224, mhcn.py:440-457).  Rows clamped by eps have inv = 1e12 and nrm = 0.
    `out` may be g_n or g_raw themselves (in place).  from_raw: `nrm` holds the RAW rows A x (n = raw * inv on the fly,
    gcr_normalize_bwd_raw_f32) — the forward of `spmm_dual_acc_into` keeps no normalised copy."""
    _lib.require_cuda(nrm, inv, g_n, g_raw, out)
    g_n = g_n.contiguous()
    g_raw = None if g_raw is None else g_raw.contiguous()
    if out is None:
        out = torch.empty_like(g_n)
    rows, d = nrm.shape
    if g_n.shape != nrm.shape or out.shape != nrm.shape or (g_raw is not None and g_raw.shape != nrm.shape) or d % 4 or d > 256:
        raise ValueError("normalize_bwd_n: [rows, d] float32 tensors of one shape, d a multiple of 4 and <= 256")
    fn = _lib.lib().gcr_normalize_bwd_raw_f32 if from_raw else _lib.lib().gcr_normalize_bwd_n_f32
    _lib.check(fn(_lib.dptr(nrm), _lib.dptr(inv), _lib.dptr(g_n), _lib.dptr(g_raw), rows, d, _lib.dptr(out),
                  _lib.cur_stream(nrm.device)), "gcr_normalize_bwd_raw_f32" if from_raw else "gcr_normalize_bwd_n_f32")
    return out


_GRAM_DIMS = (32, 64, 96, 128)


def gram_tn(x, g):
    """x^T g ([dx, dg]) for tall x [n, dx], g [n, dg] with the rows split over the whole chip (gcr_gram_tn_f32): the weight
    gradient of `em @ W` (mhcn.py:404-420).  Widths outside {32, 64, 96, 128} go to the library GEMM."""
    _lib.require_cuda(x, g)
    x, g = x.contiguous(), g.contiguous()
    n, dx = x.shape
    dg = g.shape[1]
    if dx not in _GRAM_DIMS or dg not in _GRAM_DIMS or g.shape[0] != n:
        return x.t() @ g
    L = _lib.lib()
    out = torch.empty(dx, dg, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(int(L.gcr_gram_tn_workspace_bytes(n, dx, dg)), 4) // 4, dtype=torch.float32, device=x.device)
    _lib.check(L.gcr_gram_tn_f32(_lib.dptr(x), _lib.dptr(g), n, dx, dg, _lib.dptr(out), _lib.dptr(ws), _lib.cur_stream(x.device)),
               "gcr_gram_tn_f32")
    return out


class _DenseProj(torch.autograd.Function):
    """em @ W for a tall em [n, d] and a small square-ish W (MHCN's gating / attention, mhcn.py:404-420): forward and
    d em stay library GEMMs (n is the large dimension: they tile well); dW = em^T g is `gram_tn`."""

    @staticmethod
    def forward(ctx, em, w):
        ctx.save_for_backward(em, w)
        return em @ w

    @staticmethod
    def backward(ctx, g):
        em, w = ctx.saved_tensors
        g = g.contiguous()
        gem = g @ w.t() if ctx.needs_input_grad[0] else None
        gw = gram_tn(em, g) if ctx.needs_input_grad[1] else None
        return gem, gw


def dense_proj(em, w):
    """`torch.matmul(em, W)` of mhcn.py:405,409,414 with the weight gradient computed by a row-split kernel."""
    if not em.is_cuda or em.dim() != 2 or w.dim() != 2 or em.dtype != torch.float32:
        return em @ w
    return _DenseProj.apply(em, w)


class _RowsDotVec(torch.autograd.Function):
    """em @ v for tall em [n, d] and v [d]: d em = g v^T (one write pass), d v = sum_r g_r em_r (gcr_weighted_colsum_f32:
    the library's transposed GEMV of this shape ran 1.1 ms at n = 250K)."""

    @staticmethod
    def forward(ctx, em, v):
        ctx.save_for_backward(em, v)
        v = v.contiguous()
        out = torch.empty(em.shape[0], dtype=torch.float32, device=em.device)
        _lib.check(_lib.lib().gcr_rows_dot_vec_f32(_lib.dptr(em), _lib.dptr(v), em.shape[0], em.shape[1], _lib.dptr(out),
                                                   _lib.cur_stream(em.device)), "gcr_rows_dot_vec_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        em, v = ctx.saved_tensors
        g = g.contiguous()
        gem = torch.outer(g, v) if ctx.needs_input_grad[0] else None
        gv = None
        if ctx.needs_input_grad[1]:
            L = _lib.lib()
            n, d = em.shape
            gv = torch.empty(d, dtype=torch.float32, device=em.device)
            ws = torch.empty(max(int(L.gcr_weighted_colsum_workspace_bytes(n, d)), 4) // 4, dtype=torch.float32, device=em.device)
            _lib.check(L.gcr_weighted_colsum_f32(_lib.dptr(em), _lib.dptr(g), n, d, _lib.dptr(gv), _lib.dptr(ws),
                                                 _lib.cur_stream(em.device)), "gcr_weighted_colsum_f32")
        return gem, gv


def rows_dot_vec(em, v):
    """`em @ v` ([n, d] x [d] -> [n]) with the row-split weighted column sum as the gradient of v."""
    if not em.is_cuda or em.dim() != 2 or v.dim() != 1 or em.dtype != torch.float32 or em.shape[1] > 256:
        return em @ v
    return _RowsDotVec.apply(em.contiguous(), v)


class _Gate(torch.autograd.Function):
    """em * sigmoid(z + bias) (mhcn.py:404-411 around the GEMM z = em W) as one pass forward (gcr_gate_fwd_f32) and one
    backward (gcr_gate_bwd_f32: d em, d z and the bias gradient; the sigmoid is recomputed from the saved z)."""

    @staticmethod
    def forward(ctx, em, z, bias):
        em, z = em.contiguous(), z.contiguous()
        b = None if bias is None else bias.reshape(-1).contiguous()
        out = torch.empty_like(em)
        n, d = em.shape
        _lib.check(_lib.lib().gcr_gate_fwd_f32(_lib.dptr(em), _lib.dptr(z), _lib.dptr(b), n, d, _lib.dptr(out),
                                               _lib.cur_stream(em.device)), "gcr_gate_fwd_f32")
        ctx.save_for_backward(em, z, b)
        ctx.bias_shape = None if bias is None else bias.shape
        return out

    @staticmethod
    def backward(ctx, g):
        em, z, b = ctx.saved_tensors
        L = _lib.lib()
        n, d = em.shape
        g = g.contiguous()
        d_em, d_z = torch.empty_like(em), torch.empty_like(z)
        d_b = torch.empty(d, dtype=torch.float32, device=em.device)
        ws = torch.empty(max(int(L.gcr_gate_bwd_workspace_bytes(n, d)), 4) // 4, dtype=torch.float32, device=em.device)
        _lib.check(L.gcr_gate_bwd_f32(_lib.dptr(g), _lib.dptr(em), _lib.dptr(z), _lib.dptr(b), n, d, _lib.dptr(d_em),
                                      _lib.dptr(d_z), _lib.dptr(d_b), _lib.dptr(ws), _lib.cur_stream(em.device)),
                   "gcr_gate_bwd_f32")
        return d_em, d_z, (None if ctx.bias_shape is None else d_b.reshape(ctx.bias_shape))


def gate(em, z, bias=None):
    """`em * torch.sigmoid(z + bias)` — mhcn.py:405-406 / 409-410 with z = em @ W; em, z float32 [n, d], bias [1, d] or [d]."""
    if not em.is_cuda or em.dim() != 2 or z.shape != em.shape or em.dtype != torch.float32 or z.dtype != torch.float32 \
            or em.shape[1] > 256 or (bias is not None and bias.numel() != em.shape[1]):
        return em * torch.sigmoid(z if bias is None else z + bias)
    return _Gate.apply(em, z, bias)


class _ChannelMix(torch.autograd.Function):
    """mhcn.py:413-420 on the logits e_k . v: (mixed, score) in one pass, the backward in one pass + the partial-sum
    reduction of d v (gcr_channel_mix_fwd_f32 / gcr_channel_mix_bwd_f32)."""

    @staticmethod
    def forward(ctx, e1, e2, e3, v, extra, extra_scale):
        L = _lib.lib()
        e1, e2, e3, v = e1.contiguous(), e2.contiguous(), e3.contiguous(), v.contiguous()
        ex = None if extra is None else extra.contiguous()
        n, d = e1.shape
        mixed = torch.empty_like(e1)
        score = torch.empty(3, n, dtype=torch.float32, device=e1.device)
        _lib.check(L.gcr_channel_mix_fwd_f32(_lib.dptr(e1), _lib.dptr(e2), _lib.dptr(e3), _lib.dptr(v), _lib.dptr(ex),
                                             float(extra_scale), n, d, _lib.dptr(mixed), _lib.dptr(score),
                                             _lib.cur_stream(e1.device)), "gcr_channel_mix_fwd_f32")
        ctx.save_for_backward(e1, e2, e3, v, score)
        ctx.extra_scale, ctx.has_extra = float(extra_scale), extra is not None
        ctx.mark_non_differentiable(score)
        return mixed, score

    @staticmethod
    def backward(ctx, g, _g_score):
        e1, e2, e3, v, score = ctx.saved_tensors
        L = _lib.lib()
        n, d = e1.shape
        g = g.contiguous()
        d1, d2, d3 = torch.empty_like(e1), torch.empty_like(e1), torch.empty_like(e1)
        dx = torch.empty_like(e1) if ctx.has_extra else None
        dv = torch.empty(d, dtype=torch.float32, device=e1.device)
        ws = torch.empty(max(int(L.gcr_channel_mix_bwd_workspace_bytes(n, d)), 4) // 4, dtype=torch.float32, device=e1.device)
        _lib.check(L.gcr_channel_mix_bwd_f32(_lib.dptr(g), _lib.dptr(e1), _lib.dptr(e2), _lib.dptr(e3), _lib.dptr(v),
                                             _lib.dptr(score), ctx.extra_scale, n, d, _lib.dptr(d1), _lib.dptr(d2),
                                             _lib.dptr(d3), _lib.dptr(dx), _lib.dptr(dv), _lib.dptr(ws),
                                             _lib.cur_stream(e1.device)), "gcr_channel_mix_bwd_f32")
        return d1, d2, d3, dv, dx, None


def channel_mix(e1, e2, e3, v, extra=None, extra_scale=0.0):
    """(mixed, score) of mhcn.py:413-420 for three channel tables [n, d] and the logit vector v [d]:
    score = softmax over the channels of e_k @ v ([3, n], not differentiable on this path), mixed = sum_k score_k e_k
    (+ extra_scale * extra).  Widths other than 32 / 64 / 128 / 256 and CPU tensors take the torch expression."""
    fused = e1.is_cuda and e1.dim() == 2 and e1.dtype == torch.float32 and e2.shape == e1.shape and e3.shape == e1.shape \
        and v.dim() == 1 and v.numel() == e1.shape[1] and (extra is None or extra.shape == e1.shape) \
        and bool(_lib.lib().gcr_channel_mix_supported(e1.shape[1]))
    if not fused:
        logits = torch.stack([rows_dot_vec(e, v) for e in (e1, e2, e3)])
        score = torch.softmax(logits, dim=0)
        mixed = score[0].unsqueeze(1) * e1 + score[1].unsqueeze(1) * e2 + score[2].unsqueeze(1) * e3
        return (mixed if extra is None else mixed + extra_scale * extra), score
    return _ChannelMix.apply(e1, e2, e3, v, extra, float(extra_scale))


class _NormProp(torch.autograd.Function):
    """One SEPT layer: y = normalize(A x) row-wise (sept.py:223-224); saves y and 1/||Ax||."""

    @staticmethod
    def forward(ctx, x, graph):
        y = torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device)
        inv = torch.empty(graph.n_rows, dtype=torch.float32, device=x.device)
        spmm_into(graph, x.contiguous(), y=y, l2norm=True, inv_norm_out=inv)
        ctx.graph = graph
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        # d(Ax) = (dy - y <y, dy>) / max(||Ax||, eps); rows clamped by eps have inv = 1e12 and y = 0
        dz = normalize_bwd_n(y, inv, dy)
        gt = ctx.graph.t
        dx = torch.empty(gt.n_rows, dy.shape[1], dtype=torch.float32, device=dy.device)
        spmm_into(gt, dz, y=dx)
        return dx, None


def spmm_l2norm(graph: CsrGraph, x):
    """`F.normalize(torch.sparse.mm(adj, emb), dim=1)` in one kernel, for encoders that feed the NORMALISED
    rows to the next layer (sept.py:223-224, sept_social.py:373-374,382-383).  MHCN feeds the raw product
    forward and only keeps the normalised copy: use `spmm_l2norm_dual` there."""
    _lib.require_cuda(x)
    return _NormProp.apply(x, graph)


def spmm_dual_into(graph: CsrGraph, x, y_raw, y_norm, inv_norm_out=None, keep_bits=None, val_scale=1.0):
    """Raw launch of gcr_spmm_csr_dual_f32: y_raw = A x and y_norm = normalize(A x) from one pass."""
    _lib.require_cuda(x, y_raw, y_norm, inv_norm_out, keep_bits)
    x = x.contiguous()
    _check_dense(x, graph.n_cols, "x")
    d = x.shape[1]
    for t, nm in ((y_raw, "y_raw"), (y_norm, "y_norm")):
        _check_dense(t, graph.n_rows, nm)
        if t.shape[1] != d or not t.is_contiguous():
            raise ValueError(f"{nm} must be contiguous [{graph.n_rows}, {d}]")
    if y_raw.data_ptr() == y_norm.data_ptr():
        raise ValueError("y_raw and y_norm must be different buffers")
    if keep_bits is not None and (keep_bits.dtype != torch.int32 or keep_bits.numel() * 32 < graph.nnz):
        raise ValueError("keep_bits must be an int32 bitmap with >= nnz bits")
    if inv_norm_out is not None and (inv_norm_out.dtype != torch.float32 or inv_norm_out.numel() != graph.n_rows):
        raise ValueError("inv_norm_out must be float32 [n_rows]")
    p = graph.plan
    rc = _lib.lib().gcr_spmm_csr_dual_f32(
        _lib.dptr(p.desc), p.n_parts, _lib.dptr(p.long_row), _lib.dptr(p.long_slot0), p.n_long,
        _lib.dptr(graph.rowptr), _lib.dptr(graph.col), _lib.dptr(graph.val), _lib.dptr(keep_bits), float(val_scale),
        _lib.dptr(x), d, _lib.dptr(y_raw), _lib.dptr(y_norm), _lib.dptr(inv_norm_out), _lib.dptr(graph.workspace(d)),
        graph.n_rows, graph.n_cols, _lib.cur_stream(x.device))
    _lib.check(rc, "gcr_spmm_csr_dual_f32")
    return y_raw, y_norm


def spmm_dual_acc_into(graph: CsrGraph, x, y_raw, acc_in, acc_out, inv_norm_out, y_norm=None, keep_bits=None, val_scale=1.0):
    """Raw launch of gcr_spmm_csr_dual_acc_f32: y_raw = A x, acc_out = acc_in + normalize(A x), inv_norm_out = 1 / |A x| per row
    from one pass (mhcn.py:440-457: the normalised product joins a layer list that is summed); y_norm optional."""
    _lib.require_cuda(x, y_raw, acc_in, acc_out, inv_norm_out, y_norm, keep_bits)
    x = x.contiguous()
    _check_dense(x, graph.n_cols, "x")
    d = x.shape[1]
    for t, nm in ((y_raw, "y_raw"), (acc_out, "acc_out"), (acc_in, "acc_in"), (y_norm, "y_norm")):
        if t is None:
            continue
        _check_dense(t, graph.n_rows, nm)
        if t.shape[1] != d or not t.is_contiguous():
            raise ValueError(f"{nm} must be contiguous [{graph.n_rows}, {d}]")
    if y_raw.data_ptr() == acc_out.data_ptr() or (y_norm is not None and y_norm.data_ptr() == y_raw.data_ptr()):
        raise ValueError("y_raw, y_norm and acc_out must be different buffers")
    if inv_norm_out is None or inv_norm_out.dtype != torch.float32 or inv_norm_out.numel() != graph.n_rows:
        raise ValueError("inv_norm_out must be float32 [n_rows]")
    p = graph.plan
    rc = _lib.lib().gcr_spmm_csr_dual_acc_f32(
        _lib.dptr(p.desc), p.n_parts, _lib.dptr(p.long_row), _lib.dptr(p.long_slot0), p.n_long,
        _lib.dptr(graph.rowptr), _lib.dptr(graph.col), _lib.dptr(graph.val), _lib.dptr(keep_bits), float(val_scale),
        _lib.dptr(x), d, _lib.dptr(y_raw), _lib.dptr(y_norm), _lib.dptr(acc_in), _lib.dptr(acc_out), _lib.dptr(inv_norm_out),
        _lib.dptr(graph.workspace(d)), graph.n_rows, graph.n_cols, _lib.cur_stream(x.device))
    _lib.check(rc, "gcr_spmm_csr_dual_acc_f32")
    return y_raw, acc_out


class _NormPropDual(torch.autograd.Function):
    """One MHCN channel layer: (z, n) = (A x, normalize(A x)) from one launch (univariate/mhcn.py:440-442:
    the RAW product feeds the next layer, the normalised copy joins the layer list)."""

    @staticmethod
    def forward(ctx, x, graph):
        z = torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device)
        n = torch.empty_like(z)
        inv = torch.empty(graph.n_rows, dtype=torch.float32, device=x.device)
        spmm_dual_into(graph, x, z, n, inv)
        ctx.graph = graph
        ctx.save_for_backward(n, inv)
        # an unused output (the raw rows of the last MHCN layer) arrives as None instead of a materialised [U, d] zero
        ctx.set_materialize_grads(False)
        return z, n

    @staticmethod
    def backward(ctx, gz, gn):
        n, inv = ctx.saved_tensors
        if gz is None and gn is None:
            return None, None
        # dz = gz + (gn - n <n, gn>) / max(||z||, eps); rows clamped by eps have inv = 1e12 and n = 0
        dz = normalize_bwd_n(n, inv, gn, gz) if gn is not None else gz
        gt = ctx.graph.t
        dx = torch.empty(gt.n_rows, dz.shape[1], dtype=torch.float32, device=dz.device)
        spmm_into(gt, dz.contiguous(), y=dx)
        return dx, None


def spmm_l2norm_dual(graph: CsrGraph, x):
    """(torch.sparse.mm(adj, emb), F.normalize(torch.sparse.mm(adj, emb), p=2, dim=1)) in one kernel —
    the MHCN layer step (univariate/mhcn.py:440-457), differentiable w.r.t. x through both outputs."""
    _lib.require_cuda(x)
    return _NormPropDual.apply(x, graph)


class _NormPropDualAcc(torch.autograd.Function):
    """(z, acc + normalize(z)), z = A x, from one launch (gcr_spmm_csr_dual_acc_f32): the layer-list accumulation of
    mhcn.py:440-457 folded into the product; saves z and 1 / |z| (no normalised copy)."""

    @staticmethod
    def forward(ctx, x, acc, graph):
        z = torch.empty(graph.n_rows, x.shape[1], dtype=torch.float32, device=x.device)
        out = torch.empty_like(z)
        inv = torch.empty(graph.n_rows, dtype=torch.float32, device=x.device)
        spmm_dual_acc_into(graph, x, z, acc.contiguous(), out, inv)
        ctx.graph = graph
        ctx.save_for_backward(z, inv)
        ctx.set_materialize_grads(False)
        return z, out

    @staticmethod
    def backward(ctx, gz, gs):
        z, inv = ctx.saved_tensors
        if gz is None and gs is None:
            return None, None, None
        dz = normalize_bwd_n(z, inv, gs, gz, from_raw=True) if gs is not None else gz
        gt = ctx.graph.t
        dx = torch.empty(gt.n_rows, dz.shape[1], dtype=torch.float32, device=dz.device)
        spmm_into(gt, dz.contiguous(), y=dx)
        return dx, gs, None


def spmm_l2norm_dual_acc(graph: CsrGraph, x, acc):
    """(A x, acc + F.normalize(A x, p=2, dim=1)) in one kernel: `spmm_l2norm_dual` with the running layer sum folded in."""
    _lib.require_cuda(x, acc)
    return _NormPropDualAcc.apply(x, acc, graph)


class _SplitRows(torch.autograd.Function):
    """(x[:n], x[n:]) of the stacked [users; items] table (ncl.py:422-423 `torch.split`).  Plain
    slicing back-propagates each half as zeros(N, d) + copy and then adds the two: three extra
    passes over the 280 MB table per use.  Here the backward is one concatenation."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n, ctx.rows = n, x.shape[0]
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g_top, g_bot):
        ref = g_top if g_top is not None else g_bot
        d = ref.shape[1]
        # a producer that wrote both halves into ONE [rows, d] buffer (bpr_sums on the two halves of a split table):
        # hand that buffer back instead of copying the halves together
        if g_top is not None and g_bot is not None:
            base = g_top._base
            if base is not None and base is g_bot._base and tuple(base.shape) == (ctx.rows, d) and base.is_contiguous() \
                    and g_top.data_ptr() == base.data_ptr() and tuple(g_top.shape) == (ctx.n, d) \
                    and g_bot.data_ptr() == base.data_ptr() + ctx.n * d * base.element_size() \
                    and g_top.is_contiguous() and g_bot.is_contiguous():
                return base, None
        if g_top is None:
            g_top = ref.new_zeros(ctx.n, d)
        if g_bot is None:
            g_bot = ref.new_zeros(ctx.rows - ctx.n, d)
        return torch.cat([g_top, g_bot], 0), None


def split_rows(x, n):
    """User / item halves of a stacked embedding table with a single-pass backward."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x[:n], x[n:]
    return _SplitRows.apply(x, int(n))


# ---------------------------------------------------------------------------------------------
# BPR pairwise loss (P1/P2)
# ---------------------------------------------------------------------------------------------
BPR_NCL, BPR_LOGSIGMOID, BPR_LOG_SIGMOID = 0, 1, 2


def _as_index(t, device):
    t = torch.as_tensor(t, device=device)
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t.contiguous()


# batches at least this large take the sorted backward (gcr_bpr_bwd_sorted_f32); below it the three
# row atomics per sample are cheaper than the sorts
BPR_SORTED_MIN_BATCH = 1 << 18
# full-batch edge-list BPR: order the negatives by item together with their (user, coefficient) payload (one 12-byte
# sort) instead of sorting an index and gathering through it
BPR_NEG_PAYLOAD_SORT = True
_ORDER_CACHE = {}


def _sorted_order(idx, n_keys, cache=False):
    """(keys_sorted uint32 as int32 tensor, perm int32) of a flat int64 index vector (gcr_sort_index).
    cache=True keeps the order of an index tensor that is reused step after step (lightgcn.py trains
    on the same edge arrays every step), keyed by storage pointer / length / version."""
    key = (idx.data_ptr(), idx.numel(), idx._version, int(n_keys))
    if cache and key in _ORDER_CACHE:
        return _ORDER_CACHE[key]
    L = _lib.lib()
    n = idx.numel()
    keys = torch.empty(n, dtype=torch.int32, device=idx.device)
    perm = torch.empty(n, dtype=torch.int32, device=idx.device)
    ws = torch.empty(int(L.gcr_sort_index_workspace_bytes(n)), dtype=torch.uint8, device=idx.device)
    _lib.check(L.gcr_sort_index(_lib.dptr(idx), n, int(n_keys), _lib.dptr(keys), _lib.dptr(perm), _lib.dptr(ws),
                                _lib.cur_stream(idx.device)), "gcr_sort_index")
    if cache:
        if len(_ORDER_CACHE) >= 8:
            _ORDER_CACHE.pop(next(iter(_ORDER_CACHE)))
        _ORDER_CACHE[key] = (keys, perm, idx)       # holding idx keeps the pointer from being recycled
    return keys, perm, idx


def _halves_of_one_table(top, bot):
    """True when `top` / `bot` are the two row blocks of ONE contiguous [rows, d] tensor (split_rows' outputs): decided
    from the views themselves — same base, shapes adding up to it — not from address adjacency, which two unrelated
    allocations can show by accident."""
    base = top._base
    return base is not None and base is bot._base and base.dim() == 2 and base.is_contiguous() \
        and top.is_contiguous() and bot.is_contiguous() and top.shape[1] == base.shape[1] == bot.shape[1] \
        and top.shape[0] + bot.shape[0] == base.shape[0] and top.data_ptr() == base.data_ptr() \
        and bot.data_ptr() == base.data_ptr() + top.numel() * base.element_size()


class _BprSums(torch.autograd.Function):
    """sums = [sum_b loss_b, sum|U[u]|^2, sum|I[i]|^2, sum|I[j]|^2] through gcr_bpr_fwd/bwd_f32."""

    @staticmethod
    def forward(ctx, user_tab, item_tab, u_idx, i_idx, j_idx, variant):
        L = _lib.lib()
        user_tab, item_tab = user_tab.contiguous(), item_tab.contiguous()
        batch = u_idx.numel()
        n_neg = 1 if j_idx.dim() == 1 else j_idx.shape[1]
        d = user_tab.shape[1]
        dev = user_tab.device
        dldx = torch.empty(max(batch, 1), dtype=torch.float32, device=dev)
        sums = torch.zeros(5, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.gcr_bpr_workspace_floats(batch)), dtype=torch.float32, device=dev)
        _lib.check(L.gcr_bpr_fwd_f32(_lib.dptr(user_tab), _lib.dptr(item_tab), d, _lib.dptr(u_idx), _lib.dptr(i_idx),
                                     _lib.dptr(j_idx), batch, n_neg, variant, user_tab.shape[0], item_tab.shape[0],
                                     _lib.dptr(dldx), _lib.dptr(sums), _lib.dptr(ws), _lib.cur_stream(dev)),
                   "gcr_bpr_fwd_f32")
        ctx.save_for_backward(user_tab, item_tab, u_idx, i_idx, j_idx, dldx)
        ctx.n_neg = n_neg
        # the two tables are the halves of one stacked [N, d] table (split_rows): the backward then writes both
        # gradients into one buffer, which _SplitRows.backward hands on without a copy
        ctx.stacked = _halves_of_one_table(user_tab, item_tab)
        ctx.mark_non_differentiable(u_idx, i_idx, j_idx)
        return sums

    @staticmethod
    def backward(ctx, g_sums):
        user_tab, item_tab, u_idx, i_idx, j_idx, dldx = ctx.saved_tensors
        gs = g_sums.contiguous().to(torch.float32)
        if ctx.stacked:
            gfull = torch.zeros(user_tab.shape[0] + item_tab.shape[0], user_tab.shape[1], dtype=torch.float32,
                                device=user_tab.device)
            gu, gi = gfull[: user_tab.shape[0]], gfull[user_tab.shape[0]:]
        else:
            gu = torch.zeros_like(user_tab)
            gi = torch.zeros_like(item_tab)
        batch = u_idx.numel()
        if batch >= BPR_SORTED_MIN_BATCH and batch * ctx.n_neg < 2 ** 31:
            ku, pu, _ = _sorted_order(u_idx, user_tab.shape[0], cache=True)
            ki, pi, _ = _sorted_order(i_idx, item_tab.shape[0], cache=True)
            L = _lib.lib()
            dev, stream = user_tab.device, _lib.cur_stream(user_tab.device)
            n_users, n_items, slots = user_tab.shape[0], item_tab.shape[0], batch * ctx.n_neg
            kj = pj = None
            if not BPR_NEG_PAYLOAD_SORT:
                kj, pj, _ = _sorted_order(j_idx.reshape(-1), n_items)             # fresh negatives every step
            _lib.check(L.gcr_bpr_bwd_sorted_f32(
                _lib.dptr(user_tab), _lib.dptr(item_tab), user_tab.shape[1], _lib.dptr(u_idx), _lib.dptr(i_idx),
                _lib.dptr(j_idx), batch, ctx.n_neg, n_users, n_items, _lib.dptr(dldx),
                _lib.dptr(gs), _lib.dptr(ku), _lib.dptr(pu), _lib.dptr(ki), _lib.dptr(pi), _lib.dptr(kj), _lib.dptr(pj),
                _lib.dptr(gu), _lib.dptr(gi), stream), "gcr_bpr_bwd_sorted_f32")
            if BPR_NEG_PAYLOAD_SORT:
                # the negatives' item rows: fresh every step, so their sort carries (user, coefficient) with the key and the
                # scatter streams sorted arrays (gcr_bpr_neg_items_sorted_f32) instead of gathering through a sorted index
                skey = torch.empty(slots, dtype=torch.int32, device=dev)
                spay = torch.empty(slots, dtype=torch.int64, device=dev)
                _lib.check(L.gcr_bpr_neg_block_f32(_lib.dptr(dldx), _lib.dptr(j_idx), _lib.dptr(u_idx), batch, ctx.n_neg, n_items,
                                                   _lib.dptr(gs), None, None, None, _lib.dptr(skey), _lib.dptr(spay), stream),
                           "gcr_bpr_neg_block_f32")
                ks, ps = torch.empty_like(skey), torch.empty_like(spay)
                ws = torch.empty(int(L.gcr_sort_pairs_u64_workspace_bytes(slots)), dtype=torch.uint8, device=dev)
                _lib.check(L.gcr_sort_pairs_u64(_lib.dptr(skey), _lib.dptr(spay), slots, n_items, _lib.dptr(ks), _lib.dptr(ps),
                                                _lib.dptr(ws), stream), "gcr_sort_pairs_u64")
                _lib.check(L.gcr_bpr_neg_items_sorted_f32(_lib.dptr(user_tab), _lib.dptr(item_tab), user_tab.shape[1],
                                                          _lib.dptr(ks), _lib.dptr(ps), slots, n_users, n_items, _lib.dptr(gs),
                                                          _lib.dptr(gi), stream), "gcr_bpr_neg_items_sorted_f32")
            return gu, gi, None, None, None, None
        _lib.check(_lib.lib().gcr_bpr_bwd_f32(
            _lib.dptr(user_tab), _lib.dptr(item_tab), user_tab.shape[1], _lib.dptr(u_idx), _lib.dptr(i_idx),
            _lib.dptr(j_idx), u_idx.numel(), ctx.n_neg, user_tab.shape[0], item_tab.shape[0], _lib.dptr(dldx),
            _lib.dptr(gs), _lib.dptr(gu), _lib.dptr(gi), _lib.cur_stream(user_tab.device)), "gcr_bpr_bwd_f32")
        return gu, gi, None, None, None, None


class _BprEdgeSums(torch.autograd.Function):
    """`bpr_sums` for the batch that IS the training graph's edge list (lightgcn.py:91-118 trains on every edge at once).
    Same forward launch; in the backward the positive-pair parts of both gradients,
        dU[u] += sum_{i in N(u)} c_ui I[i],   dI[i] += sum_{u in N(i)} c_ui U[u],   c_ui = dL/dx of the pair (u, i),
    are ONE launch of the SpMM kernel on the graph's own structure with the coefficients as values (the item-major half
    through `CsrGraph.mirror_perm`) instead of two sorted scatters over E samples; the negatives keep their sorted scatter."""

    @staticmethod
    def forward(ctx, table, graph, n_users, j_idx, variant):
        L = _lib.lib()
        table = table.contiguous()
        n_items = table.shape[0] - n_users
        u_idx, i_idx = graph.user_major_edges(n_users)
        batch = u_idx.numel()
        n_neg = 1 if j_idx.dim() == 1 else j_idx.shape[1]
        d = table.shape[1]
        dev = table.device
        user_tab, item_tab = table[:n_users], table[n_users:]
        dldx = torch.empty(max(batch, 1), dtype=torch.float32, device=dev)
        sums = torch.zeros(5, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.gcr_bpr_workspace_floats(batch)), dtype=torch.float32, device=dev)
        _lib.check(L.gcr_bpr_fwd_f32(_lib.dptr(user_tab), _lib.dptr(item_tab), d, _lib.dptr(u_idx), _lib.dptr(i_idx),
                                     _lib.dptr(j_idx), batch, n_neg, variant, n_users, n_items, _lib.dptr(dldx),
                                     _lib.dptr(sums), _lib.dptr(ws), _lib.cur_stream(dev)), "gcr_bpr_fwd_f32")
        ctx.save_for_backward(table, j_idx, dldx)
        ctx.graph, ctx.n_users, ctx.n_neg = graph, n_users, n_neg
        ctx.mark_non_differentiable(j_idx)
        return sums

    @staticmethod
    def backward(ctx, g_sums):
        table, j_idx, dldx = ctx.saved_tensors
        graph, n_users = ctx.graph, ctx.n_users
        n_items = table.shape[0] - n_users
        gs = g_sums.contiguous().to(torch.float32)
        u_idx, i_idx = graph.user_major_edges(n_users)
        batch = u_idx.numel()
        # positive pairs: G = C_sym [U; I], C_sym = the graph's pattern with c_ui at (u, i) and (i, u)
        val = torch.empty(2 * batch, dtype=torch.float32, device=table.device)
        dropped = torch.zeros(n_items, dtype=torch.float32, device=table.device)   # samples the forward dropped, per positive item
        _lib.check(_lib.lib().gcr_bpr_edge_values_f32(_lib.dptr(dldx), _lib.dptr(graph.mirror_perm()), _lib.dptr(graph.col),
                                                      n_users, batch, _lib.dptr(gs), _lib.dptr(val), _lib.dptr(dropped),
                                                      _lib.cur_stream(table.device)), "gcr_bpr_edge_values_f32")
        gfull = torch.empty_like(table)
        spmm_into(graph.with_values(val), table, y=gfull)
        gu, gi = gfull[:n_users], gfull[n_users:]
        # ... and the |I[i]|^2 term of the positives: 2 g_2 (number of live samples with item i) I[i]  (the users' |U[u]|^2
        # term rides in the launch below, which walks the samples user by user)
        live = graph.row_degrees()[n_users:] - dropped
        gi.addcmul_(table[n_users:], (2.0 * gs[2] * live).unsqueeze(1))
        # the negatives' USER side: the edges are in user-major order, so this step's negatives are a CSR block [U x I] on the
        # graph's own user row pointer (x n_neg) — one more SpMM on a cached work plan instead of a sorted scatter over E
        # samples (0.84 -> ~0.4 ms at cfg2); its |U[u]|^2 term counts the live samples of every user
        L = _lib.lib()
        n_neg = ctx.n_neg
        nb = graph.negatives_user_block(n_users, n_items, n_neg)
        ncol = torch.empty(batch * n_neg, dtype=torch.int32, device=table.device)
        nval = torch.empty(batch * n_neg, dtype=torch.float32, device=table.device)
        dropped_u = torch.zeros(n_users, dtype=torch.float32, device=table.device)
        payload_sort = BPR_NEG_PAYLOAD_SORT and batch * n_neg < 2 ** 31
        skey = spay = None
        if payload_sort:                     # the same walk also writes the item side's (key j, payload (u, value)) pairs
            skey = torch.empty(batch * n_neg, dtype=torch.int32, device=table.device)
            spay = torch.empty(batch * n_neg, dtype=torch.int64, device=table.device)
        _lib.check(L.gcr_bpr_neg_block_f32(_lib.dptr(dldx), _lib.dptr(j_idx), _lib.dptr(u_idx), batch, n_neg, n_items,
                                           _lib.dptr(gs), _lib.dptr(ncol), _lib.dptr(nval), _lib.dptr(dropped_u),
                                           _lib.dptr(skey), _lib.dptr(spay), _lib.cur_stream(table.device)),
                   "gcr_bpr_neg_block_f32")
        nb.col, nb.val = ncol, nval
        spmm_into(nb, table[n_users:], acc_in=gu, acc_out=gu)
        gu.addcmul_(table[:n_users], (2.0 * gs[1] * (graph.row_degrees()[:n_users] - dropped_u)).unsqueeze(1))
        # the negatives' item rows (fresh every step: one sort, one sorted scatter).  The sort carries (user, value) with
        # the key, so the scatter streams its three arrays instead of chasing perm -> sample -> (u_idx, dloss_dx)
        if payload_sort:
            ks, ps = torch.empty_like(skey), torch.empty_like(spay)
            ws = torch.empty(int(L.gcr_sort_pairs_u64_workspace_bytes(batch * n_neg)), dtype=torch.uint8, device=table.device)
            _lib.check(L.gcr_sort_pairs_u64(_lib.dptr(skey), _lib.dptr(spay), batch * n_neg, n_items, _lib.dptr(ks),
                                            _lib.dptr(ps), _lib.dptr(ws), _lib.cur_stream(table.device)), "gcr_sort_pairs_u64")
            _lib.check(L.gcr_bpr_neg_items_sorted_f32(_lib.dptr(table[:n_users]), _lib.dptr(table[n_users:]), table.shape[1],
                                                      _lib.dptr(ks), _lib.dptr(ps), batch * n_neg, n_users, n_items,
                                                      _lib.dptr(gs), _lib.dptr(gi), _lib.cur_stream(table.device)),
                       "gcr_bpr_neg_items_sorted_f32")
            return gfull, None, None, None, None
        kj, pj, _ = _sorted_order(j_idx.reshape(-1), n_items)
        _lib.check(L.gcr_bpr_bwd_sorted_f32(
            _lib.dptr(table[:n_users]), _lib.dptr(table[n_users:]), table.shape[1], _lib.dptr(u_idx), None,
            _lib.dptr(j_idx), batch, n_neg, n_users, n_items, _lib.dptr(dldx), _lib.dptr(gs), None,
            None, None, None, _lib.dptr(kj), _lib.dptr(pj), _lib.dptr(gu), _lib.dptr(gi),
            _lib.cur_stream(table.device)), "gcr_bpr_bwd_sorted_f32")
        return gfull, None, None, None, None


def bpr_edge_sums(graph, table, n_users, j_idx, variant=BPR_LOG_SIGMOID):
    """`bpr_sums(table[:U], table[U:], u_idx, i_idx, j_idx, variant)` for (u_idx, i_idx) = the non-zeros of the user rows
    of `graph` in CSR order (`graph.user_major_edges`), i.e. a full batch over the training edges the graph was built from
    (lightgcn.py:91-118).  `table`: the stacked [U + I, d] encoder output; j_idx: [E] or [E, n_neg] negatives in that
    order.  Same sums, same gradients (the sums in another order: ~1e-6 relative); the backward's positive-pair parts are
    one SpMM launch.  The graph must be a symmetric bipartite operator with the users first."""
    _lib.require_cuda(table)
    if not graph.symmetric:
        raise ValueError("bpr_edge_sums needs a symmetric bipartite operator (users first)")
    if table.dtype != torch.float32 or table.dim() != 2 or table.shape[0] != graph.n_rows:
        raise ValueError("table must be float32 [n_rows, d]")
    j_idx = _as_index(j_idx, table.device)
    e = graph.user_major_edges(n_users)[0].numel()
    if j_idx.shape[0] != e or j_idx.dim() > 2:
        raise ValueError("j_idx: [E] or [E, n_neg] with E = the number of training edges")
    if e < BPR_SORTED_MIN_BATCH or e * (1 if j_idx.dim() == 1 else j_idx.shape[1]) >= 2 ** 31:
        u_idx, i_idx = graph.user_major_edges(n_users)
        ue, ie = split_rows(table, n_users)
        return bpr_sums(ue, ie, u_idx, i_idx, j_idx, variant)
    return _BprEdgeSums.apply(table, graph, int(n_users), j_idx, int(variant))


def bpr_sums(user_tab, item_tab, u_idx, i_idx, j_idx, variant=BPR_NCL):
    """Fused gather + BPR + squared norms.  Returns a differentiable float32[5]:
    [sum_b loss_b, sum_b |U[u_b]|^2, sum_b |I[i_b]|^2, sum_bk |I[j_bk]|^2, #samples with a bad id].
    j_idx may be [B] or [B, n_neg] (lightgcn.py:93: negatives' scores are averaged)."""
    _lib.require_cuda(user_tab, item_tab)
    if user_tab.dtype != torch.float32 or item_tab.dtype != torch.float32 or user_tab.shape[1] != item_tab.shape[1]:
        raise ValueError("user_tab / item_tab must be float32 [*, d] with the same d")
    dev = user_tab.device
    u_idx, i_idx, j_idx = _as_index(u_idx, dev), _as_index(i_idx, dev), _as_index(j_idx, dev)
    if u_idx.dim() != 1 or i_idx.shape != u_idx.shape or j_idx.shape[0] != u_idx.shape[0] or j_idx.dim() > 2:
        raise ValueError("u_idx, i_idx: [B]; j_idx: [B] or [B, n_neg]")
    return _BprSums.apply(user_tab, item_tab, u_idx, i_idx, j_idx, int(variant))


# ---------------------------------------------------------------------------------------------
# negative sampler (N1) and edge-dropout bitmaps (A1)
# ---------------------------------------------------------------------------------------------
def neg_sample(user_rowptr, user_items_sorted, u_idx, n_negs, num_items, seed, offset=0, max_trials=101):
    """Uniform negatives with rejection against the user's sorted training row (device Philox).
    max_trials=0: no rejection (lightgcn.py:92); slots that exhaust max_trials come back as -1
    (ncl.py:110-112).  Returns int64 [B * n_negs]."""
    _lib.require_cuda(user_rowptr, u_idx)
    dev = u_idx.device
    u_idx = _as_index(u_idx, dev)
    out = torch.empty(u_idx.numel() * n_negs, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().gcr_neg_sample(_lib.dptr(user_rowptr), _lib.dptr(user_items_sorted), _lib.dptr(u_idx),
                                         u_idx.numel(), int(n_negs), user_rowptr.numel() - 1, int(num_items),
                                         int(seed) & (2 ** 64 - 1), int(offset) & (2 ** 64 - 1), int(max_trials),
                                         _lib.dptr(out), _lib.cur_stream(dev)), "gcr_neg_sample")
    return out


def edge_mask_bits(nnz, pe, seed, device, edge_id=None):
    """Bernoulli keep bitmap (`rand >= pe`, gcl.py:22-25) as int32 words, bit e <-> non-zero e.
    edge_id (int64 [nnz]) gives every non-zero a canonical id so that A and A^T share one mask."""
    bits = torch.zeros((nnz + 31) // 32, dtype=torch.int32, device=device)
    if edge_id is not None:
        edge_id = _as_index(edge_id, device)
    _lib.check(_lib.lib().gcr_edge_mask_bits(int(nnz), float(pe), int(seed) & (2 ** 64 - 1), _lib.dptr(edge_id),
                                             _lib.dptr(bits), _lib.cur_stream(device)), "gcr_edge_mask_bits")
    return bits


# ---------------------------------------------------------------------------------------------
# InfoNCE family (C1-C4): row logsumexp of the all-pairs logits on the fp32 MFMA
# ---------------------------------------------------------------------------------------------
_MFMA_DIMS = (32, 64, 128, 256)

# True: the column logsumexp of the symmetric loss (gcl.py:34) is computed by a second, bitwise
# reproducible pass instead of float atomics in the first one
COL_DETERMINISTIC = False


def _pad_dim(x):
    """Zero-pad the feature dim to the next width the MFMA kernels are built for (dot products
    and norms are unchanged by zero columns)."""
    d = x.shape[1]
    for w in _MFMA_DIMS:
        if d == w:
            return x
        if d < w:
            return torch.nn.functional.pad(x, (0, w - d))
    raise ValueError("embedding dim must be <= 256")


def row_inv_norm(x, eps=1e-12, out=None):
    """1 / max(||x_r||, eps) per row — F.normalize's denominator (ncl.py:127, gcl.py:29-30)."""
    _lib.require_cuda(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    elif out.shape != (x.shape[0],) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
        raise ValueError("row_inv_norm: out must be a contiguous float32 [rows] tensor on x's device")
    _lib.check(_lib.lib().gcr_row_inv_norm_f32(_lib.dptr(x), x.shape[0], x.shape[1], float(eps), _lib.dptr(out),
                                               _lib.cur_stream(x.device)), "gcr_row_inv_norm_f32")
    return out


INFONCE_EXCLUDE_DIAGONAL = 1
INFONCE_UNIT_ROWS = 2
INFONCE_ENGINE_F32 = 4

# "auto": the library default — split-operand engine for d <= 128 (its two-product launches on two f16 planes when
# the rows are normalised by the op itself, d <= 64 — flash forward and backward also d = 128: the
# GCR_INFONCE_UNIT_ROWS promise), f32 MFMA for d = 256;
# "b3": withhold the unit-rows promise (three bf16 planes everywhere); "f32": force the f32 MFMA.
# Read ONCE per forward (`_resolve_engine`); the backward reuses what the forward ran on.
INFONCE_ENGINE = "auto"


def _resolve_engine(engine=None, unit_rows=False):
    e = INFONCE_ENGINE if engine is None else engine
    if e not in ("auto", "b3", "f32"):
        raise ValueError("InfoNCE engine must be 'auto', 'b3' or 'f32'")
    if e == "f32":
        return INFONCE_ENGINE_F32
    return INFONCE_UNIT_ROWS if (unit_rows and e == "auto") else 0


def infonce_lse_raw(a, a_scale, b, b_scale, inv_tau, col_bound=None, exclude_diagonal=False, engine_flag=None):
    """lse[i] = log sum_j exp(inv_tau * a_scale[i] b_scale[j] <a_i, b_j>) (no autograd).
    col_bound (an upper bound of every logit, e.g. inv_tau for unit rows): also return the column
    logsumexp over the anchors [N] from the same pass (float atomics) as a second value.
    exclude_diagonal: the sum runs over j != i (grace.py:396-404, intra-view negatives)."""
    L = _lib.lib()
    m, d = a.shape
    n = b.shape[0]
    lse = torch.empty(m, dtype=torch.float32, device=a.device)
    col_sum = torch.empty(n, dtype=torch.float32, device=a.device) if col_bound is not None else None
    ws = torch.empty(max(int(L.gcr_infonce_fwd_workspace_bytes(m, n, d)), 8) // 4, dtype=torch.float32, device=a.device)
    _lib.check(L.gcr_infonce_fwd_ex_f32(_lib.dptr(a), _lib.dptr(a_scale), m, _lib.dptr(b), _lib.dptr(b_scale), n, d,
                                        float(inv_tau), _lib.dptr(lse), _lib.dptr(col_sum),
                                        float(col_bound) if col_bound is not None else 0.0, _lib.dptr(ws),
                                        (INFONCE_EXCLUDE_DIAGONAL if exclude_diagonal else 0) |
                                        (_resolve_engine() if engine_flag is None else engine_flag),
                                        _lib.cur_stream(a.device)), "gcr_infonce_fwd_ex_f32")
    if col_bound is None:
        return lse
    return lse, torch.log(col_sum) + float(col_bound)


# True: a row-softmax problem whose anchors need a gradient runs the flash-style forward (lse AND the
# softmax-weighted row sum o in one pass, gcr_infonce_fwd_o_f32): the anchor-side gradient is then
# dL/dlse * inv_tau * o and the backward recomputes the score tile once (table side) instead of twice.
FWD_O = True


def infonce_fwd_o_supported(d, engine_flag=0):
    return bool(_lib.lib().gcr_infonce_fwd_o_supported(int(d), int(engine_flag)))


def infonce_fwd_o_raw(a, a_scale, b, b_scale, inv_tau, exclude_diagonal=False, engine_flag=None, lse_out=None, o_out=None):
    """(lse [M], o [M, d]): lse[i] = log sum_j exp(s_ij), o[i] = sum_j softmax(s_i.)_j * (b_scale[j] b_j)
    (no autograd).  lse_out / o_out: caller-owned contiguous outputs (e.g. slices of one stacked buffer)."""
    L = _lib.lib()
    m, d = a.shape
    n = b.shape[0]
    lse = torch.empty(m, dtype=torch.float32, device=a.device) if lse_out is None else lse_out
    o = torch.empty(m, d, dtype=torch.float32, device=a.device) if o_out is None else o_out
    if lse.shape != (m,) or o.shape != (m, d) or lse.dtype != torch.float32 or o.dtype != torch.float32:
        raise ValueError("lse_out [M] / o_out [M, d] must be float32")
    ws = torch.empty(max(int(L.gcr_infonce_fwd_o_workspace_bytes(m, n, d)), 8) // 4, dtype=torch.float32, device=a.device)
    _lib.check(L.gcr_infonce_fwd_o_f32(_lib.dptr(a), _lib.dptr(a_scale), m, _lib.dptr(b), _lib.dptr(b_scale), n, d,
                                       float(inv_tau), _lib.dptr(lse), _lib.dptr(o), _lib.dptr(ws),
                                       (INFONCE_EXCLUDE_DIAGONAL if exclude_diagonal else 0) |
                                       (_resolve_engine() if engine_flag is None else engine_flag),
                                       _lib.cur_stream(a.device)), "gcr_infonce_fwd_o_f32")
    return lse, o


def pos_logit_raw(a, a_scale, b, b_scale, pos, scale, out=None):
    m, d = a.shape
    out = torch.empty(m, dtype=torch.float32, device=a.device) if out is None else out
    _lib.check(_lib.lib().gcr_pos_logit_f32(_lib.dptr(a), _lib.dptr(a_scale), _lib.dptr(b), _lib.dptr(b_scale),
                                            _lib.dptr(pos), m, b.shape[0], d, float(scale), _lib.dptr(out),
                                            _lib.cur_stream(a.device)), "gcr_pos_logit_f32")
    return out


def _infonce_bwd_raw(x, x_scale, y, y_scale, inv_tau, lse_x, w_x, lse_y, w_y, exclude_diagonal=False, engine_flag=None,
                     out=None):
    """g = inv_tau * sum_j P_ij yhat_j (see gcr_infonce_bwd_f32): gradient w.r.t. the scaled rows of x.
    out: caller-owned contiguous [Mx, d] (e.g. one half of a stacked gradient table)."""
    L = _lib.lib()
    mx, d = x.shape
    g = torch.empty_like(x) if out is None else out
    if g.shape != x.shape or g.dtype != torch.float32:
        raise ValueError("out must be float32 with x's shape")
    nbytes = int(L.gcr_infonce_bwd_workspace_bytes(mx, y.shape[0], d))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device) if nbytes else None
    _lib.check(L.gcr_infonce_bwd_ex_f32(_lib.dptr(x), _lib.dptr(x_scale), mx, _lib.dptr(y), _lib.dptr(y_scale), y.shape[0],
                                        d, float(inv_tau), _lib.dptr(lse_x), _lib.dptr(w_x), _lib.dptr(lse_y),
                                        _lib.dptr(w_y), _lib.dptr(g), _lib.dptr(ws),
                                        (INFONCE_EXCLUDE_DIAGONAL if exclude_diagonal else 0) |
                                        (_resolve_engine() if engine_flag is None else engine_flag),
                                        _lib.cur_stream(x.device)),
               "gcr_infonce_bwd_ex_f32")
    return g


class _InfoNCEStats(torch.autograd.Function):
    """(a, b) -> row_lse [M], pos_logit [M] (and col_lse [N]) of S = inv_tau * ahat bhat^T, with the
    flash-style HIP backward.  Every loss of the InfoNCE family is a few [M]-vector ops on top."""

    @staticmethod
    def forward(ctx, a, b, pos, inv_tau, normalize, want_col, exd=False, grad_a=False):
        a_p, b_p = _pad_dim(a).contiguous(), _pad_dim(b).contiguous()
        sa = row_inv_norm(a_p) if normalize else None
        sb = row_inv_norm(b_p) if normalize else None
        # unit-norm rows bound every logit by 1/tau: on the f32-MFMA engine row and column LSE come out
        # of ONE pass (the column sums by float atomics: 15.9 vs 21.0 ms at 100K x 100K);
        # COL_DETERMINISTIC forces the bitwise-reproducible second pass with the roles swapped, which
        # is also the path for un-normalised inputs — and for the split-operand engine, whose MFMA
        # work is cheap enough that the second pass costs no more than the column-sum epilogue
        # (two passes vs one: 11.5 vs 11.2 ms on three bf16 planes, 8.2 vs 13.6 ms on two f16 planes at 100K x 100K;
        # 0.34-0.49 vs 0.62-0.89 ms at 20K x 20K; scripts/perf_infonce_sym.py)
        eng = _resolve_engine(unit_rows=normalize)   # once per problem: the backward runs on the same engine
        on_f32 = bool(eng & INFONCE_ENGINE_F32) or _lib.lib().gcr_infonce_engine(a_p.shape[1]) == 0
        one_pass = want_col and normalize and inv_tau <= 40.0 and not COL_DETERMINISTIC and not exd and on_f32
        o = None
        if FWD_O and not want_col and grad_a and a_p.shape[0] > 0 and \
                infonce_fwd_o_supported(a_p.shape[1], eng):
            lse, o = infonce_fwd_o_raw(a_p, sa, b_p, sb, inv_tau, exclude_diagonal=exd, engine_flag=eng)
            col = None
        elif one_pass:
            lse, col = infonce_lse_raw(a_p, sa, b_p, sb, inv_tau, col_bound=inv_tau * 1.0001, engine_flag=eng)
        else:
            lse = infonce_lse_raw(a_p, sa, b_p, sb, inv_tau, exclude_diagonal=exd, engine_flag=eng)
            col = infonce_lse_raw(b_p, sb, a_p, sa, inv_tau, exclude_diagonal=exd, engine_flag=eng) if want_col else None
        pl = pos_logit_raw(a_p, sa, b_p, sb, pos, inv_tau)
        ctx.save_for_backward(a_p, b_p, pos, sa, sb, lse, col, o)
        ctx.inv_tau, ctx.d, ctx.exd, ctx.eng = inv_tau, a.shape[1], exd, eng
        if want_col:
            return lse, pl, col
        return lse, pl

    @staticmethod
    def backward(ctx, g_lse, g_pos, g_col=None):
        a, b, pos, sa, sb, lse, col, o = ctx.saved_tensors
        L = _lib.lib()
        inv_tau = ctx.inv_tau
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_lse = g_lse.contiguous().float() if g_lse is not None else None
        g_col = g_col.contiguous().float() if (g_col is not None and col is not None) else None
        lse_r = lse if g_lse is not None else None
        col_r = col if g_col is not None else None
        ga = gb = None
        stream = _lib.cur_stream(a.device)
        if need_a and o is not None:
            # flash-style forward kept o[i] = sum_j softmax_ij bhat_j: the softmax part of dL/dahat_i is one scale
            ga = o * (g_lse * inv_tau).unsqueeze(1) if g_lse is not None else torch.zeros_like(o)
        elif need_a:
            ga = _infonce_bwd_raw(a, sa, b, sb, inv_tau, lse_r, g_lse, col_r, g_col, ctx.exd, ctx.eng)
        if need_b:
            gb = _infonce_bwd_raw(b, sb, a, sa, inv_tau, col_r, g_col, lse_r, g_lse, ctx.exd, ctx.eng)
        if g_pos is not None and (need_a or need_b):
            _lib.check(L.gcr_infonce_pos_bwd_f32(_lib.dptr(a), _lib.dptr(sa), _lib.dptr(b), _lib.dptr(sb), _lib.dptr(pos),
                                                 _lib.dptr(g_pos.contiguous().float()), a.shape[0], b.shape[0], a.shape[1],
                                                 float(inv_tau), _lib.dptr(ga), _lib.dptr(gb), stream),
                       "gcr_infonce_pos_bwd_f32")
        if sa is not None:
            for x, s, g in ((a, sa, ga), (b, sb, gb)):
                if g is not None:
                    _lib.check(L.gcr_normalize_bwd_f32(_lib.dptr(x), _lib.dptr(s), _lib.dptr(g), x.shape[0], x.shape[1],
                                                       _lib.dptr(g), stream), "gcr_normalize_bwd_f32")
        d = ctx.d
        if ga is not None and ga.shape[1] != d:
            ga = ga[:, :d].contiguous()
        if gb is not None and gb.shape[1] != d:
            gb = gb[:, :d].contiguous()
        return ga, gb, None, None, None, None, None, None


def infonce_stats(a, b, pos=None, temperature=0.2, normalize=True, want_col=False, exclude_diagonal=False):
    """Row logsumexp and positive logit of S = (ahat @ bhat.T) / temperature without materialising S.
    a: [M, d] anchors, b: [N, d] candidates, pos: int64 [M] index of each anchor's positive row in b
    (None: the diagonal, requires N >= M).  Returns (lse [M], pos_logit [M]) and, with want_col,
    also the column logsumexp [N] (gcl.py:34 `cross_entropy(sim.T, labels)`).  exclude_diagonal: the
    sums leave out the pair (i, j = i) (grace.py:396-404).  Differentiable w.r.t. a and b."""
    _lib.require_cuda(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("a [M, d] and b [N, d] must be float32 with the same d")
    if b.shape[0] == 0:
        raise ValueError("empty candidate table")
    if pos is None:
        if b.shape[0] < a.shape[0]:
            raise ValueError("diagonal positives need N >= M")
    else:
        pos = _as_index(pos, a.device)
        if pos.shape != (a.shape[0],):
            raise ValueError("pos must be [M]")
    # the flash-style forward (lse + weighted row sum) only pays off when the anchors will get a gradient: decided
    # here, where the caller's grad mode is still visible (inside Function.forward it is always off)
    grad_a = torch.is_grad_enabled() and a.requires_grad
    return _InfoNCEStats.apply(a, b, pos, 1.0 / float(temperature), bool(normalize), bool(want_col), bool(exclude_diagonal),
                               grad_a)


# ---------------------------------------------------------------------------------------------
# BCE-with-logits over all pairs (lightgcn.py:109-113, `loss_type == "bce"`)
# ---------------------------------------------------------------------------------------------
BCE_TWO_PLANES = 8          # include/gcr.h GCR_BCE_TWO_PLANES


def _bce_flags(engine=None):
    """'auto': two f16 planes on rows scaled to unit norm inside the launch (scores un-scaled by the norms); 'b3': three bf16
    planes on the raw rows; 'f32': the f32 MFMA."""
    e = INFONCE_ENGINE if engine is None else engine
    if e not in ("auto", "b3", "f32"):
        raise ValueError("BCE engine must be 'auto', 'b3' or 'f32'")
    return {"auto": BCE_TWO_PLANES, "b3": 0, "f32": INFONCE_ENGINE_F32}[e]


def bce_fwd_raw(a, b, want_o=False, engine_flag=0):
    """(rowsum [M], o [M, d] or None): rowsum[i] = sum_j softplus(<a_i, b_j>), o[i] = sum_j sigmoid(<a_i, b_j>) b_j (no autograd)."""
    L = _lib.lib()
    m, d = a.shape
    n = b.shape[0]
    rows = torch.empty(m, dtype=torch.float32, device=a.device)
    o = torch.empty(m, d, dtype=torch.float32, device=a.device) if want_o else None
    ws = torch.empty(max(int(L.gcr_bce_fwd_workspace_bytes(m, n, d)), 8) // 4, dtype=torch.float32, device=a.device)
    _lib.check(L.gcr_bce_fwd_f32(_lib.dptr(a), m, _lib.dptr(b), n, d, _lib.dptr(rows), _lib.dptr(o), _lib.dptr(ws),
                                 int(engine_flag), _lib.cur_stream(a.device)), "gcr_bce_fwd_f32")
    return rows, o


def bce_bwd_raw(x, y, w_x=None, w_y=None, engine_flag=0, out=None):
    """g[i] = sum_j (w_x[i] | w_y[j]) sigmoid(<x_i, y_j>) y_j (exactly one of the weight vectors; gcr_bce_bwd_f32)."""
    L = _lib.lib()
    mx, d = x.shape
    g = torch.empty_like(x) if out is None else out
    nbytes = int(L.gcr_bce_bwd_workspace_bytes(mx, y.shape[0], d))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device) if nbytes else None
    _lib.check(L.gcr_bce_bwd_f32(_lib.dptr(x), mx, _lib.dptr(y), y.shape[0], d, _lib.dptr(w_x), _lib.dptr(w_y), _lib.dptr(g),
                                 _lib.dptr(ws), int(engine_flag), _lib.cur_stream(x.device)), "gcr_bce_bwd_f32")
    return g


class _BceRows(torch.autograd.Function):
    """(a, b) -> rowsum [M] of softplus(a b^T); the backward is sigmoid-weighted operand sums (no M x N matrix)."""

    @staticmethod
    def forward(ctx, a, b, eng, grad_a):
        a_p, b_p = _pad_dim(a).contiguous(), _pad_dim(b).contiguous()
        want_o = bool(grad_a and a_p.shape[0] > 0 and _lib.lib().gcr_bce_fwd_o_supported(a_p.shape[1], eng))
        rows, o = bce_fwd_raw(a_p, b_p, want_o, eng)
        ctx.save_for_backward(a_p, b_p, o)
        ctx.eng, ctx.d = eng, a.shape[1]
        return rows

    @staticmethod
    def backward(ctx, g):
        a, b, o = ctx.saved_tensors
        g = g.contiguous().float()
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = o * g.unsqueeze(1) if o is not None else bce_bwd_raw(a, b, w_x=g, engine_flag=ctx.eng)
        if ctx.needs_input_grad[1]:
            gb = bce_bwd_raw(b, a, w_y=g, engine_flag=ctx.eng)
        d = ctx.d
        if ga is not None and ga.shape[1] != d:
            ga = ga[:, :d].contiguous()
        if gb is not None and gb.shape[1] != d:
            gb = gb[:, :d].contiguous()
        return ga, gb, None, None


def bce_softplus_rowsum(a, b, engine=None):
    """rowsum[i] = sum_j softplus(<a_i, b_j>) = sum_j BCE-with-logits(s_ij, label 0) without materialising the [M, N]
    scores of lightgcn.py:110; differentiable w.r.t. a and b.  With a one-hot label per row the loss of lightgcn.py:113
    is (rowsum.sum() - sum_i s_{i, pos_i}) / (M N)  (`losses.lightgcn_bce_loss`)."""
    _lib.require_cuda(a, b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1] or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise ValueError("a [M, d] and b [N, d] must be float32 with the same d")
    if b.shape[0] == 0:
        raise ValueError("empty item table")
    return _BceRows.apply(a, b, _bce_flags(engine), torch.is_grad_enabled() and a.requires_grad)


class _BceEdgeLoss(torch.autograd.Function):
    """lightgcn.py:109-113 for the batch that IS the training graph's edge list (the full batch of lightgcn.py:86-87):
        loss = ( sum_u deg(u) rowsum_u  -  sum_{(u,i) in E} <U_u, I_i> ) / (E * I)
    * rows of the [E, I] score matrix that belong to the same user are equal: the softplus part runs over the U distinct
      users, weighted by their number of training edges (E / U times fewer pairs than the reference's matmul);
    * the positive-logit sum is 1/2 <T, A1 T> with A1 the graph's pattern with unit values (duplicate edges counted), and
      its gradient w.r.t. the stacked table T is A1 T itself: ONE SpMM launch gives the value and both gradients."""

    @staticmethod
    def forward(ctx, table, graph, n_users, eng):
        table = table.contiguous()
        ue, ie = table[:n_users], table[n_users:]
        n_edges = graph.user_major_edges(n_users)[0].numel()
        deg = graph.row_degrees()[:n_users].contiguous()
        want_o = bool(_lib.lib().gcr_bce_fwd_o_supported(table.shape[1], eng))
        rows, o = bce_fwd_raw(ue, ie, want_o, eng)
        ones = getattr(graph, "_unit_values", None)
        if ones is None:
            ones = graph if graph.val is None else \
                graph.with_values(torch.ones(graph.nnz, dtype=torch.float32, device=table.device))
            graph._unit_values = ones
        y = torch.empty_like(table)
        spmm_into(ones, table, y=y)
        scale = 1.0 / (float(n_edges) * float(ie.shape[0]))
        ctx.save_for_backward(table, o, y, deg)
        ctx.n_users, ctx.eng, ctx.scale = n_users, eng, scale
        return (torch.dot(deg, rows) - 0.5 * torch.dot(table.reshape(-1), y.reshape(-1))) * scale

    @staticmethod
    def backward(ctx, g):
        table, o, y, deg = ctx.saved_tensors
        n_users = ctx.n_users
        ue, ie = table[:n_users], table[n_users:]
        c = g * ctx.scale
        gt = torch.empty_like(table)
        w = (deg * c).contiguous()
        if o is not None:
            torch.mul(o, w.unsqueeze(1), out=gt[:n_users])
        else:
            bce_bwd_raw(ue, ie, w_x=w, engine_flag=ctx.eng, out=gt[:n_users])
        bce_bwd_raw(ie, ue, w_y=w, engine_flag=ctx.eng, out=gt[n_users:])
        torch.addcmul(gt, y, (-c).reshape(1, 1), out=gt)          # (c stays on the device: no host read-back)
        return gt, None, None, None


def bce_edge_loss(graph, table, n_users, engine=None):
    """`F.binary_cross_entropy_with_logits(user_emb[pos_u] @ item_emb.T, one_hot(pos_i))` (lightgcn.py:109-113) for
    (pos_u, pos_i) = the training edges `graph` was built from (`graph.user_major_edges`); `table` = the stacked [U + I, d]
    encoder output.  d must be one of the MFMA widths (32 / 64 / 128 / 256)."""
    _lib.require_cuda(table)
    if not graph.symmetric:
        raise ValueError("bce_edge_loss needs a symmetric bipartite operator (users first)")
    if table.dtype != torch.float32 or table.dim() != 2 or table.shape[0] != graph.n_rows:
        raise ValueError("table must be float32 [n_rows, d]")
    if table.shape[1] not in _MFMA_DIMS:
        raise ValueError("bce_edge_loss: embedding dim must be 32, 64, 128 or 256 (use losses.lightgcn_bce_loss otherwise)")
    return _BceEdgeLoss.apply(table, graph, int(n_users), _bce_flags(engine))


def edge_mask_exact_bits(nnz, n_keep, seed, device):
    """Keep bitmap with exactly n_keep of nnz bits set, a uniformly random subset without replacement
    (univariate/sept.py:55-61: `np.random.choice(idx, int(len(idx) * (1 - drop_rate)), replace=False)`)."""
    L = _lib.lib()
    bits = torch.empty((nnz + 31) // 32, dtype=torch.int32, device=device)
    ws = torch.empty(int(L.gcr_edge_mask_exact_workspace_bytes(nnz)), dtype=torch.uint8, device=device)
    _lib.check(L.gcr_edge_mask_exact_bits(int(nnz), int(n_keep), int(seed) & (2 ** 64 - 1), _lib.dptr(bits),
                                          _lib.dptr(ws), _lib.cur_stream(device)), "gcr_edge_mask_exact_bits")
    return bits


# ---------------------------------------------------------------------------------------------
# PyGCL feature masking (univariate/grace.py:261-278)
# ---------------------------------------------------------------------------------------------
class _MaskColumns(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, keep_bits):
        ctx.save_for_backward(keep_bits)
        return mask_columns_raw(x, keep_bits)

    @staticmethod
    def backward(ctx, g):
        (keep_bits,) = ctx.saved_tensors
        return mask_columns_raw(g, keep_bits), None


def mask_columns_raw(x, keep_bits):
    _lib.require_cuda(x, keep_bits)
    x = x.contiguous()
    if x.dim() != 2 or x.dtype != torch.float32 or keep_bits.dtype != torch.int32 or keep_bits.numel() * 32 < x.shape[1]:
        raise ValueError("x float32 [n, d]; keep_bits int32 bitmap with >= d bits")
    out = torch.empty_like(x)
    _lib.check(_lib.lib().gcr_mask_columns_f32(_lib.dptr(x), x.shape[0], x.shape[1], _lib.dptr(keep_bits), _lib.dptr(out),
                                               _lib.cur_stream(x.device)), "gcr_mask_columns_f32")
    return out


def feature_masking(x, pf, seed, keep_bits=None):
    """`drop_feature(x, pf)` / FeatureMasking(pf) (univariate/grace.py:261-278): every feature COLUMN is zeroed with
    probability pf (one draw per column: `uniform_(0, 1) < drop_prob` drops), x itself untouched; differentiable.
    The d draws come from the counter RNG (gcr_edge_mask_bits over the d column ids: keep = u_c >= pf); pass
    `keep_bits` to replay a recorded mask.  Returns (masked x, keep_bits)."""
    if keep_bits is None:
        keep_bits = edge_mask_bits(x.shape[1], pf, seed, x.device)
    return _MaskColumns.apply(x, keep_bits), keep_bits


# ---------------------------------------------------------------------------------------------
# batch-row gathers of the loss functions
# ---------------------------------------------------------------------------------------------
class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx):
        table = table.contiguous()
        out = torch.empty(idx.numel(), table.shape[1], dtype=torch.float32, device=table.device)
        _lib.check(_lib.lib().gcr_gather_rows_f32(_lib.dptr(table), _lib.dptr(idx), idx.numel(), table.shape[1], table.shape[0],
                                                  _lib.dptr(out), _lib.cur_stream(table.device)), "gcr_gather_rows_f32")
        ctx.save_for_backward(idx)
        ctx.shape = table.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gt = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        g = g.contiguous()
        _lib.check(_lib.lib().gcr_scatter_add_rows_f32(_lib.dptr(g), _lib.dptr(idx), idx.numel(), g.shape[1], ctx.shape[0],
                                                       _lib.dptr(gt), _lib.cur_stream(g.device)), "gcr_scatter_add_rows_f32")
        return gt, None


# True: gather_rows checks its ids on the host (one min / max read-back per call) and raises IndexError like `table[idx]`
# does; off by default because the read-back stalls the stream (and forbids hipGraph capture)
CHECK_INDEX = False


def gather_rows(table, idx):
    """`table[idx]` for a float32 [N, d] table and int64 ids in [0, N) — NOT torch's full indexing contract: negative
    ids do not wrap, and an id outside [0, N) yields a zero row and is skipped by the backward instead of raising (set
    `functional.CHECK_INDEX = True` to get torch's IndexError, at the price of a host read-back).  The backward adds
    duplicate ids with float atomics: their order, hence the last bits of such rows, can differ from run to run.

    `table[idx]` for a float32 [N, d] table and int64 ids (`rec_user_emb[user_idx]`, `context[user]` ...; ncl.py:314-316,
    360-361,370-373) whose backward scatters the row gradients with float atomics (gcr_scatter_add_rows_f32) instead of
    the sort + segmented reduction of a generic index_put(accumulate=True) (which grows with the batch; at B = 2048
    the two cost the same within noise, profiles/r02_ncl_step_kernel_stats.csv)."""
    _lib.require_cuda(table)
    if table.dim() != 2 or table.dtype != torch.float32:
        raise ValueError("table must be float32 [N, d]")
    idx = _as_index(idx, table.device).reshape(-1)
    if CHECK_INDEX and idx.numel() > 0:
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= table.shape[0]:
            raise IndexError(f"gather_rows: index {lo if lo < 0 else hi} is out of bounds for a table of {table.shape[0]} rows")
    return _GatherRows.apply(table, idx)
