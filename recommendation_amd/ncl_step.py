"""One NCL training iteration (the loop body ncl.py:311-329) as a fixed, hand-derived launch sequence.

`NCLModel.train_step` through autograd is the general path (any loss mix); this is the same arithmetic for the
reference's fixed recipe — LightGCN propagation + BPR + structure contrast + per-batch e_step + prototype contrast +
Adam — with the backward written out, so that no gradient ever takes a detour through a dense zero-filled [N, d]
buffer or a separate add pass:

  forward   y_k = A y_{k-1} (gcr_spmm_csr_acc2_f32, running mean fused), BPR sums (gcr_bpr_fwd_f32), structure contrast
            as the flash-style forward (lse + softmax-weighted row sum, gcr_infonce_fwd_o_f32), e_step (kmeans.py, no host
            read-back), prototype contrast (B x B)
  backward  G0  = d loss / d x0 through the layer-0 table of the structure contrast: ONE [N, d] buffer written by the two
                  table-side launches (users -> rows [0, U), items -> rows [U, N)), positives and F.normalize's backward
                  in place, the prototype rows' gradients scattered on top                       (ncl.py:358-375)
            Zg  = d loss / d final from BPR, scattered into the step's only zero-filled buffer      (ncl.py:314-317)
            Horner recurrence of the K-layer pass on A^T:  h_K = Zg (+ g_ctx / c if the context layer is the last),
                  h_k = Zg + A^T h_{k+1} — the FIRST launch with a column bitmap of the <= 3 B non-zero rows of h_K, so it
                  gathers almost nothing — with the context rows' sparse gradient ADDED AFTER the launch that produces
                  its layer's h (4096 row atomics instead of a dense addend), dx0 = c (Zg + G0 / c + A^T h_1): the last
                  launch takes G0 as the epilogue's second addend                                   (ncl.py:415-422)
            Adam (gcr_adam_step_f32 / _dev_f32) on the two parameter views of the stacked table      (ncl.py:305,327-329)

Nothing in here synchronises with the host or depends on host-side values that change between steps, so the whole step
can be captured in a hipGraph (`capture()`): batch indices live in static buffers, Adam's step count on the device.
Checked against the autograd path by tests/test_ncl_step_gpu.py (same losses, same parameter update).
"""
from __future__ import annotations

import torch

from . import _lib
from . import functional as Fn


class FusedNCLStep:
    def __init__(self, owner, optimizer):
        """owner: NCLModel (reads n_layers / hyper_layers / ssl_temp / ssl_reg / alpha / proto_reg / reg / batch_size and
        calls owner.e_step); optimizer: FusedAdam over owner.model.parameters()."""
        self.owner, self.opt = owner, optimizer
        enc = owner.model
        self.enc, self.graph = enc, enc.norm_adj
        self.n_u, self.n_i = owner.data.user_num, owner.data.item_num
        self._cuda_graph = None
        self._static = None
        # False: leave the per-batch e_step (ncl.py:324) out and contrast against the centroids the owner already holds —
        # NOT the reference's loop body; bench.py uses it to report the e_step's share of the step
        self.e_step_every_batch = True
        # True: the table-side gradient branch of the backward runs on a side stream beside the SpMM recurrence
        self.overlap_backward = True
        self._side = None
        # True: the structure contrast's forward runs on the side stream beside the message-pass launches that follow its
        # context layer
        self.overlap_forward = True
        self.side_priority = None              # None: -1 when the owner replays a captured step, else 0
        # True: the e_step's streams are joined only in front of the prototype contrast (the structure contrast's forward
        # is issued meanwhile)
        self.early_e_step = True
        # "late": the e_step's launches are issued behind the backward recurrence (ordered behind `final` by an event);
        # "early": right after the forward (scripts/exp/ncl_step_quick.py compares them)
        self.e_step_issue = "late"

    def _side_stream(self, dev):
        if self._side is None:
            # high priority inside a captured step: its kernels are few and big-tiled (one 512-thread workgroup per CU),
            # and a gather kernel that got there first holds every wave slot until its grid drains (cfg3, replay: 7.93 ->
            # 7.69 ms without / 9.85 -> 9.69 ms with the e_step).  NOT in eager mode: beside the e_step's ~150 small
            # launches on their own two streams a high-priority stream made the step 12-15 ms instead of 9.9.
            pr = self.side_priority if self.side_priority is not None else (-1 if self.owner.graph_capture else 0)
            self._side = torch.cuda.Stream(device=dev, priority=pr)
        return self._side

    @staticmethod
    def supported(owner):
        enc = owner.model
        d, k = enc.latent_size, enc.layers
        if not hasattr(enc, "table") or d not in Fn._MFMA_DIMS or k < 1 or owner.hyper_layers < 1:
            return False
        if not enc.table.is_cuda:
            return False
        eng = Fn._resolve_engine(unit_rows=True)
        return Fn.infonce_fwd_o_supported(d, eng) and 1.0 / float(owner.ssl_temp) > 0

    # ------------------------------------------------------------------------------------------------------------
    def _contrast_fwd(self, a, sa, b, sb, pos, inv_tau, eng, lse, o, pl):
        Fn.infonce_fwd_o_raw(a, sa, b, sb, inv_tau, engine_flag=eng, lse_out=lse, o_out=o)
        Fn.pos_logit_raw(a, sa, b, sb, pos, inv_tau, out=pl)

    def __call__(self, user_idx, pos_idx, neg_idx):
        """user_idx / pos_idx / neg_idx: int64 device tensors [B] (a batch of next_batch_pairwise).  Returns
        (rec_loss, ssl_loss, proto_loss, total) as device scalars; the parameters are updated."""
        o_ = self.owner
        enc, graph, n_u, n_i = self.enc, self.graph, self.n_u, self.n_i
        L = _lib.lib()
        if self._cuda_graph is not None and not enc._aliased():
            raise RuntimeError("the encoder's parameters were re-bound (deepcopy / load_state_dict(assign=True) / .data = ...) "
                               "after this step was captured: call capture() again")
        x0 = enc.restack()                 # the parameters' storage IS the table (re-stacked if something replaced it)
        dev = x0.device
        n, d = x0.shape
        K = enc.layers
        c = 1.0 / (K + 1)
        inv_c = float(K + 1)
        ci = K if o_.hyper_layers * 2 >= K + 1 else o_.hyper_layers * 2          # ncl.py:319-322
        stream = _lib.cur_stream(dev)
        bsz = user_idx.numel()
        inv_tau = 1.0 / float(o_.ssl_temp)
        eng = Fn._resolve_engine(unit_rows=True)
        gat = torch.cat([user_idx, pos_idx + n_u])                                # rows of the stacked table

        # ---- structure contrast, forward (ncl.py:358-367): batch rows of the context layer against ALL layer-0 rows.  It
        # needs the context layer only (2 of K = 3 launches at the reference's hyper_layers = 1): with `overlap_forward`
        # it runs on the side stream beside the remaining launches of the message pass — matrix-core work next to an
        # HBM-bound gather — and is joined where its losses are summed ----
        rows_c = torch.empty(2 * bsz, d, dtype=torch.float32, device=dev)
        sa = torch.empty(2 * bsz, dtype=torch.float32, device=dev)
        sb = torch.empty(n, dtype=torch.float32, device=dev)
        lse = torch.empty(2 * bsz, dtype=torch.float32, device=dev)
        pl = torch.empty(2 * bsz, dtype=torch.float32, device=dev)
        o = torch.empty(2 * bsz, d, dtype=torch.float32, device=dev)
        xu, xi, sbu, sbi = x0[:n_u], x0[n_u:], sb[:n_u], sb[n_u:]

        Fn.row_inv_norm(x0, out=sb)                                                # F.normalize(iu) / F.normalize(ii)

        def contrast_rows(ctx):                                                    # the small launches stay on the main stream:
            _lib.check(L.gcr_gather_rows_f32(_lib.dptr(ctx), _lib.dptr(gat), 2 * bsz, d, n, _lib.dptr(rows_c), stream),
                       "gcr_gather_rows_f32")                                      # the fork releases the two big ones together
            Fn.row_inv_norm(rows_c, out=sa)

        def structure_contrast():
            self._contrast_fwd(rows_c[:bsz], sa[:bsz], xu, sbu, user_idx, inv_tau, eng, lse[:bsz], o[:bsz], pl[:bsz])
            self._contrast_fwd(rows_c[bsz:], sa[bsz:], xi, sbi, pos_idx, inv_tau, eng, lse[bsz:], o[bsz:], pl[bsz:])

        main = torch.cuda.current_stream(dev)
        fwd_side = self._side_stream(dev) if (self.overlap_forward and ci < K) else None

        # ---- forward: K-layer message pass, mean of the K + 1 layer outputs in the epilogue (ncl.py:415-422) ----
        cur, acc, ctx_layer = x0, x0, None
        for k in range(1, K + 1):
            last = k == K
            need_y = (not last) or ci == K
            y = torch.empty_like(x0) if need_y else None
            acc_out = torch.empty_like(x0) if k == 1 else acc
            Fn.spmm_into(graph, cur, y=y, acc_in=acc, acc_out=acc_out, acc_scale=c if last else 1.0)
            acc = acc_out
            if need_y:
                cur = y
            if k == ci:
                ctx_layer = y
                if fwd_side is not None:
                    contrast_rows(ctx_layer)
                    fwd_side.wait_stream(main)
                    with torch.cuda.stream(fwd_side):
                        structure_contrast()
        final = acc
        fu, fi = final[:n_u], final[n_u:]

        # ---- e_step (ncl.py:324, every batch): k-means of the CURRENT encoder outputs, no host read-back, on its own two
        # streams (NCLModel.e_step).  It only has to be back for the prototype contrast, which `early_e_step` places behind
        # the backward recurrence (nothing else in the step reads the centroids).  It waits for `final` (this event), but
        # is ISSUED behind the backward's big launches: its ~150 short launches in front of them kept them waiting —
        # in eager mode for the host, and in a replayed graph too (its nodes are dispatched in capture order) ----
        ev_final = None
        if self.e_step_every_batch:
            if self.e_step_issue == "early":
                # issued right here, joined in front of the prototype contrast: the chain runs beside everything in between
                o_.e_step(fu, fi, assign_all=False, join=False)
            elif self.early_e_step:
                ev_final = torch.cuda.Event()
                ev_final.record(main)
            else:
                o_.e_step(fu, fi, assign_all=False, join=True)

        # ---- BPR + the three squared norms of l2_reg_loss (ncl.py:314-317,116-123) ----
        dldx = torch.empty(max(bsz, 1), dtype=torch.float32, device=dev)
        sums = torch.zeros(5, dtype=torch.float32, device=dev)
        ws = torch.empty(int(L.gcr_bpr_workspace_floats(bsz)), dtype=torch.float32, device=dev)
        _lib.check(L.gcr_bpr_fwd_f32(_lib.dptr(fu), _lib.dptr(fi), d, _lib.dptr(user_idx), _lib.dptr(pos_idx), _lib.dptr(neg_idx),
                                     bsz, 1, Fn.BPR_NCL, n_u, n_i, _lib.dptr(dldx), _lib.dptr(sums), _lib.dptr(ws), stream),
                   "gcr_bpr_fwd_f32")
        rec_loss = sums[0] / bsz
        roots = sums[1:4].sqrt()
        l2 = o_.reg * roots.sum() / bsz                                            # l2_reg_loss(reg, u, p, n)
        # d total / d sums: total = sums[0] / B + reg (sqrt s1 + sqrt s2 + sqrt s3) / B / batch_size + ...
        # (`t[i] = python_scalar` on a device tensor is a host-to-device copy, which a stream capture refuses: fill_ views)
        gs = torch.zeros(5, dtype=torch.float32, device=dev)
        gs[0:1].fill_(1.0 / bsz)
        gs[1:4].copy_((0.5 * o_.reg / (bsz * o_.batch_size)) / roots)

        # ---- structure contrast: joined (or run here) ----
        if fwd_side is not None:
            main.wait_stream(fwd_side)
        else:
            contrast_rows(ctx_layer)
            structure_contrast()
        w = torch.empty(2 * bsz, dtype=torch.float32, device=dev)                  # d total / d lse  (= - d total / d pos)
        w[:bsz].fill_(float(o_.ssl_reg))
        w[bsz:].fill_(float(o_.ssl_reg * o_.alpha))
        ssl_loss = ((lse - pl) * w).sum()

        # =================================== backward ===================================
        # Two independent branches meet in the LAST launch of the Horner recurrence:
        #   branch T (matrix-core bound, ~1.7 ms at cfg3): G0 = gradient w.r.t. the layer-0 table through the structure
        #            contrast's candidate side (+ positives, F.normalize backward); the prototype rows join it at the end
        #   branch S (HBM bound, ~1.0 ms): anchor-side gradients, Zg from BPR, every launch of the recurrence but the last
        # With `overlap_backward` branch T runs on a side stream beside branch S (the table-side kernel leaves one wave slot
        # per SIMD that an SpMM wave fits into); the join is in front of the launch that takes G0 as its second addend.
        g0 = torch.empty_like(x0)
        side = self._side_stream(dev) if (self.overlap_backward and K > 1) else None

        def branch_table():
            st = _lib.cur_stream(dev)
            Fn._infonce_bwd_raw(xu, sbu, rows_c[:bsz], sa[:bsz], inv_tau, None, None, lse[:bsz], w[:bsz], engine_flag=eng, out=g0[:n_u])
            Fn._infonce_bwd_raw(xi, sbi, rows_c[bsz:], sa[bsz:], inv_tau, None, None, lse[bsz:], w[bsz:], engine_flag=eng, out=g0[n_u:])
            for a_sl, tab, stab, pos, g_tab in ((slice(0, bsz), xu, sbu, user_idx, g0[:n_u]), (slice(bsz, 2 * bsz), xi, sbi, pos_idx, g0[n_u:])):
                _lib.check(L.gcr_infonce_pos_bwd_f32(_lib.dptr(rows_c[a_sl]), _lib.dptr(sa[a_sl]), _lib.dptr(tab), _lib.dptr(stab),
                                                     _lib.dptr(pos), _lib.dptr(neg_w[a_sl]), bsz, tab.shape[0], d, inv_tau,
                                                     None, _lib.dptr(g_tab), st), "gcr_infonce_pos_bwd_f32")
            _lib.check(L.gcr_normalize_bwd_f32(_lib.dptr(x0), _lib.dptr(sb), _lib.dptr(g0), n, d, _lib.dptr(g0), st),
                       "gcr_normalize_bwd_f32")

        neg_w = -w
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                branch_table()
        else:
            branch_table()

        # anchor side: softmax part = w * inv_tau * o (the flash forward kept o), positives added, then F.normalize backward
        ga = o * (w * inv_tau).unsqueeze(1)
        for a_sl, tab, stab, pos in ((slice(0, bsz), xu, sbu, user_idx), (slice(bsz, 2 * bsz), xi, sbi, pos_idx)):
            _lib.check(L.gcr_infonce_pos_bwd_f32(_lib.dptr(rows_c[a_sl]), _lib.dptr(sa[a_sl]), _lib.dptr(tab), _lib.dptr(stab),
                                                 _lib.dptr(pos), _lib.dptr(neg_w[a_sl]), bsz, tab.shape[0], d, inv_tau,
                                                 _lib.dptr(ga[a_sl]), None, stream), "gcr_infonce_pos_bwd_f32")
        _lib.check(L.gcr_normalize_bwd_f32(_lib.dptr(rows_c), _lib.dptr(sa), _lib.dptr(ga), 2 * bsz, d, _lib.dptr(ga), stream),
                   "gcr_normalize_bwd_f32")

        # Zg: d total / d final (BPR rows), the only zero-filled [N, d] buffer of the step
        zg = torch.zeros_like(x0)
        _lib.check(L.gcr_bpr_bwd_f32(_lib.dptr(fu), _lib.dptr(fi), d, _lib.dptr(user_idx), _lib.dptr(pos_idx), _lib.dptr(neg_idx),
                                     bsz, 1, n_u, n_i, _lib.dptr(dldx), _lib.dptr(gs), _lib.dptr(zg[:n_u]), _lib.dptr(zg[n_u:]),
                                     stream), "gcr_bpr_bwd_f32")
        # Horner recurrence on A^T in h' = h / c:  h'_K = Zg + g_K / c,  h'_k = Zg + g_k / c + A^T h'_{k+1},  dx0 = c h'_0 + G0
        gt = graph.t
        ga_c = ga * inv_c

        def add_ctx(buf):
            _lib.check(L.gcr_scatter_add_rows_f32(_lib.dptr(ga_c), _lib.dptr(gat), 2 * bsz, d, n, _lib.dptr(buf), stream),
                       "gcr_scatter_add_rows_f32")

        h = zg
        # h_K is zero outside the batch rows (<= 3 B, + 2 B context rows when the context layer is the last): the first
        # launch skips every non-zero whose column is clear in this bitmap before its gather
        active = [user_idx, pos_idx + n_u, neg_idx + n_u]
        if ci == K:
            h = zg.clone()
            add_ctx(h)
            active.append(gat)
        bits = Fn.active_rows_bitmap(torch.cat(active), n)
        for k in range(K - 1, 0, -1):
            out = torch.empty_like(x0)
            Fn.spmm_into(gt, h, acc_in=zg, acc_out=out, col_active_bits=bits)
            bits = None                            # the product of a sparse input is dense enough already
            if k == ci:
                add_ctx(out)                       # the context layer's sparse gradient joins h_k after the launch
            h = out
        # ---- the e_step is issued and joined HERE: it depends on `final` only, so it runs beside the structure contrast's
        # backward and the recurrence above; everything that needs the centroids is the 2 B-row prototype contrast below ----
        if ev_final is not None:
            o_.e_step(fu, fi, assign_all=False, join=False, after=ev_final)
            o_.e_step_join()
        elif self.e_step_every_batch and self.e_step_issue == "early":
            o_.e_step_join()

        # ---- prototype contrast (ncl.py:369-375): InfoNCE(e0[idx], centroid of idx's cluster) * batch_size ----
        rows_0 = torch.empty(2 * bsz, d, dtype=torch.float32, device=dev)
        _lib.check(L.gcr_gather_rows_f32(_lib.dptr(x0), _lib.dptr(gat), 2 * bsz, d, n, _lib.dptr(rows_0), stream),
                   "gcr_gather_rows_f32")
        # kmeans.index.search(x, 1) for the batch's rows only (ncl.py:371-372 read user_2cluster[user_idx] / item_2cluster[item_idx])
        from .kmeans import assign_to_centroids
        rows_f = torch.empty(2 * bsz, d, dtype=torch.float32, device=dev)
        _lib.check(L.gcr_gather_rows_f32(_lib.dptr(final), _lib.dptr(gat), 2 * bsz, d, n, _lib.dptr(rows_f), stream),
                   "gcr_gather_rows_f32")
        cents = torch.cat([o_.user_centroids[assign_to_centroids(rows_f[:bsz], o_.user_centroids)],
                           o_.item_centroids[assign_to_centroids(rows_f[bsz:], o_.item_centroids)]])
        s0, sc = Fn.row_inv_norm(rows_0), Fn.row_inv_norm(cents)
        lse_p = torch.empty(2 * bsz, dtype=torch.float32, device=dev)
        pl_p = torch.empty(2 * bsz, dtype=torch.float32, device=dev)
        o_p = torch.empty(2 * bsz, d, dtype=torch.float32, device=dev)
        for lo in (0, bsz):
            sl = slice(lo, lo + bsz)
            self._contrast_fwd(rows_0[sl], s0[sl], cents[sl], sc[sl], None, inv_tau, eng, lse_p[sl], o_p[sl], pl_p[sl])
        wp = o_.proto_reg * o_.batch_size / bsz                                    # mean over the batch, * batch_size
        proto_loss = wp * (lse_p - pl_p).sum()
        total = rec_loss + l2 / o_.batch_size + ssl_loss + proto_loss
        # its backward: only the anchors get a gradient (the centroids are constants of the e_step)
        gp = o_p * (wp * inv_tau)
        neg_wp = torch.full((2 * bsz,), -wp, dtype=torch.float32, device=dev)
        for lo in (0, bsz):
            sl = slice(lo, lo + bsz)
            _lib.check(L.gcr_infonce_pos_bwd_f32(_lib.dptr(rows_0[sl]), _lib.dptr(s0[sl]), _lib.dptr(cents[sl]), _lib.dptr(sc[sl]),
                                                 None, _lib.dptr(neg_wp[sl]), bsz, bsz, d, inv_tau, _lib.dptr(gp[sl]), None,
                                                 stream), "gcr_infonce_pos_bwd_f32")
        _lib.check(L.gcr_normalize_bwd_f32(_lib.dptr(rows_0), _lib.dptr(s0), _lib.dptr(gp), 2 * bsz, d, _lib.dptr(gp), stream),
                   "gcr_normalize_bwd_f32")
        if side is not None:
            main.wait_stream(side)                 # G0 complete
        _lib.check(L.gcr_scatter_add_rows_f32(_lib.dptr(gp), _lib.dptr(gat), 2 * bsz, d, n, _lib.dptr(g0), stream),
                   "gcr_scatter_add_rows_f32")
        dx0 = torch.empty_like(x0)
        Fn.spmm_into(gt, h, acc_in=zg, acc_in2=g0, acc_in2_scale=inv_c, acc_out=dx0, acc_scale=c, col_active_bits=bits)

        # ---- Adam on the two parameter views of the stacked table (ncl.py:327-329) ----
        pu, pi = enc.embedding_dict["user_emb"], enc.embedding_dict["item_emb"]
        pu.grad, pi.grad = dx0[:n_u], dx0[n_u:]
        self.opt.step()
        return rec_loss, ssl_loss, proto_loss, total

    # ------------------------------------------------------------------------------------------------------------
    def capture(self, batch_size, warmup=2):
        """Arms hipGraph replay (torch.cuda.CUDAGraph) of the step over static index buffers of `batch_size` entries:
        `replay(user_idx, pos_idx, neg_idx)` copies a batch in; its first `warmup` calls run eagerly on a side stream
        (allocator warm-up, cached k-means index vectors, Adam state), the next one captures, every call trains on
        exactly one batch.  The optimizer must be capturable (device step count)."""
        dev = self.enc.table.device
        if not all(g.get("capturable") for g in self.opt.param_groups):
            raise ValueError("capture() needs FusedAdam(capturable=True)")
        if int(warmup) < 1:
            # the first step allocates optimiser state, cached k-means index vectors and the allocator's pools: captured,
            # those allocations would become graph nodes that every replay repeats
            raise ValueError("capture() needs at least one eager warm-up step")
        z = torch.zeros(int(batch_size), dtype=torch.int64, device=dev)
        self._static = [z.clone(), z.clone(), z.clone()]
        self._cuda_graph, self._warm, self._warmup, self._out = None, 0, int(warmup), None
        return self

    def capturable_for(self, batch_size):
        """A graph replays ONE batch size (the epoch's ragged last batch runs eagerly)."""
        return self._static is not None and self._static[0].numel() == int(batch_size)

    def replay(self, user_idx, pos_idx, neg_idx):
        st = self._static
        if st is None:
            raise RuntimeError("call capture(batch_size) first")
        if self._cuda_graph is not None and not self.enc._aliased():
            raise RuntimeError("the encoder's parameters were re-bound after this step was captured: call capture() again")
        dev = st[0].device
        for dst, src in zip(st, (user_idx, pos_idx, neg_idx)):
            dst.copy_(torch.as_tensor(src, device=dev), non_blocking=True)
        if self._cuda_graph is None:
            if self._warm < self._warmup:
                s = torch.cuda.Stream(device=dev)
                s.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(s):
                    out = self(*st)
                torch.cuda.current_stream(dev).wait_stream(s)
                self._warm += 1
                return out
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self(*st)
            self._cuda_graph = g               # capturing does not execute: the replay below trains on this batch
        self._cuda_graph.replay()
        return self._out
