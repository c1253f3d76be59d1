"""Host-side mirror of univariate/mhcn.py's multi-channel encoder (BASELINE config 5) on the HIP ops.

  MHCNEncoder.propagate         mhcn.py:422-466  five SpMMs per layer over H_s, H_j, H_p [U x U], R^T [I x U],
                                                 R [U x I]: the RAW product feeds the next layer, its row-
                                                 normalised copy is appended and summed (gcr_spmm_csr_dual_f32),
                                                 the five launches on separate HIP streams
  hierarchical_self_supervision mhcn.py:480-506  edge embeddings = one more SpMM per channel
  self_gating / channel_attention mhcn.py:404-420 dense [U, d] x [d, d] products: plain library GEMMs (torch)

The operators are CsrGraph handles (graph.py); `build_hyper_graphs` makes them from the social and
interaction lists (motif adjacency, mhcn.py:340-368, on the device: graph_ops.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as Fn
from .distributed import _hip_dual, _hip_spmm_t
from .encoders import multi_stream_spmm
from .graph import CsrGraph


class MHCNEncoder(nn.Module):
    """Parameters and forward of univariate/mhcn.py's MHCN (:370-478) with the reference's attribute names.
    h_s, h_j, h_p: CsrGraph [U, U] (row-normalised motif adjacencies); r: CsrGraph [U, I] (row-normalised
    interaction matrix, `Graph.normalize_graph_mat`, mhcn.py:401)."""

    def __init__(self, h_s: CsrGraph, h_j: CsrGraph, h_p: CsrGraph, r: CsrGraph, emb_size=64, n_layers=2,
                 ss_rate=0.01, concurrent=True):
        super().__init__()
        self.H_s, self.H_j, self.H_p, self.R = h_s, h_j, h_p, r
        self.user_num, self.item_num = r.n_rows, r.n_cols
        self.emb_size, self.n_layers, self.ss_rate, self.n_channel = emb_size, n_layers, ss_rate, 4
        dev = r.device
        xav = nn.init.xavier_uniform_
        self.user_embeddings = nn.Parameter(xav(torch.empty(self.user_num, emb_size, device=dev)))
        self.item_embeddings = nn.Parameter(xav(torch.empty(self.item_num, emb_size, device=dev)))
        self.gating_weights, self.gating_bias = nn.ParameterDict(), nn.ParameterDict()
        self.sgating_weights, self.sgating_bias = nn.ParameterDict(), nn.ParameterDict()
        for c in range(1, self.n_channel + 1):
            self.gating_weights[str(c)] = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
            self.gating_bias[str(c)] = nn.Parameter(torch.zeros(1, emb_size, device=dev))
            self.sgating_weights[str(c)] = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
            self.sgating_bias[str(c)] = nn.Parameter(torch.zeros(1, emb_size, device=dev))
        self.attention = nn.Parameter(xav(torch.empty(1, emb_size, device=dev)))
        self.attention_mat = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(5)] if (concurrent and dev.type == "cuda") else None

    # -- dense pieces (mhcn.py:404-420) ---------------------------------------------------------
    def self_gating(self, em, channel):
        return Fn.gate(em, Fn.dense_proj(em, self.gating_weights[str(channel)]), self.gating_bias[str(channel)])

    def self_supervised_gating(self, em, channel):
        return Fn.gate(em, Fn.dense_proj(em, self.sgating_weights[str(channel)]), self.sgating_bias[str(channel)])

    def channel_attention(self, *channel_embeddings, extra=None, extra_scale=0.0):
        """mhcn.py:413-420 -> (mixed, score [3, U]).  sum(attention * (e @ attention_mat), 1) = e @ (attention_mat @
        attention^T): one [U, d] x [d] product per channel instead of a [U, d] x [d, d] GEMM, a multiply and a row
        reduction; logits, softmax and the weighted sum are one pass (Fn.channel_mix).  extra / extra_scale: a fourth table
        added to `mixed` in the same pass (the `+ simple_embeddings / 2` of mhcn.py:443,460)."""
        v = (self.attention_mat @ self.attention.t()).squeeze(1)
        if len(channel_embeddings) == 3:
            return Fn.channel_mix(*channel_embeddings, v, extra=extra, extra_scale=extra_scale)
        logits = torch.stack([Fn.rows_dot_vec(e, v) for e in channel_embeddings])
        score = torch.softmax(logits, dim=0)
        mixed = sum(score[k].unsqueeze(1) * e for k, e in enumerate(channel_embeddings))
        return (mixed if extra is None else mixed + extra_scale * extra), score

    # -- the layer loop (mhcn.py:422-466) -------------------------------------------------------
    def _five_spmm(self, c1, c2, c3, mixed, items, sums=None):
        """The five dual products of one layer; sums (per operator: the running sum its normalised product joins): each
        pair comes back as (raw product, sum + normalised product) from the same launch."""
        graphs = [self.H_s, self.H_j, self.H_p, self.R.t, self.R]
        xs = [c1, c2, c3, mixed, items]
        if self._streams is not None:
            return multi_stream_spmm(graphs, xs, self._streams, l2norm="dual", acc=sums)
        pairs = [Fn.spmm_l2norm_dual(g, x) for g, x in zip(graphs, xs)]
        return pairs if sums is None else [(raw, a + n) for (raw, n), a in zip(pairs, sums)]

    def propagate(self):
        """(final_user_embeddings [U, d], final_item_embeddings [I, d])."""
        c1, c2, c3 = (self.self_gating(self.user_embeddings, k) for k in (1, 2, 3))
        simple = self.self_gating(self.user_embeddings, 4)
        items = self.item_embeddings
        sums = [c1, c2, c3, simple, items]          # running sums of the layer lists (layer 0 = the inputs)
        for _ in range(self.n_layers):
            mixed, _ = self.channel_attention(c1, c2, c3, extra=simple, extra_scale=0.5)
            # operator order H_s, H_j, H_p, R^T (-> items), R (-> simple): their running sums are sums[0..2], sums[4], sums[3]
            (c1, s0), (c2, s1), (c3, s2), (new_items, s4), (simple, s3) = \
                self._five_spmm(c1, c2, c3, mixed, items, [sums[0], sums[1], sums[2], sums[4], sums[3]])
            sums = [s0, s1, s2, s3, s4]
            items = new_items
        final_user, _ = self.channel_attention(sums[0], sums[1], sums[2], extra=sums[3], extra_scale=0.5)
        return final_user, sums[4]

    def hierarchical_self_supervision(self, em, adj: CsrGraph, perms=None):
        """mhcn.py:480-506.  perms: the three row permutations the reference draws with torch.randperm
        (row_shuffle, row_column_shuffle x 2); None draws them on the device."""
        n = em.shape[0]
        if perms is None:
            perms = [torch.randperm(n, device=em.device) for _ in range(3)]
        edge = Fn.spmm(adj, em)
        pos = (em * edge).sum(1)
        neg1 = (em[perms[0]] * edge).sum(1)
        neg2 = (edge[perms[1]] * em).sum(1)
        local = (-torch.log(torch.sigmoid(pos - neg1)) - torch.log(torch.sigmoid(neg1 - neg2))).sum()
        graph = edge.mean(0, keepdim=True)
        pos = (edge * graph).sum(1)
        neg1 = (edge[perms[2]] * graph).sum(1)
        return (-torch.log(torch.sigmoid(pos - neg1))).sum() + local

    def forward(self, u_idx, v_idx, neg_idx, perms=None):
        """Same 6-tuple as mhcn.py:422-478: batch user / positive / negative rows, ss_loss, final embeddings.
        perms: optional 9 row permutations (3 per channel) replaying the reference's randperm draws."""
        final_user, final_item = self.propagate()
        ss = 0
        for c, adj in enumerate((self.H_s, self.H_j, self.H_p)):
            p = None if perms is None else perms[3 * c:3 * c + 3]
            ss = ss + self.hierarchical_self_supervision(self.self_supervised_gating(final_user, c + 1), adj, p)
        return final_user[u_idx], final_item[v_idx], final_item[neg_idx], self.ss_rate * ss, final_user, final_item


class HipOps:
    """The SpMM primitives of the sharded encoder on the HIP path (the product default).  Tests that exercise
    only the collectives' choreography on CPU tensors (gloo) inject stand-ins with the same four methods."""

    @staticmethod
    def spmm(graph, x):                       # autograd-aware A x
        return Fn.spmm(graph, x)

    @staticmethod
    def dual(graph, x):                       # autograd-aware (A x, normalize(A x))
        return Fn.spmm_l2norm_dual(graph, x)

    channel_dual = staticmethod(_hip_dual)            # raw launches (no autograd) used inside the channel layer
    channel_spmm_t = staticmethod(_hip_spmm_t)


class ShardedMHCNEncoder(MHCNEncoder):
    """MHCN's layer loop (mhcn.py:422-466) over `world` GPUs, users row-sharded (BASELINE config 5):
      * every rank owns `users_per_rank` user rows: their embeddings, the row blocks H_c[users_g, :] of the three
        channel operators (distributed.ShardedChannels) and R_g = R[users_g, :];
      * channels: one all-gather of each channel's own [U, d] operand per layer, overlapped with the previous
        channel's SpMM on its own stream (distributed.sharded_channel_layer);
      * items: R_g^T mixed_g is a partial [I, d] sum -> all-reduce -> replicated item rows (normalised after the
        reduction); the simple channel R_g items is local.
    Replicated parameters (gating / attention weights, item embeddings) get partial gradients per rank:
    call `allreduce_grads()` after backward."""

    def __init__(self, channels, r_local: CsrGraph, emb_size=64, n_layers=2, ss_rate=0.01, ops=HipOps):
        nn.Module.__init__(self)
        self.channels, self.R, self.ops = channels, r_local, ops
        self.H_s, self.H_j, self.H_p = channels.blocks
        self.user_num, self.item_num = r_local.n_rows, r_local.n_cols      # LOCAL user rows
        self.emb_size, self.n_layers, self.ss_rate, self.n_channel = emb_size, n_layers, ss_rate, 4
        dev = r_local.device
        xav = nn.init.xavier_uniform_
        self.user_embeddings = nn.Parameter(xav(torch.empty(self.user_num, emb_size, device=dev)))
        self.item_embeddings = nn.Parameter(xav(torch.empty(self.item_num, emb_size, device=dev)))
        self.gating_weights, self.gating_bias = nn.ParameterDict(), nn.ParameterDict()
        self.sgating_weights, self.sgating_bias = nn.ParameterDict(), nn.ParameterDict()
        for c in range(1, self.n_channel + 1):
            self.gating_weights[str(c)] = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
            self.gating_bias[str(c)] = nn.Parameter(torch.zeros(1, emb_size, device=dev))
            self.sgating_weights[str(c)] = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
            self.sgating_bias[str(c)] = nn.Parameter(torch.zeros(1, emb_size, device=dev))
        self.attention = nn.Parameter(xav(torch.empty(1, emb_size, device=dev)))
        self.attention_mat = nn.Parameter(xav(torch.empty(emb_size, emb_size, device=dev)))
        self._streams = None

    def propagate(self):
        from . import distributed as gd
        ops, ch = self.ops, self.channels
        c1, c2, c3 = (self.self_gating(self.user_embeddings, k) for k in (1, 2, 3))
        simple = self.self_gating(self.user_embeddings, 4)
        items = self.item_embeddings
        sums = [c1, c2, c3, simple, items]
        for _ in range(self.n_layers):
            mixed, _ = self.channel_attention(c1, c2, c3, extra=simple, extra_scale=0.5)
            hip = ops is HipOps          # the HIP launches fold the layer-list sums in (CPU stand-ins: separate adds)
            pairs = gd.sharded_channel_layer(ch, [c1, c2, c3], ops.channel_dual, ops.channel_spmm_t,
                                             accs=sums[:3] if hip else None)
            new_items = gd.all_reduce_sum(ops.spmm(self.R.t, mixed), ch.group)      # R^T mixed: sum over the ranks' users
            n_i = torch.nn.functional.normalize(new_items, p=2, dim=1)
            if hip:
                (c1, s0), (c2, s1), (c3, s2) = pairs
                simple, s3 = Fn.spmm_l2norm_dual_acc(self.R, items, sums[3])
            else:
                (c1, n1), (c2, n2), (c3, n3) = pairs
                simple, n_s = ops.dual(self.R, items)
                s0, s1, s2, s3 = sums[0] + n1, sums[1] + n2, sums[2] + n3, sums[3] + n_s
            sums = [s0, s1, s2, s3, sums[4] + n_i]
            items = new_items
        final_user, _ = self.channel_attention(sums[0], sums[1], sums[2], extra=sums[3], extra_scale=0.5)
        return final_user, sums[4]

    def replicated_parameters(self):
        return [p for n, p in self.named_parameters() if n != "user_embeddings"]

    def allreduce_grads(self):
        from . import distributed as gd
        gd.allreduce_replicated_grads(self.replicated_parameters(), self.channels.group)

    def forward(self, *a, **kw):
        raise NotImplementedError("the sharded encoder provides propagate(); the per-batch losses are the caller's")
