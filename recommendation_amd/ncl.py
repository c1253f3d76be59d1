"""NCLModel with the reference's interface (ncl.py:282-394), every per-step stage on the HIP path:

    NCLModel(conf, train_set, test_set).train()  ->  {'Hit Ratio': .., 'Precision': .., 'Recall': .., 'NDCG': ..}

Stage by stage (reference line -> here):
    Interaction / LGCNEncoder            ncl.py:46-88, 397-422   encoders.py (gcr_coo_to_csr, gcr_spmm_csr_f32)
    next_batch_pairwise                  ncl.py:91-114            sampler.py  (gcr_neg_sample)
    e_step / run_kmeans (faiss)          ncl.py:340-356           kmeans.py   (gcr_kmeans_*; parity with faiss unpinned)
    bpr_loss, l2_reg_loss                ncl.py:116-123           losses.py   (gcr_bpr_*)
    ssl_layer_loss, ProtoNCE_loss        ncl.py:358-375           losses.py   (gcr_infonce_*)
    test / evaluate                      ncl.py:253-264, 377-384  evaluate.py (gcr_score_rows_f32, gcr_topk_masked_f32)
The tuner, logging and result files around it are out of scope.
"""
from __future__ import annotations

import torch

from . import functional as Fn
from . import losses as Ls
from .encoders import Interaction, LGCNEncoder
from .evaluate import ranking_evaluation, test as rank_test
from .kmeans import FAISS_MIN_POINTS_PER_CENTROID, FAISS_NITER, FAISS_SEED, assign_to_centroids, run_kmeans
from .ncl_step import FusedNCLStep
from .optim import FusedAdam
from .sampler import next_batch_pairwise


# attribute <- key of conf["NCL"] (ncl.py:286-292) and attribute <- (key of conf, default) (ncl.py:293-298): the schema a
# drop-in has to keep, as data
_NCL_ARGS = {"n_layers": "n_layers", "ssl_temp": "tau", "ssl_reg": "ssl_reg", "proto_reg": "proto_reg",
             "hyper_layers": "hyper_layers", "alpha": "alpha", "k": "num_clusters"}
_CONF_DEFAULTS = {"batch_size": ("batch.size", 2048), "emb_size": ("embedding.size", 64), "lRate": ("learning.rate", 0.001),
                  "reg": ("reg.lambda", 0.0001), "max_epoch": ("max.epoch", 1),
                  "ranking": ("item.ranking.topN", [10, 20, 30, 50])}


class NCLModel:
    def __init__(self, conf, train_set, test_set, device=None, seed=0, kmeans_niter=FAISS_NITER, kmeans_seed=FAISS_SEED,
                 fused_step=True, graph_capture=False, reorder=None, reorder_guard=True):
        """kmeans_niter / kmeans_seed: the two parameters of `faiss.Kmeans(d, k, gpu=False)` that ncl.py:352 leaves at
        faiss' defaults (ClusteringParameters: niter = 25, seed = 1234; see kmeans.py for provenance — faiss is absent,
        parity with it unpinned), explicit here.  fused_step: run the loop body as the hand-derived launch sequence of
        ncl_step.FusedNCLStep when the configuration allows it (same arithmetic as the autograd path);
        graph_capture: replay that sequence from a hipGraph.  reorder="spectral": locality renumbering of users / items
        (reorder.py; ids stay consistent everywhere because every id comes from `self.data`; with reorder_guard it skips
        itself on graphs it cannot speed up, `self.data.reorder_decision`)."""
        self.config, self.seed = conf, seed
        self.kmeans_niter, self.kmeans_seed = int(kmeans_niter), int(kmeans_seed)
        self.fused_step, self.graph_capture = bool(fused_step), bool(graph_capture)
        self._fused = None
        self.model_name = conf.get("model", {}).get("name", "NCL")
        for attr, (key, default) in _CONF_DEFAULTS.items():
            setattr(self, attr, conf.get(key, default))
        for attr, key in _NCL_ARGS.items():
            setattr(self, attr, conf["NCL"][key])
        self.topN = list(map(int, self.ranking))
        self.max_N = max(self.topN)
        # `train_set` may be a prepared data object (user_num, item_num, norm_adj, device, ...) instead of the triple list:
        # NCLModel.from_graph builds one around a device-resident operator (graphs too large for Python id maps)
        self.data = train_set if hasattr(train_set, "norm_adj") else Interaction(conf, train_set, test_set, device=device,
                                                                                  reorder=reorder, reorder_guard=reorder_guard)
        self.model = LGCNEncoder(self.data, self.emb_size, self.n_layers)
        self.user_centroids = self.item_centroids = None
        self._user_2cluster = self._item_2cluster = self._e_inputs = None
        self.concurrent_e_step, self._e_streams = True, None
        self.bestPerformance = []

    @classmethod
    def from_graph(cls, conf, norm_adj, user_num, item_num, **kw):
        """NCLModel over an operator that already lives on the device (bench.py's 1M x 100K graph): everything the
        training step reads from `Interaction` — sizes, the operator, the device — without the Python id maps."""
        import types
        data = types.SimpleNamespace(user_num=int(user_num), item_num=int(item_num), norm_adj=norm_adj, device=norm_adj.device,
                                     test_set={}, training_set_u={}, user={}, item={})
        return cls(conf, data, None, device=norm_adj.device, **kw)

    # ncl.py:340-356
    def e_step(self, user_emb=None, item_emb=None, assign_all=True, join=True, after=None):
        """`user_emb` / `item_emb`: the encoder outputs when the caller has just computed them with the
        current parameters (the training step has: the reference runs the same forward a second time,
        ncl.py:341, with identical results).
        assign_all=False: train the centroids only; `user_2cluster` / `item_2cluster` — `kmeans.index.search(x, 1)` over
        ALL rows, of which ncl.py:371-372 reads the batch's 2 x B entries — are then computed when (and if) somebody
        reads the attribute, from the same embeddings and centroids (the hand-derived step assigns just its batch rows).
        join=False (concurrent form only): the caller's stream does NOT wait for the two k-means streams; the caller
        goes on issuing independent work and calls `e_step_join()` in front of the first use of the centroids.
        after (concurrent form only): a torch.cuda.Event behind which the embeddings are complete; the two streams wait for
        IT instead of for everything the caller's stream has been given so far — the caller may then issue the e_step late
        in program order (behind independent big launches) without making it wait for them."""
        with torch.no_grad():
            if user_emb is None or item_emb is None:
                user_emb, item_emb, _ = self.model()
            user_emb, item_emb = user_emb.detach(), item_emb.detach()
            # ncl.py:350-351 clamps `self.k = min(self.k, max(2, n // 39))` and KEEPS it: the item k-means
            # inherits the users' clamp and, from the second e_step on, the users inherit the items'
            # ncl.py:352 builds a fresh faiss.Kmeans for each table: both run with the same seed
            k_users = min(int(self.k), max(2, user_emb.shape[0] // FAISS_MIN_POINTS_PER_CENTROID))
            k_items = min(k_users, max(2, item_emb.shape[0] // FAISS_MIN_POINTS_PER_CENTROID))
            kw = dict(niter=self.kmeans_niter, seed=self.kmeans_seed, assign_points=assign_all)
            if user_emb.is_cuda and self.concurrent_e_step:
                # the two k-means are independent and each is a chain of short launches that leaves most of the chip idle
                # (~300 workgroups per search): run them side by side on two streams, join on the caller's
                dev = user_emb.device
                cur = torch.cuda.current_stream(dev)
                if self._e_streams is None:
                    pr = int(getattr(self, "e_stream_priority", 0))
                    self._e_streams = (torch.cuda.Stream(device=dev, priority=pr), torch.cuda.Stream(device=dev, priority=pr))
                res = []
                for s, (x, k) in zip(self._e_streams, ((user_emb, k_users), (item_emb, k_items))):
                    if after is not None:
                        s.wait_event(after)
                    else:
                        s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        res.append(run_kmeans(x.contiguous(), k, **kw))
                (self.user_centroids, self._user_2cluster), (self.item_centroids, self._item_2cluster) = res
                self._e_unjoined = True
                if join:
                    self.e_step_join()
            else:
                self.user_centroids, self._user_2cluster = run_kmeans(user_emb.contiguous(), k_users, **kw)
                self.item_centroids, self._item_2cluster = run_kmeans(item_emb.contiguous(), k_items, **kw)
            self.k = int(self.item_centroids.shape[0])
            self._e_inputs = None if assign_all else (user_emb, item_emb)

    def e_step_join(self):
        """The caller's current stream waits for the e_step's two k-means streams (no-op when already joined)."""
        if getattr(self, "_e_unjoined", False):
            cur = torch.cuda.current_stream(self.user_centroids.device)
            for s in self._e_streams:
                cur.wait_stream(s)
            for t in (self.user_centroids, self._user_2cluster, self.item_centroids, self._item_2cluster):
                if t is not None:
                    t.record_stream(cur)
            self._e_unjoined = False

    @property
    def user_2cluster(self):
        if self._user_2cluster is None and self._e_inputs is not None:
            self._user_2cluster = assign_to_centroids(self._e_inputs[0], self.user_centroids)
        return self._user_2cluster

    @user_2cluster.setter
    def user_2cluster(self, value):
        self._user_2cluster = value

    @property
    def item_2cluster(self):
        if self._item_2cluster is None and self._e_inputs is not None:
            self._item_2cluster = assign_to_centroids(self._e_inputs[1], self.item_centroids)
        return self._item_2cluster

    @item_2cluster.setter
    def item_2cluster(self, value):
        self._item_2cluster = value

    # ncl.py:358-367
    def ssl_layer_loss(self, context, initial, user, item):
        return Ls.ssl_layer_loss(context, initial, user, item, self.data.user_num, self.ssl_temp, self.ssl_reg, self.alpha)

    # ncl.py:369-375
    def ProtoNCE_loss(self, initial_emb, user_idx, item_idx):
        return Ls.ProtoNCE_loss(initial_emb, user_idx, item_idx, self.data.user_num, self.user_centroids,
                                self.user_2cluster, self.item_centroids, self.item_2cluster, self.ssl_temp,
                                self.proto_reg, self.batch_size)

    def train_step(self, batch, optimizer, check_negatives=True, fused=None):
        """One iteration of the loop body ncl.py:311-329.  A batch with an unfilled negative slot (-1, the
        sampler's 100-trial bail-out) is skipped like the reference does (ncl.py:113: it is never yielded);
        returns None then (check_negatives=False: the caller's sampler has already dropped such batches —
        `next_batch_pairwise` does — and the step starts without a host read-back).
        fused (default: self.fused_step): the hand-derived launch sequence of ncl_step.FusedNCLStep."""
        user_idx, pos_idx, neg_idx = batch
        if check_negatives and bool((torch.as_tensor(neg_idx) < 0).any()):
            return None
        fused = self.fused_step if fused is None else fused
        if fused and FusedNCLStep.supported(self):
            if self._fused is None or self._fused.opt is not optimizer:
                self._fused = FusedNCLStep(self, optimizer)
                if self.graph_capture:
                    self._fused.capture(len(user_idx))
            dev = self.model.table.device
            idx = [torch.as_tensor(t, device=dev, dtype=torch.int64).contiguous() for t in (user_idx, pos_idx, neg_idx)]
            if self.graph_capture and self._fused.capturable_for(len(user_idx)):
                return self._fused.replay(*idx)
            return self._fused(*idx)
        rec_user_emb, rec_item_emb, emb_list = self.model()
        # ncl.py:314-317 gathers three [B, d] row blocks and feeds bpr_loss / l2_reg_loss; here the gathers, the BPR
        # terms and the three squared norms come out of ONE kernel (gcr_bpr_fwd_f32) and its backward scatters row
        # gradients straight into the tables (no dense zero-filled gradient per gather)
        n_b = len(user_idx)
        sums = Fn.bpr_sums(rec_user_emb, rec_item_emb, user_idx, pos_idx, neg_idx, Fn.BPR_NCL)
        rec_loss = sums[0] / n_b
        # l2_reg_loss(reg, u, p, n) = reg * (|u|_F + |p|_F + |n|_F) / B   (ncl.py:122-123), then / batch_size (ncl.py:326)
        l2 = self.reg * (sums[1].sqrt() + sums[2].sqrt() + sums[3].sqrt()) / n_b
        initial_emb = emb_list[0]
        context_emb = emb_list[-1] if self.hyper_layers * 2 >= len(emb_list) else emb_list[self.hyper_layers * 2]
        ssl_loss = self.ssl_layer_loss(context_emb, initial_emb, user_idx, pos_idx)
        self.e_step(rec_user_emb, rec_item_emb)                             # ncl.py:324 (every batch, Q8)
        proto_loss = self.ProtoNCE_loss(initial_emb, user_idx, pos_idx)
        total = rec_loss + l2 / self.batch_size + ssl_loss + proto_loss
        optimizer.zero_grad()
        total.backward()
        optimizer.step()
        return rec_loss, ssl_loss, proto_loss, total

    def train(self):
        # ncl.py:305 torch.optim.Adam: gcr_adam_step_f32 (step count on the device when the step is replayed from a graph)
        optimizer = FusedAdam(self.model.parameters(), lr=self.lRate, capturable=self.graph_capture)
        self.model.train()
        for epoch in range(self.max_epoch):
            self.e_step()
            batches = next_batch_pairwise(self.data, self.batch_size, seed=self.seed, epoch=epoch)
            for n, batch in enumerate(batches):
                out = self.train_step(batch, optimizer, check_negatives=False)
                if out is not None and n % 100 == 99:
                    names = ("Rec_loss", "ssl_loss", "Proto_loss", "Total_loss")
                    print(f"Batch {n + 1}: " + ", ".join(f"{k}={float(v):.4f}" for k, v in zip(names, out)))
        self.model.eval()
        with torch.no_grad():
            self.user_emb, self.item_emb, _ = self.model()
        return self.evaluate()

    def test(self):
        return rank_test(self.data, self.user_emb.contiguous(), self.item_emb.contiguous(), self.max_N)

    def evaluate(self):
        metrics = ranking_evaluation(self.data.test_set, self.test(), self.topN)
        print("Detailed TopN Evaluation :")
        print("".join(metrics))
        result = {}
        for line in metrics[1:]:          # "Hit Ratio:0.1234\n" ... of the last cut-off, as the reference returns
            name, sep, value = line.partition(":")
            if sep:
                result[name] = float(value)
        return result

    def predict(self, u):
        uid = self.data.get_user_id(u)
        return torch.matmul(self.user_emb[uid], self.item_emb.T).cpu().numpy()
