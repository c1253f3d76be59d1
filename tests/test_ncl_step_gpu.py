"""The hand-derived NCL iteration (recommendation_amd/ncl_step.py, the loop body ncl.py:311-329 as a fixed launch
sequence) against the autograd path of NCLModel.train_step — which tests/test_infonce_gpu.py, test_bpr_sampler_gpu.py and
test_spmm_gpu.py pin to the reference's golden outputs stage by stage: same four losses, same parameter gradients, same
update; and the hipGraph replay of the sequence against its eager execution."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _toy(seed=0, n_u=400, n_i=150, n_pairs=6000):
    rng = np.random.default_rng(seed)
    pairs = set()
    while len(pairs) < n_pairs:
        u = int(rng.integers(0, n_u))
        i = int(rng.integers(0, n_i // 5)) * 5 + u % 5 if rng.random() < 0.85 else int(rng.integers(0, n_i))
        pairs.add((u, i))
    pairs = sorted(pairs)
    rng.shuffle(pairs)
    return [[f"u{u}", f"i{i}", 1.0] for u, i in pairs]


def _conf(n_layers, hyper_layers, batch=512):
    return {"model": {"name": "NCL", "type": "graph"}, "embedding.size": 64, "batch.size": batch, "learning.rate": 0.005,
            "reg.lambda": 1e-4, "max.epoch": 1, "item.ranking.topN": [10],
            "NCL": {"n_layers": n_layers, "tau": 0.1, "ssl_reg": 1e-3, "proto_reg": 1e-3, "alpha": 0.5, "num_clusters": 8,
                    "hyper_layers": hyper_layers}}


def _twin_models(n_layers, hyper_layers, **kw):
    from recommendation_amd.ncl import NCLModel
    train = _toy()
    a = NCLModel(_conf(n_layers, hyper_layers), train, train[:50], device="cuda", seed=3, **kw)
    b = NCLModel(_conf(n_layers, hyper_layers), train, train[:50], device="cuda", seed=3)
    # the raw 0/1 adjacency (ncl.py:74-85) grows the rows by the degree every layer: start small so that three layers
    # stay in a range where fp32 round-off of the two schedules is comparable
    with torch.no_grad():
        a.model.table.mul_(0.1)
        b.model.table.copy_(a.model.table)
    return a, b


@pytest.mark.parametrize("n_layers,hyper_layers", [(2, 1), (3, 1), (3, 2), (1, 1)])
def test_fused_step_equals_autograd_step(n_layers, hyper_layers):
    from recommendation_amd.ncl_step import FusedNCLStep
    from recommendation_amd.optim import FusedAdam
    from recommendation_amd.sampler import next_batch_pairwise
    fused, auto = _twin_models(n_layers, hyper_layers)
    assert FusedNCLStep.supported(fused)
    of, oa = FusedAdam(fused.model.parameters(), lr=0.005), FusedAdam(auto.model.parameters(), lr=0.005)
    batches = next_batch_pairwise(fused.data, 512, seed=1)
    for step in range(2):
        batch = next(batches)
        if step:
            # Adam's normalised update amplifies round-off where a gradient is tiny, so the two tables drift apart by
            # fractions of lr per step: re-align the state, the second step then checks a state with non-zero moments
            with torch.no_grad():
                fused.model.table.copy_(auto.model.table)
                for key in ("user_emb", "item_emb"):
                    sf, sa = of.state[fused.model.embedding_dict[key]], oa.state[auto.model.embedding_dict[key]]
                    sf["exp_avg"].copy_(sa["exp_avg"])
                    sf["exp_avg_sq"].copy_(sa["exp_avg_sq"])
        lf = fused.train_step(batch, of, check_negatives=False, fused=True)
        la = auto.train_step(batch, oa, check_negatives=False, fused=False)
        for name, x, y in zip(("rec", "ssl", "proto", "total"), lf, la):
            assert float(x.detach()) == pytest.approx(float(y.detach()), rel=2e-5, abs=1e-7), (step, name)
        for key in ("user_emb", "item_emb"):
            gf, ga = fused.model.embedding_dict[key].grad, auto.model.embedding_dict[key].grad
            scale = float(ga.abs().max())
            assert float((gf - ga).abs().max()) <= 2e-5 * scale, (step, key)
        # Adam divides by sqrt(v): elements whose gradient is at round-off level may move differently; the tables must
        # agree wherever the gradient is resolved
        mask = auto.model.embedding_dict["user_emb"].grad.abs() > 1e-3 * float(auto.model.embedding_dict["user_emb"].grad.abs().max())
        du = (fused.model.embedding_dict["user_emb"].detach() - auto.model.embedding_dict["user_emb"].detach()).abs()
        assert float(du[mask].max()) <= 0.005 * 0.02
        # the e_step ran inside both (same seeds, same inputs up to round-off): same partition
        assert float((fused.user_2cluster == auto.user_2cluster).float().mean()) > 0.99


def test_fused_step_unsupported_configuration_falls_back():
    from recommendation_amd.ncl import NCLModel
    from recommendation_amd.ncl_step import FusedNCLStep
    train = _toy()
    conf = _conf(2, 1)
    conf["embedding.size"] = 48                               # not an MFMA width: the autograd path pads, the fused one declines
    m = NCLModel(conf, train, train[:50], device="cuda", seed=0)
    assert not FusedNCLStep.supported(m)


def test_graph_replay_equals_eager_sequence():
    """capture() + replay(): the first calls run eagerly, the next is captured, later ones replay — every call trains on
    exactly one batch, and Adam's bias correction follows the device step count.  After 6 steps the table equals the one
    trained by 6 eager fused steps on the same batches (float atomics: equal to round-off, not bitwise)."""
    from recommendation_amd.optim import FusedAdam
    from recommendation_amd.sampler import next_batch_pairwise
    graphed, eager = _twin_models(2, 1, graph_capture=True)
    og = FusedAdam(graphed.model.parameters(), lr=0.005, capturable=True)
    oe = FusedAdam(eager.model.parameters(), lr=0.005)
    batches = [b for b, _ in zip(next_batch_pairwise(graphed.data, 512, seed=2), range(6))]
    for batch in batches:
        lg = graphed.train_step(batch, og, check_negatives=False)
        le = eager.train_step(batch, oe, check_negatives=False, fused=True)
        assert float(lg[3].detach()) == pytest.approx(float(le[3].detach()), rel=1e-3)
    assert graphed._fused._cuda_graph is not None                       # the graph path really ran
    assert int(og.state[graphed.model.embedding_dict["user_emb"]]["step_dev"]) == 6
    diff = (graphed.model.table - eager.model.table).abs()
    # Adam's normalised update amplifies round-off on elements with tiny gradients, so the comparison that binds is on
    # the elements whose gradient is RESOLVED (second moment within 20x of the largest): there the two tables agree to a
    # hundredth of ONE step of size lr — a frozen bias correction or a dropped gradient branch moves them by ~lr per step
    assert float(diff.mean()) < 0.005 * 0.02
    for name in ("user_emb", "item_emb"):
        pe, pg = eager.model.embedding_dict[name], graphed.model.embedding_dict[name]
        rms = oe.state[pe]["exp_avg_sq"].sqrt()
        resolved = rms > 0.05 * rms.max()
        assert int(resolved.sum()) > 100
        assert float((pg - pe).abs()[resolved].max()) < 0.01 * 0.005
        assert float((og.state[pg]["exp_avg"] - oe.state[pe]["exp_avg"]).abs().max()) <= 2e-5 * float(oe.state[pe]["exp_avg"].abs().max())
    # a checkpoint taken after replays carries the device-side step count
    assert all(st["step"] == 6 for st in og.state_dict()["state"].values())


def test_capturable_adam_matches_host_step_count():
    from recommendation_amd.optim import FusedAdam
    g = torch.Generator(device="cuda").manual_seed(0)
    p1 = torch.nn.Parameter(torch.randn(1000, 64, device="cuda", generator=g))
    p2 = torch.nn.Parameter(p1.detach().clone())
    o1, o2 = FusedAdam([p1], lr=1e-2), FusedAdam([p2], lr=1e-2, capturable=True)
    for _ in range(4):
        grad = torch.randn(1000, 64, device="cuda", generator=g)
        p1.grad, p2.grad = grad.clone(), grad.clone()
        o1.step()
        o2.step()
    torch.testing.assert_close(p1, p2, rtol=1e-6, atol=1e-7)
