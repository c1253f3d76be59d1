import os, sys, io, contextlib, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recommendation_amd.ncl import NCLModel

def data():
    rng = np.random.default_rng(0)
    n_u, n_i, groups = 300, 120, 6
    pairs = set()
    while len(pairs) < 7000:
        u = int(rng.integers(0, n_u)); g = u % groups
        i = int(rng.integers(0, n_i // groups)) * groups + g if rng.random() < 0.9 else int(rng.integers(0, n_i))
        pairs.add((u, i))
    pairs = sorted(pairs); rng.shuffle(pairs)
    return [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[:6000]], [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[6000:]]

base = {"model": {"name": "NCL", "type": "graph"}, "embedding.size": 64, "batch.size": 512, "learning.rate": 0.01,
        "reg.lambda": 1e-4, "max.epoch": 8, "item.ranking.topN": [10, 20],
        "NCL": {"n_layers": 3, "tau": 0.1, "ssl_reg": 1e-4, "proto_reg": 1e-4, "alpha": 1.0, "num_clusters": 20, "hyper_layers": 1}}
variants = {"base": {}, "lr3e-3_ep12": {"learning.rate": 0.003, "max.epoch": 12}, "L2_lr5e-3": {"learning.rate": 0.005, "NCL.n_layers": 2},
            "L1_lr1e-2": {"NCL.n_layers": 1}, "lr1e-3_ep20": {"learning.rate": 0.001, "max.epoch": 20}}
for name, ch in variants.items():
    recs = []
    for rep in range(4):
        conf = copy.deepcopy(base)
        for k, v in ch.items():
            if k.startswith("NCL."):
                conf["NCL"][k[4:]] = v
            else:
                conf[k] = v
        tr, te = data()
        m = NCLModel(conf, tr, te, device="cuda", seed=1)
        with contextlib.redirect_stdout(io.StringIO()):
            res = m.train()
        recs.append(res["Recall"])
    print(name, ["%.3f" % r for r in recs], flush=True)
