import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recommendation_amd.ncl import NCLModel
from recommendation_amd.sampler import next_batch_pairwise
rng = np.random.default_rng(0)
n_u, n_i, groups = 300, 120, 6
pairs = set()
while len(pairs) < 7000:
    u = int(rng.integers(0, n_u)); g = u % groups
    i = int(rng.integers(0, n_i // groups)) * groups + g if rng.random() < 0.9 else int(rng.integers(0, n_i))
    pairs.add((u, i))
pairs = sorted(pairs); rng.shuffle(pairs)
train = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[:6000]]
test = [[f"u{u}", f"i{i}", 1.0] for u, i in pairs[6000:]]
conf = {"model": {"name": "NCL", "type": "graph"}, "embedding.size": 64, "batch.size": 512, "learning.rate": 0.01,
        "reg.lambda": 1e-4, "max.epoch": 8, "item.ranking.topN": [10, 20],
        "NCL": {"n_layers": 3, "tau": 0.1, "ssl_reg": 1e-4, "proto_reg": 1e-4, "alpha": 1.0, "num_clusters": 20, "hyper_layers": 1}}
def run(nsteps):
    torch.manual_seed(0)
    m = NCLModel(conf, train, test, device="cuda", seed=1)
    opt = torch.optim.Adam(m.model.parameters(), lr=0.01, fused=True)
    m.e_step()
    out = []
    for n, batch in enumerate(next_batch_pairwise(m.data, 512, seed=1, epoch=0)):
        if n >= nsteps: break
        rec, ssl, proto, total = m.train_step(batch, opt)
        out.append((float(rec), float(ssl), float(proto), m.user_2cluster.clone()))
    return out, m.model.embedding_dict["user_emb"].detach().clone()
a, ea = run(6); b, eb = run(6)
for k, (x, y) in enumerate(zip(a, b)):
    print(k, "rec %.6f/%.6f ssl %.6f/%.6f proto %.6f/%.6f  assign diff %d" % (x[0], y[0], x[1], y[1], x[2], y[2], int((x[3] != y[3]).sum())))
print("emb max abs diff", float((ea - eb).abs().max()), "emb scale", float(ea.abs().max()))
