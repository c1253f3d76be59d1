import os, sys
sys.path.insert(0, "/root/repo")
sys.argv = [sys.argv[0]]
import torch
from torch.profiler import profile, ProfilerActivity
exec(open("/root/repo/profiles/ncl_step_probe.py").read().split("for _ in range(6):")[0])
def step():
    final, layers = Fn.lightgcn_propagate(graph, xp, 3, "mean", return_layers=True)
    ue, ie = Fn.split_rows(final, n_u)
    bs = Fn.bpr_sums(ue, ie, uidx, iidx, jn, Fn.BPR_NCL)
    loss = bs[0] / bsz + 1e-4 * (bs[1].sqrt() + bs[2].sqrt() + bs[3].sqrt()) / bsz / bsz + \
        Ls.ssl_layer_loss(layers[2], layers[0], uidx, iidx, n_u, 0.1, 1e-6, 1.0) + \
        Ls.ProtoNCE_loss(layers[0], uidx, iidx, n_u, cent, u2c, cent, i2c, 0.1, 1e-7, bsz)
    opt.zero_grad()
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::add", "aten::add_", "aten::mul", "aten::fill_", "aten::zero_", "aten::cat", "aten::copy_", "aten::zeros", "aten::index", "aten::sum"):
        rows.append((e.device_time_total / 3e3, e.count / 3, e.key, str(e.input_shapes)[:90]))
for r in sorted(rows, reverse=True)[:18]:
    print("%.3f ms/step  %4.1f calls  %-12s %s" % r)
