import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from recommendation_amd import functional as Fn
from recommendation_amd import _lib
if "--lib" in sys.argv:
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
rng = np.random.default_rng(0)
for m in (33, 64, 65, 129, 257, 300):
    d = 64
    x = (rng.standard_normal((m, d)) * 0.5).astype(np.float32)
    w = rng.standard_normal(m).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    s = Fn.row_inv_norm(xt)
    inv_tau = 4.0
    xn = x.astype(np.float64) / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)
    sc = inv_tau * xn @ xn.T
    for exd in (True, False):
        S = sc.copy()
        if exd:
            np.fill_diagonal(S, -np.inf)
        lse = np.log(np.exp(S).sum(1))
        P = np.exp(S - lse[:, None])
        ref_y = inv_tau * (P * w[:, None]).T @ xn          # streamed-side stats (lse_y, w_y): g_j = sum_i w_i P_ij x_i
        ref_x = inv_tau * (P * w[:, None]) @ xn            # stationary-side stats
        lt, wt = torch.from_numpy(lse.astype(np.float32)).cuda(), torch.from_numpy(w).cuda()
        for name, ef in (("h2", Fn.INFONCE_UNIT_ROWS), ("b3", 0)):
            gy = Fn._infonce_bwd_raw(xt, s, xt, s, inv_tau, None, None, lt, wt, exclude_diagonal=exd, engine_flag=ef).cpu().numpy()
            gx = Fn._infonce_bwd_raw(xt, s, xt, s, inv_tau, lt, wt, None, None, exclude_diagonal=exd, engine_flag=ef).cpu().numpy()
            print(m, "exd" if exd else "   ", name, "y-side err %.2e  x-side err %.2e" % (
                np.abs(gy - ref_y).max() / np.abs(ref_y).max(), np.abs(gx - ref_x).max() / np.abs(ref_x).max()), flush=True)
