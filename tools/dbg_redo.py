"""debug aid: where does the duo scan's redo path (x scaled by 2^-12) go wrong?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import invflow_hip as H
from oracle import oracle
amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.02
rng = np.random.default_rng(21)
B, C, Hh, Ww, K = 2, 64, 32, 32, 3
w = np.zeros((C, C, K, K)); w[np.arange(C), np.arange(C), 1, 1] = 1.0
w = (w + amp * rng.standard_normal((C, C, K, K))).astype(np.float32)
x = rng.standard_normal((B, C, Hh, Ww)).astype(np.float32)
z_o = oracle.inverse(x.astype(np.float64), w.astype(np.float64), 0, "TL", nthreads=8)
for flags, name in ((0, "split"), (H.FLAG_WHOLE_IMAGE, "solo")):
    z = H.inverse(torch.from_numpy(x).cuda(), torch.from_numpy(w).cuda(), "TL", flags).cpu().numpy().astype(np.float64)
    bad = ~np.isfinite(z) | (np.abs(z - z_o) > 1e-3 * (np.abs(z_o) + 1))
    print(name, "bad count", bad.sum(), "of", bad.size)
    if bad.sum():
        idx = np.argwhere(bad)
        print(" images", np.unique(idx[:, 0], return_counts=True))
        print(" channels", np.unique(idx[:, 1], return_counts=True))
        print(" rows", np.unique(idx[:, 2], return_counts=True))
        print(" cols", np.unique(idx[:, 3], return_counts=True))
        print(" diag parity", np.unique((idx[:, 2] + idx[:, 3]) & 1, return_counts=True))
        print(" first", idx[:10].tolist())
    if bad.sum():
        zb = z[0]; bb = bad[0]
        print(" nan count", np.isnan(z).sum(), "inf", np.isinf(z).sum())
        # granularity: per (row, col), how many channels of each 16-group / 4-group are bad
        g16 = bb.reshape(4, 16, 32, 32).sum(1)      # [wave][h][w]
        print(" 16-groups fully bad", (g16 == 16).sum(), "partially", ((g16 > 0) & (g16 < 16)).sum())
        g4 = bb.reshape(16, 4, 32, 32).sum(1)
        print(" 4-groups fully bad", (g4 == 4).sum(), "partially", ((g4 > 0) & (g4 < 4)).sum())
        for (c, h, w_) in np.argwhere(bb)[:12]:
            print("  c%d h%d w%d  got %r want %r  x %r" % (c, h, w_, z[0, c, h, w_], z_o[0, c, h, w_], x[0, c, h, w_]))
