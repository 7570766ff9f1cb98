"""debug aid: is a small duo-scan case deterministic?  python tools/dbg_race.py H C K order"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import invflow_hip as H
Hh, C, K, order = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
torch.manual_seed(3)
B, Ww = 3, 32
w = torch.zeros(C, C, K, K); 
for c in range(C): w[c, c, K - 1, K - 1] = 1.0
w = (w + 0.02 * torch.randn(C, C, K, K)).cuda()
x = torch.randn(B, C, Hh, Ww).cuda()
ref = H.inverse(x, w, order)
nbad = 0
for it in range(30):
    z = H.inverse(x, w, order)
    if not torch.equal(z, ref):
        nbad += 1
        idx = (z != ref).nonzero()
        if nbad <= 3:
            print("it", it, "differs at", idx.shape[0], "elements; first", idx[:6].tolist(), "vals", z[tuple(idx[0])].item(), ref[tuple(idx[0])].item())
print("H=%d C=%d K=%d %s: %d of 30 runs differ from the first" % (Hh, C, K, order, nbad))
