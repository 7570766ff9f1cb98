"""Wall-clock of the dense pieces around the scan at the north-star shape (development aid, GPU box):
the layer's reverse x^ = A z (+ log-det) and SelfNormConv's conv2d / backward_input / backward_weight."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW

gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
z = torch.randn(B, C, HH, WW, device="cuda"); xh = torch.empty_like(z)
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn(B, C, HH, WW, device="cuda")
wt = torch.randn(C, C, 3, 3, device="cuda") * 0.05
ops = (("forward + logdet (x^ = A z)", lambda: H.forward(z, w, "TL", 0, out=xh, want_logdet=True)),
       ("conv2d p=1", lambda: H.conv2d(x, wt, None, (1, 1))),
       ("conv2d_igrad p=1", lambda: H.conv2d_igrad(g, wt, x.shape, (1, 1))),
       ("conv2d_wgrad p=1", lambda: H.conv2d_wgrad(g, x, (C, C, 3, 3), (1, 1))))
for name, f in ops:
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize(); print("%-30s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
