"""Wall-clock of inverse + backward on the reference's other layer shapes (development aid, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
for (B, C, HH, WW, K) in [(32, 12, 16, 16, 3), (32, 24, 8, 8, 3), (32, 48, 4, 4, 3), (100, 4, 14, 14, 2), (100, 8, 7, 7, 2),
                          (64, 1, 28, 28, 3), (16, 256, 8, 8, 3), (128, 32, 32, 32, 3), (128, 64, 32, 32, 3)]:
    w = torch.nn.init.dirac_(torch.empty(C, C, K, K)) if K == 3 else torch.zeros(C, C, K, K)
    w = (w + 0.02 * torch.randn(C, C, K, K)).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
    z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
    def step():
        H.inverse(x, w, "TL", 0, out=z)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("B=%-4d C=%-4d %2dx%-2d K=%d  %.3f ms/step  %.0f img/s" % (B, C, HH, WW, K, ms, B / ms * 1e3))
