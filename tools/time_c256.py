import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
B, C, HH, WW, K = 16, 256, 8, 8, 3
w = torch.zeros(C, C, K, K); w[torch.arange(C), torch.arange(C), K - 1, K - 1] = 1.0
w = (w + 0.01 * torch.randn(C, C, K, K)).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
for _ in range(2):
    H.inverse(x, w, "TL", 0, out=z); H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw)
torch.cuda.synchronize()
for name, f in (("inverse", lambda: H.inverse(x, w, "TL", 0, out=z)), ("backward dx only", lambda: H.backward(g, z, w, "TL", 0, dx_out=dx, need_dw=False)),
                ("backward", lambda: H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print(name, "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
