import sys, time
sys.path.insert(0, "/root/repo/inverse-flow_amd"); sys.path.insert(0, "/root/repo")
import torch
import invflow_hip as H
from bench import B, C, HH, WW
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn(B, C, HH, WW, device="cuda")
for name, f in (("conv2d_wgrad p=1", lambda: H.conv2d_wgrad(g, x, (C, C, 3, 3), (1, 1))),
                ("conv2d_igrad p=1", lambda: H.conv2d_igrad(g, torch.randn(C, C, 3, 3, device="cuda"), x.shape, (1, 1)))):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); print(name, "ms", (time.perf_counter() - t0) / 5 * 1e3)
