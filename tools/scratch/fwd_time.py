import os, sys, time
sys.path.insert(0, "/root/repo/inverse-flow_amd"); sys.path.insert(0, "/root/repo")
import torch
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
z = torch.randn(B, C, HH, WW, device="cuda"); xh = torch.empty_like(z)
for _ in range(3): H.forward(z, w, "TL", 0, out=xh)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): H.forward(z, w, "TL", 0, out=xh)
torch.cuda.synchronize(); print("forward ms", (time.perf_counter()-t0)/20*1e3)
x = torch.randn(B, C, HH, WW, device="cuda"); wt = torch.randn(C, C, 3, 3, device="cuda")*0.05
for name, f in (("conv2d", lambda: H.conv2d(x, wt, None, (1,1))),):
    try:
        for _ in range(2): f()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); print(name, "ms", (time.perf_counter()-t0)/10*1e3)
    except Exception as e: print(name, "err", e)
