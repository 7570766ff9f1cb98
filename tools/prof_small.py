"""One ImageNet-32 Glow level's layer shapes (if_multiGPU_imagenet32.py: batch 100, 2x2 kernels): inverse + backward,
wall clock per step; run under rocprofv3 --kernel-trace for the kernel list."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
shapes = [(100, 12, 16, 16, 2), (100, 24, 8, 8, 2), (100, 48, 4, 4, 2)]
if len(sys.argv) > 1:  # B,C,H,W,K ...
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, C, HH, WW, K) in shapes:
    w = torch.zeros(C, C, K, K); w[:, :, -1, -1] = torch.eye(C)
    w = (w + 0.01 * torch.randn(C, C, K, K)).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
    z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
    carry = H.new_carry(w)
    def step():
        H.inverse(x, w, "TL", 0, out=z, carry=carry)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print("B=%-4d C=%-4d %2dx%-2d K=%d  %.3f ms/step  %.0f img/s" % (B, C, HH, WW, K, ms, B / ms * 1e3))
