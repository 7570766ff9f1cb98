"""The C = 256 layers of the ImageNet-32 Glow (BASELINE configs[4]: per-GPU batch 16, 8x8, 3x3): inverse + backward,
wall clock per step; run under rocprofv3 --kernel-trace (tools/kstats_any.sh wide tools/prof_wide.py) for the kernel list."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
shapes = [(16, 256, 8, 8, 3), (16, 128, 16, 16, 3), (32, 256, 8, 8, 3)]
args = sys.argv[1:]
if args and args[0] == "--lib":
    H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", args[1])
    args = args[2:]
if args:
    shapes = [tuple(int(v) for v in a.split(",")) for a in args]
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
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    for _ in range(20): H.inverse(x, w, "TL", 0, out=z, carry=carry)
    ev[1].record()
    for _ in range(20): H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
    ev[2].record(); torch.cuda.synchronize()
    print("%s B=%-4d C=%-4d %2dx%-2d K=%d  %.3f ms/step  %.0f img/s   inverse %.1f us  backward %.1f us  voided launches %d" % (
        os.path.basename(H.LIB_PATH), B, C, HH, WW, K, ms, B / ms * 1e3, ev[0].elapsed_time(ev[1]) * 50, ev[1].elapsed_time(ev[2]) * 50,
        H.scan_voided(x.device)))
