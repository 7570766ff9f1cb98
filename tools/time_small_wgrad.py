"""inverse + backward of the 32x32x3 models' layer shapes, many times (run under rocprofv3 --kernel-trace --stats:
tools/kstats_any.sh <tag> tools/time_small_wgrad.py) -- the per-kernel durations of the small layers' kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
torch.manual_seed(0)
SHAPES = [(32, 12, 16, 16, 3), (32, 24, 8, 8, 3), (32, 48, 4, 4, 3)]
for (B, C, HH, WW, K) in ([SHAPES[int(sys.argv[1])]] if len(sys.argv) > 1 else SHAPES):
    w = (torch.nn.init.dirac_(torch.empty(C, C, K, K)) + 0.02 * torch.randn(C, C, K, K)).cuda()
    x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
    z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
    for _ in range(200):
        H.inverse(x, w, "TL", 0, out=z)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw)
    torch.cuda.synchronize()
    print("done", (B, C, HH, WW, K), flush=True)
