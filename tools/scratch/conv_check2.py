import sys
sys.path.insert(0, "/root/repo/inverse-flow_amd"); sys.path.insert(0, "/root/repo")
import torch
import invflow_hip as H
torch.manual_seed(0)
B, C, HH, WW, K, p = 1, 64, 32, 32, 3, 1
x = torch.randn(B, C, HH, WW, device="cuda"); w = torch.randn(C, C, K, K, device="cuda") * 0.05
y = H.conv2d(x, w, None, (p, p))
ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, 1, p).float()
d = (y.cpu() - ref).abs()
for co in (0, 17, 63):
    print("co", co)
    for h in range(0, 32):
        print("".join("X" if d[0, co, h, ww] > 1e-3 else "." for ww in range(32)))
# which taps are wrong? use a delta weight
for kh in range(3):
    for kw in range(3):
        w2 = torch.zeros_like(w); 
        for c in range(C): w2[c, c, kh, kw] = 1.0
        y2 = H.conv2d(x, w2, None, (p, p)); r2 = torch.nn.functional.conv2d(x.cpu(), w2.cpu(), None, 1, p)
        dd = (y2.cpu() - r2).abs()
        print("tap", kh, kw, "bad", int((dd > 1e-3).sum()), "bad cols", sorted(set((dd[0, 0] > 1e-3).nonzero()[:, 1].tolist()))[:40])
