import sys
sys.path.insert(0, "/root/repo/inverse-flow_amd"); sys.path.insert(0, "/root/repo")
import torch
import invflow_hip as H
torch.manual_seed(0)
for (B, C, HH, WW, K, p) in [(2, 64, 32, 32, 3, 1), (1, 64, 8, 32, 3, 1), (2, 32, 16, 16, 3, 1), (2, 64, 16, 16, 2, 0), (2, 64, 32, 32, 3, 2), (2, 64, 32, 32, 3, 0)]:
    x = torch.randn(B, C, HH, WW, device="cuda"); w = torch.randn(C, C, K, K, device="cuda") * 0.05
    b = torch.randn(C, device="cuda")
    try:
        y = H.conv2d(x, w, b, (p, p))
    except Exception as e:
        print((B, C, HH, WW, K, p), "err", e); continue
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), 1, p).float()
    if y.shape != ref.shape:
        print((B, C, HH, WW, K, p), "shape", y.shape, ref.shape); continue
    d = (y.cpu() - ref)
    print((B, C, HH, WW, K, p), "rel", float(d.norm() / ref.norm()), "max", float(d.abs().max()))
    if float(d.norm() / ref.norm()) > 1e-4:
        bad = (d.abs() > 1e-3).nonzero()
        print(" bad count", len(bad), "first", bad[:5].tolist(), "last", bad[-3:].tolist())
