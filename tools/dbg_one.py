import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", sys.argv[1] if len(sys.argv) > 1 else "libinvflow_hip.so")
torch.manual_seed(0)
B, C, HH, WW = 1, 64, 8, 8
w = (0.05 * torch.randn(C, C, 3, 3)).cuda()
x = torch.randn(B, C, HH, WW, device="cuda")
z = H.inverse(x, w)
torch.cuda.synchronize()
print("ran", float(z.abs().max()))
zr = H.inverse(x, w, flags=H.FLAG_NO_MFMA)
print("rel err vs general:", float((z - zr).norm() / zr.norm()))
