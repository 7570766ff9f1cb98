"""Device time of the weight-gradient pair (k_wgrad_mfma + k_wgrad_reduce) at the north-star shape, by stream events:
python tools/time_wgrad.py [--lib libinvflow_hip_<name>.so] ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", sys.argv[2])
from bench import B, C, HH, WW
torch.manual_seed(0)
z = torch.randn(B, C, HH, WW, device="cuda"); dx = torch.randn_like(z)
dw = torch.empty(C, C, 3, 3, device="cuda")
for _ in range(10): H.dw_from(z, dx, (3, 3), out=dw)
res = []
for rep in range(3):  # the library's own event pairs around the tagged launch (the calls themselves are host-bound)
    torch.cuda.synchronize(); H.profile_enable(True)
    for _ in range(100): H.dw_from(z, dx, (3, 3), out=dw)
    torch.cuda.synchronize(); H.profile_enable(False)
    ms, n = H.profile_collect()["wgrad"]
    res.append(ms / n * 1e3)
print("%-28s k_wgrad_mfma: %s us   checksum %.9e" % (os.path.basename(H.LIB_PATH), " ".join("%.2f" % r for r in res), float(dw.double().abs().sum())))
