"""Run only the inverse scan (and optionally the other ops) a few times: profiling target."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW

what = sys.argv[1] if len(sys.argv) > 1 else "scan"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
for _ in range(n):
    if what in ("scan", "all"):
        H.inverse(x, w, out=z)
    if what in ("bwd", "all"):
        H.backward(g, z, w, dx_out=dx, dw_out=dw)
    if what == "dw":
        H.dw_from(z, g, (3, 3), out=dw)
torch.cuda.synchronize()
print("done")
