"""Per-section cycle counts of the MFMA weight-gradient kernel (development aid).

    HIPCC_EXTRA=-DIFL_STAMPS python inverse-flow_amd/build.py --force && python tools/wstamps.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(4 * 8 + 2 * 1024, dtype=torch.int64, device="cuda")
os.environ["IFL_WSTAMPS"] = str(buf.data_ptr())
import invflow_hip as H
H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", "libinvflow_hip_stamps.so")
from bench import B, C, HH, WW
z = torch.randn(B, C, HH, WW, device="cuda"); dx = torch.randn_like(z)
for _ in range(3):
    dw = H.dw_from(z, dx, (3, 3))
torch.cuda.synchronize()
t = buf.cpu()[:32].view(4, 8)
names = ["prologue", "wait row", "convert+loads", "shift+mfma", "-", "dump + barrier", "sum + store", "-"]
for wv in range(4):
    r = t[wv].tolist()
    print("wave", wv, {names[k]: r[k] for k in range(7) if names[k] != "-"}, "total", sum(r))
se = buf.cpu()[32:].view(1024, 2)
se = se[se[:, 1] > 0].double() / 100.0  # us
t0 = float(se[:, 0].min())
starts, ends = (se[:, 0] - t0), (se[:, 1] - t0)
print("%d workgroups: entry %.2f .. %.2f us after the first, exit %.2f .. %.2f us (median %.2f); residence %.2f .. %.2f us" % (
    len(se), float(starts.min()), float(starts.max()), float(ends.min()), float(ends.max()), float(ends.median()),
    float((ends - starts).min()), float((ends - starts).max())))
