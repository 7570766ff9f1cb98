"""Cycle count of one workgroup of the MFMA conv kernel (development aid; build with HIPCC_EXTRA=-DIFL_STAMPS)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(8, dtype=torch.int64, device="cuda")
os.environ["IFL_CSTAMPS"] = str(buf.data_ptr())
import invflow_hip as H
from bench import B, C, HH, WW
x = torch.randn(B, C, HH, WW, device="cuda"); w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
for _ in range(3):
    H.conv2d(x, w, None, (1, 1))
torch.cuda.synchronize()
t = buf.cpu()
for wv in range(4):
    print("wave", wv, "cycles for the workgroup's bands", int(t[wv]))
print("wave 0 phases, summed over its bands:", dict(zip(["convert (loads landed, split, LDS writes)", "barrier + next loads issued", "multiply + stores", "barrier at band end"], [int(v) for v in t[4:8]])))
