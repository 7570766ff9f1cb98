import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(8 * 2 * 8, dtype=torch.int64, device="cuda")
os.environ["IFL_STAMPS"] = str(buf.data_ptr())
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
for _ in range(3):
    H.inverse(x, w, out=z)
torch.cuda.synchronize()
t = buf.cpu().view(8, 2, 8)
names = ["dma", "B", "wait+bar", "C", "reads", "D+store", "epilogue", "iters"]
for wv in range(8):
    for a in range(2):
        r = t[wv, a].tolist()
        n = max(r[7], 1)
        print("wave", wv, "act" if a else "idle", {names[k]: round(r[k] / n) for k in range(7)}, "iters", r[7], "total", sum(r[:7]))
