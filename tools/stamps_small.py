"""tools/stamps.py for a small-image shape of the whole-image MFMA scan (k_scan_mfma): per-section cycles of its step.
    HIPCC_EXTRA=-DIFL_STAMPS python inverse-flow_amd/build.py --force && python tools/stamps_small.py B,C,H,W,K"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(176, dtype=torch.int64, device="cuda")
os.environ["IFL_STAMPS"] = str(buf.data_ptr())
import invflow_hip as H
B, C, HH, WW, K = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "32,12,16,16,3").split(","))
torch.manual_seed(0)
w = torch.zeros(C, C, K, K); w[:, :, -1, -1] = torch.eye(C)
w = (w + 0.01 * torch.randn(C, C, K, K)).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); z = torch.empty_like(x)
for _ in range(3):
    H.inverse(x, w, out=z)
torch.cuda.synchronize()
full = buf.cpu()[:80]
print("span", int(full[79] - full[78]), "memtime ticks (100 MHz: x10 ns)")
t = full[:64].view(8, 8)
names = ["dma", "wait+bar", "reads+lead", "crit", "epilogue", "trail", "stores", "loop"]
for wv in range(8):
    r = t[wv].tolist()
    if sum(r):
        print("wave", wv, {names[k]: r[k] for k in range(8)}, "total", sum(r))
for k, name in enumerate(["no tile", "tile 0", "tile 1", "both tiles"]):
    c = int(full[68 + k]); tot = int(full[64 + k])
    if c:
        print("steps with", name, ":", c, "steps,", tot // c, "cycles each")
