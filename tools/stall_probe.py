"""Enqueue many steps of one layer without any host synchronisation (idle periods of 20-80 ms between two kernels were
seen about once per 100-200 launches in such runs).  Run under
    rocprofv3 --hip-trace --kernel-trace -d <dir> -o stall -- python3 tools/stall_probe.py [B,C,H,W,K] [steps]
and look at the gaps with tools/stall_gaps.py <dir>/stall_results.db"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,256,8,8,3").split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
B, C, HH, WW, K = shape
torch.manual_seed(0)
w = torch.zeros(C, C, K, K); w[:, :, -1, -1] = torch.eye(C)
w = (w + 0.01 * torch.randn(C, C, K, K)).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
carry = H.new_carry(w)
for _ in range(3):
    H.inverse(x, w, "TL", 0, out=z, carry=carry); H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    H.inverse(x, w, "TL", 0, out=z, carry=carry)
    H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("shape %s: %d steps enqueued in %.1f ms (%.1f us per step on the host), drained %.1f ms later; %.1f us per step in all"
      % (shape, steps, (t1 - t0) * 1e3, (t1 - t0) / steps * 1e6, (t2 - t1) * 1e3, (t2 - t0) / steps * 1e6))
