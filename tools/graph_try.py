"""Experiment: does capturing inverse + backward in a HIP graph shorten the step? (development aid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
from bench import ref_init_weight, B, C, HH, WW
gen = torch.Generator().manual_seed(0)
w = ref_init_weight(gen).cuda()
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)
z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w)
carry = H.new_carry(w)
def step():
    H.inverse(x, w, "TL", 0, out=z, carry=carry)
    H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 50 * 1e3)
dw_ref = dw.clone(); z_ref = z.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
for _ in range(5): gr.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): gr.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 50 * 1e3)
print("same results:", bool(torch.equal(dw, dw_ref)), bool(torch.equal(z, z_ref)))
