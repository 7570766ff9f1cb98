"""bf16 storage against fp32 storage: the layer's step at the north-star shape and the one-pass neighbours (device time by
stream events, 50 calls each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
if len(sys.argv) > 2 and sys.argv[1] == "--lib":
    H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", sys.argv[2])
from bench import B, C, HH, WW

def timed(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

torch.manual_seed(0)
w = torch.zeros(C, C, 3, 3); w[:, :, -1, -1] = torch.eye(C); w = (w + 0.01 * torch.randn(C, C, 3, 3)).cuda()
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(B, C, HH, WW, device="cuda").to(dt); g = torch.randn_like(x)
    z = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w); carry = H.new_carry(w)
    def step():
        H.inverse(x, w, "TL", 0, out=z, carry=carry)
        H.backward(g, z, w, "TL", 0, dx_out=dx, dw_out=dw, carry=carry)
    print("%-9s layer step (%d,%d,%d,%d): %7.1f us" % (str(dt)[6:], B, C, HH, WW, timed(step)))
    n = x.numel()
    t, ls = torch.randn(C, device="cuda"), 0.1 * torch.randn(C, device="cuda")
    h = torch.randn_like(x)
    for name, fn, passes in (("actnorm", lambda: H.actnorm(x, t, ls), 2), ("squeeze", lambda: H.space_to_depth(x), 2),
                             ("coupling", lambda: H.coupling(x, h), 4), ("coupling bwd", lambda: H.coupling_backward(g, None, x, h), 7)):
        us = timed(fn)
        # bytes: reads + writes of the activation-sized tensors (coupling: x, h in, y out; backward: gy, x2, hs in, gx, gh out)
        gb = {"actnorm": 2, "squeeze": 2, "coupling": 3, "coupling bwd": 4.5}[name] * n * x.element_size() / 1e9
        print("%-9s %-13s %7.1f us  %6.0f GB/s" % (str(dt)[6:], name, us, gb / (us * 1e-6)))
