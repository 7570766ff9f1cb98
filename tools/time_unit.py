"""Inverse-flow block (TL -> TR -> BL -> BR) at the north-star layer shape: one library call per direction
(ifl_unit_inverse_f32 / ifl_unit_backward_f32) against four layer calls (development aid, GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
from bench import B, C, HH, WW
from inf.layers.inv_flow import Inv_FlowUnit

torch.manual_seed(0)
orders = ("TL", "TR", "BL", "BR")
unit = Inv_FlowUnit(C, C, (3, 3)).cuda()  # the reference's own init, stored per order (inv_conv.py:149-186)
ws = [l.weight_fwd.detach().contiguous() for l in unit._chain()]
x = torch.randn(B, C, HH, WW, device="cuda"); g = torch.randn_like(x)

def fused():
    car = [H.new_carry(w) for w in ws]
    zs = H.unit_inverse(x, ws, 0, car)
    return H.unit_backward(g, zs, ws, 0, car)

def layerwise():
    car = [H.new_carry(w) for w in ws]
    h, zs = x, []
    for w, o, c in zip(ws, orders, car):
        h = H.inverse(h, w, o, 0, carry=c); zs.append(h)
    gg, dws = g, []
    for w, o, c, z in reversed(list(zip(ws, orders, car, zs))):
        gg, dw, _ = H.backward(gg, z, w, o, 0, carry=c); dws.append(dw)
    return gg, dws[::-1]

for name, f in (("layer by layer", layerwise), ("one call per direction", fused)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    print("%-24s %.3f ms per block step (inverse + backward of 4 layers), %.0f images/s" % (name, ms, B / ms * 1e3))
