"""The inverse scan at the north-star shape in three memory contexts: the same (x, z) pair every launch; eight pairs in
rotation (working set beyond the 256 MB infinity cache); and the bench step's own sequence.  Device time by stream events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H
if len(sys.argv) > 1:
    H.LIB_PATH = os.path.join(ROOT, "inverse-flow_amd", "lib", sys.argv[1])
from bench import ref_init_weight, B, C, HH, WW
w = ref_init_weight(torch.Generator().manual_seed(0)).cuda()
NP = 8
xs = [torch.randn(B, C, HH, WW, device="cuda") for _ in range(NP)]
zs = [torch.empty_like(xs[0]) for _ in range(NP)]
carry = H.new_carry(w)
H.inverse(xs[0], w, out=zs[0], carry=carry)  # the carry holds the packed adjoint: backward(need_dw=False) is a scan alone


def timed(fn, n=100):
    for _ in range(10):
        fn(0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / n


dx = torch.empty_like(xs[0])
t_same = timed(lambda i: H.backward(xs[0], zs[0], w, dx_out=dx, need_dw=False, carry=carry))
t_rot = timed(lambda i: H.backward(xs[i % NP], zs[i % NP], w, dx_out=zs[(i + 1) % NP], need_dw=False, carry=carry))
t_inv = timed(lambda i: H.inverse(xs[0], w, out=zs[0], carry=carry))
t_inv_rot = timed(lambda i: H.inverse(xs[i % NP], w, out=zs[i % NP], carry=carry))
print("%s: scan alone (dx = A^-T g, packed adjoint from the carry): same buffers %.1f us, %d buffer pairs in rotation %.1f us; "
      "inverse (fold + scan): same %.1f us, rotation %.1f us" % (os.path.basename(H.LIB_PATH), t_same, NP, t_rot, t_inv, t_inv_rot))
