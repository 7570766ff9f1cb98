"""Time the Glow-step neighbours (csrc/glow_step.hip) at the north-star activation size and next to the eager torch
expressions of the reference layers (inf/layers/actnorm.py, squeeze.py, coupling.py).  Prints us and the HBM rate of
the algorithmic bytes (reads + writes of whole activations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
import invflow_hip as H

B, C, HH, WW = 128, 64, 32, 32
N = B * C * HH * WW * 4  # bytes of one activation
torch.manual_seed(0)
x = torch.randn(B, C, HH, WW, device="cuda"); h = torch.randn_like(x); gy = torch.randn_like(x)
gld = torch.randn(B, device="cuda")
t = torch.randn(C, device="cuda"); ls = torch.randn(C, device="cuda") * 0.3


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def eager_actnorm():
    tv, lv = t.view(1, C, 1, 1), ls.view(1, C, 1, 1)
    return (x - tv) * torch.exp(-lv), -ls.sum().expand(B) * HH * WW


def eager_s2d():
    return x.view(B, C, HH // 2, 2, WW // 2, 2).permute(0, 1, 3, 5, 2, 4).contiguous().view(B, 4 * C, HH // 2, WW // 2)


def eager_coupling():
    x1, x2 = x[:, :C // 2], x[:, C // 2:]
    log_s = 2. * torch.tanh(h[:, ::2] / 2.)
    return torch.cat([x1, x2 * torch.exp(log_s) + h[:, 1::2]], dim=1), log_s.flatten(start_dim=1).sum(-1)


rows = [
    ("actnorm forward", lambda: H.actnorm(x, t, ls), 2 * N, eager_actnorm),
    ("actnorm backward", lambda: H.actnorm_backward(gy, gld, x, t, ls), 3 * N, None),
    ("actnorm stats (init)", lambda: H.actnorm_stats(x), 2 * N, None),
    ("space_to_depth", lambda: H.space_to_depth(x), 2 * N, eager_s2d),
    ("coupling forward", lambda: H.coupling(x, h), 3 * N, eager_coupling),
    ("coupling backward", lambda: H.coupling_backward(gy, gld, x, h), int(4.5 * N), None),
]
print("activation (%d,%d,%d,%d) fp32 = %.1f MB" % (B, C, HH, WW, N / 1e6))
for name, fn, nbytes, eager in rows:
    us = timeit(fn)
    line = "%-22s %8.1f us  %6.2f TB/s (%.0f MB algorithmic)" % (name, us, nbytes / us / 1e6, nbytes / 1e6)
    if eager is not None:
        line += "   eager torch: %8.1f us" % timeit(eager)
    print(line)

# ---- activations --------------------------------------------------------------------------------------------------
import numpy as np
from inf.layers.activations import SplineActivation, SmoothLeakyRelu, spline_tables
sp = SplineActivation((C, HH, WW)).cuda()
cw, ch, dv = [t.detach() for t in spline_tables(sp.unnormalized_widths, sp.unnormalized_heights, sp.unnormalized_derivatives, 10.0)]
xs = x * 3


def eager_slr():
    return 0.3 * xs + 0.7 * torch.logsumexp(torch.stack((torch.zeros_like(xs), xs)), dim=0), torch.log(0.3 + 0.7 * torch.sigmoid(xs)).flatten(1).sum(-1)


rows = [
    ("slr forward", lambda: H.slr(xs, 0.3), 2 * N, eager_slr),
    ("slr backward", lambda: H.slr_backward(gy, gld, xs, 0.3), 3 * N, None),
    ("slr reverse (100 Newton its)", lambda: H.slr(xs, 0.3, reverse=True), 2 * N, None),
    ("spline forward", lambda: H.rqspline(xs, cw, ch, dv, 10.0), 2 * N, None),
    ("spline backward", lambda: H.rqspline_backward(gy, gld, xs, cw, ch, dv, 10.0), 3 * N, None),
    ("spline inverse", lambda: H.rqspline(xs, cw, ch, dv, 10.0, inverse=True), 2 * N, None),
]
for name, fn, nbytes, eager in rows:
    us = timeit(fn)
    line = "%-30s %8.1f us  %6.2f TB/s (%.0f MB algorithmic)" % (name, us, nbytes / us / 1e6, nbytes / 1e6)
    if eager is not None:
        line += "   eager torch: %8.1f us" % timeit(eager)
    print(line)
