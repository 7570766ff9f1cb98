"""Does a stack of this library's layers capture into one HIP graph (forward + backward) with
torch.cuda.make_graphed_callables?  (development aid: which layer types, what speed-up)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.layers.activations import SplineActivation, SmoothLeakyRelu
from inf.layers.coupling import Coupling
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import inv_flow_with_pad
from inf.train.losses import NegativeGaussianLoss

which = sys.argv[1] if len(sys.argv) > 1 else "inv+spline"
torch.manual_seed(0)
B, size, N = 100, (12, 16, 16), 8
layers = []
for k in range(N):
    if "inv" in which:
        layers.append(inv_flow_with_pad(12, 12, (2, 2), order="TL"))
    if "spline" in which:
        layers.append(SplineActivation(size))
    if "slr" in which:
        layers.append(SmoothLeakyRelu(0.3))
    if "coupling" in which:
        layers.append(Coupling(size, width=256))
model = FlowSequential(NegativeGaussianLoss(size=size), *layers).cuda()
x = torch.randn(B, *size, device="cuda", requires_grad=True)


def run(fn, n=20):
    def step():
        for p in model.parameters():
            p.grad = None
        z, lp = fn(x)
        (-(lp.sum() / B)).backward()
    for _ in range(5):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# One mode per process: make_graphed_callables replaces the module's forward, and it has to run before any eager backward
# (parameters whose AccumulateGrad nodes were created on the default stream make the autograd engine insert cross-stream
# waits into the backward capture, which breaks it).
mode = sys.argv[2] if len(sys.argv) > 2 else "graph"
if mode == "graph":
    fn = torch.cuda.make_graphed_callables(model, (x,))
else:
    fn = model
print("%-22s %-6s %.3f ms per forward+backward" % (which, mode, run(fn)), flush=True)
