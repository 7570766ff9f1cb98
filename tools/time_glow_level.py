"""One level of the ImageNet-32 Glow of inf/if_multiGPU_imagenet32.py (batch 100, first level: 12 channels at 16x16, 2x2
inverse-flow kernels, spline activations, couplings of width 256), N steps of [inv_flow_with_pad, SplineActivation,
Coupling], every flow layer on the HIP library: forward + backward wall clock, and -- under tools/kstats_any.sh -- where
the device time goes (inverse-conv kernels vs the elementwise passes vs the conditioners' library convolutions)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.layers.activations import SplineActivation
from inf.layers.coupling import Coupling
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import inv_flow_with_pad
from inf.layers.squeeze import Squeeze
from inf.train.losses import NegativeGaussianLoss

torch.manual_seed(0)
B, NSTEP = 100, 8
size = (12, 16, 16)
layers = [Squeeze()]
for k in range(NSTEP):
    layers += [inv_flow_with_pad(size[0], size[0], (2, 2), order="TL"), SplineActivation(size), Coupling(size, width=256)]
model = FlowSequential(NegativeGaussianLoss(size=size), *layers).cuda()
x = torch.randn(B, 3, 32, 32, device="cuda")


def step():
    for p in model.parameters():
        p.grad = None
    z, lp = model(x)
    (-(lp.sum() / B)).backward()


for _ in range(5):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print("Glow level, %d steps of [inv_flow 2x2, spline, coupling(256)] at (%d, 12, 16, 16): %.2f ms per forward+backward, %.3f ms per step"
      % (NSTEP, B, ms, ms / NSTEP))

