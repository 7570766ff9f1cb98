"""Wall clock per forward+backward of stacks of ONE layer type at the ImageNet-32 level shape (100, 12, 16, 16): tells a
launch-bound layer (time independent of the tensor size) from a device-bound one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from inf.layers.activations import SplineActivation, SmoothLeakyRelu
from inf.layers.actnorm import ActNorm
from inf.layers.coupling import Coupling
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import inv_flow_with_pad
from inf.train.losses import NegativeGaussianLoss

torch.manual_seed(0)
size = (12, 16, 16)
N = 8
kinds = {
    "inv_flow 2x2": lambda: inv_flow_with_pad(12, 12, (2, 2), order="TL"),
    "inv_flow 3x3": lambda: inv_flow_with_pad(12, 12, (3, 3), order="TL"),
    "spline": lambda: SplineActivation(size),
    "smooth leaky relu": lambda: SmoothLeakyRelu(0.3),
    "actnorm": lambda: ActNorm(12),
    "coupling(256)": lambda: Coupling(size, width=256),
}
for B in (100, 400):
    x = torch.randn(B, *size, device="cuda", requires_grad=True)
    for name, mk in kinds.items():
        model = FlowSequential(NegativeGaussianLoss(size=size), *[mk() for _ in range(N)]).cuda()

        def step():
            for p in model.parameters():
                p.grad = None
            z, lp = model(x)
            (-(lp.sum() / B)).backward()

        for _ in range(5):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 20 / N * 1e6
        print("B=%-4d %-20s %8.1f us per layer forward+backward" % (B, name, us))
