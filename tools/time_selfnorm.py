"""SelfNormConv 3x3 (inf/layers/selfnorm.py) forward + self-normalised backward at B=128, C=64, 32x32 (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
import torch
from bench import B, C, HH, WW
from inf.layers.selfnorm import SelfNormConv
torch.manual_seed(0)
layer = SelfNormConv(C, C, (3, 3), bias=True, stride=1, padding=1).cuda()
x = torch.randn(B, C, HH, WW, device="cuda", requires_grad=True)
g = torch.randn(B, C, HH, WW, device="cuda")
def step():
    for p in layer.parameters(): p.grad = None
    x.grad = None
    z, _ = layer(x)
    z.backward(g)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
print("SelfNormConv fwd + bwd: %.3f ms/step, %.0f images/s" % (ms, B / ms * 1e3))
