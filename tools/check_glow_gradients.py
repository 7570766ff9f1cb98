"""Directional derivatives of a small Glow level (all layers on the HIP library) against central differences, one parameter
tensor at a time (development aid; the assertion form is tests/test_hip_layers.py::test_mini_glow_stack_end_to_end)."""
import sys, os, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "inverse-flow_amd")); sys.path.insert(0, ROOT)
from inf.layers.actnorm import ActNorm
from inf.layers.activations import SmoothLeakyRelu, SplineActivation
from inf.layers.coupling import Coupling
from inf.layers.flowsequential import FlowSequential
from inf.layers.inv_conv import inv_flow_with_pad
from inf.layers.squeeze import Squeeze
from inf.train.losses import NegativeGaussianLoss
torch.manual_seed(3)
B, C, Hh, Ww = 4, 3, 8, 8
size = (4 * C, Hh // 2, Ww // 2)
layers = [Squeeze()]
for k, order in enumerate(["TL", "BR"]):
    layers += [ActNorm(size[0]), inv_flow_with_pad(size[0], size[0], (3, 3), order=order),
               SplineActivation(size, n_bins=5, tail_bound=4.0) if k == 0 else SmoothLeakyRelu(0.3),
               Coupling(size, width=16)]
model = FlowSequential(NegativeGaussianLoss(size=size), *layers).cuda()
x = torch.randn(B, C, Hh, Ww, device="cuda")
with torch.no_grad():
    model(x)
    for m in model.modules():
        if isinstance(m, Coupling):
            for p in m.net.parameters():
                p.add_(0.05 * torch.randn_like(p))
def nll(inp):
    z, lp = model(inp)
    return -(lp.sum() / B)
xg = x.clone().requires_grad_(True)
loss = nll(xg); loss.backward()
print("loss", float(loss))
named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
for eps in (2e-3, 5e-4):
    for n, p in named + [("input", None)]:
        d = torch.randn_like(p if p is not None else x)
        with torch.no_grad():
            if p is None:
                a = float((xg.grad * d).sum()); lp = nll(x + eps * d); lm = nll(x - eps * d)
            else:
                a = float((p.grad * d).sum())
                p.add_(eps * d); lp = nll(x); p.sub_(2 * eps * d); lm = nll(x); p.add_(eps * d)
        num = float(lp - lm) / (2 * eps)
        flag = "" if abs(a - num) < 0.02 * max(1, abs(num)) + 0.05 else "   <<<<"
        print("eps %.0e %-40s analytic %12.4f numeric %12.4f%s" % (eps, n, a, num, flag))
