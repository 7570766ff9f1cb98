"""Regenerate tests/golden/state_dict_glow_L2K2.npz: the state dict (keys, shapes and values) of a small if_glow_mnist
model assembled from the REFERENCE'S OWN layer classes, in the order of its create_model (inf/experiments/
if_glow_mnist.py:34-130: dequantize, normalise x2, logit, then per block squeeze + block_size x [ActNorm,
inv_flow_with_pad, inv_flow_no_pad, SplineActivation, Coupling], SplitPrior between blocks, Gaussian base).  The package's
create_model must load it with strict=True (tests/test_checkpoint.py) -- that is what "a reference checkpoint
(inf/train/experiment.py:475-502) loads unchanged" means.  Run in the build container only; the fixture is data.

The experiment module itself does not import here (wandb, torchvision, the CUDA extension), so the layers are imported one
by one with three empty stand-in modules in sys.modules: `wandb` (logging only), `inv_conv_with_bp` (the CUDA extension,
only called inside forward/backward, never at construction) and `inf.utils.convbackward` (import-time JIT build into the
read-only tree, used by SelfNormConv's autograd only); the Gaussian base distribution is a stateless placeholder (below).
No reference code is executed beyond constructors and state_dict().

    python tests/golden/make_golden_state_dict.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
for name in ("wandb", "inv_conv_with_bp"):
    sys.modules[name] = types.ModuleType(name)
cb = types.ModuleType("inf.utils.convbackward")
cb.conv2d_backward = None
cb.conv_bias_map = None
sys.modules["inf.utils.convbackward"] = cb

from inf.layers import Dequantization, Normalization  # noqa: E402
from inf.layers.actnorm import ActNorm  # noqa: E402
from inf.layers.activations import SplineActivation  # noqa: E402
from inf.layers.coupling import Coupling  # noqa: E402
from inf.layers.distributions.uniform import UniformDistribution  # noqa: E402
from inf.layers.flowsequential import FlowSequential  # noqa: E402
from inf.layers.inv_conv import inv_flow_no_pad, inv_flow_with_pad  # noqa: E402
from inf.layers.splitprior import SplitPrior  # noqa: E402
from inf.layers.squeeze import Squeeze  # noqa: E402
from inf.layers.transforms import LogitTransform  # noqa: E402


class NegativeGaussianLoss(torch.nn.Module):
    """inf/train/losses.py:21-31 builds its two constants with device='cuda' in the constructor (no GPU here) and registers
    neither parameters nor buffers: the base distribution contributes no state-dict entry, which is all this stands for."""

    def __init__(self, size):
        super().__init__()
        self.size = size


CFG = dict(image_size=(1, 8, 8), num_blocks=2, block_size=2, if_kernel_size=3, coupling_width=16, n_bins=5, tail_bound=20)


def reference_model(cfg):
    alpha = 1e-7
    size = tuple(cfg["image_size"])
    layers = [Dequantization(UniformDistribution(size=size)), Normalization(translation=0, scale=256),
              Normalization(translation=-alpha, scale=1 / (1 - 2 * alpha)), LogitTransform()]
    for block in range(cfg["num_blocks"]):
        layers.append(Squeeze())
        size = (size[0] * 4, size[1] // 2, size[2] // 2)
        for _ in range(cfg["block_size"]):
            layers.append(ActNorm(size[0]))
            layers.append(inv_flow_with_pad(size[0], size[0], (cfg["if_kernel_size"],) * 2, order="TL"))
            layers.append(inv_flow_no_pad(size[0], size[0], (2, 2)))
            layers.append(SplineActivation(size, n_bins=cfg["n_bins"], tail_bound=cfg["tail_bound"], individual_weights=True))
            layers.append(Coupling(size, width=cfg["coupling_width"]))
        if block < cfg["num_blocks"] - 1:
            layers.append(SplitPrior(size, NegativeGaussianLoss, width=cfg["coupling_width"]))
            size = (size[0] // 2, size[1], size[2])
    return FlowSequential(NegativeGaussianLoss(size=size), *layers)


def main():
    torch.manual_seed(11)
    model = reference_model(CFG)
    gen = torch.Generator().manual_seed(12)
    # constructors leave many tensors at zero / identity: perturb the float ones IN PLACE so that a load is visible in the
    # outputs (in place: Conv2dZeros' `bias` and `logs` are two Parameters over one tensor, coupling.py:30-36, and stay so)
    with torch.no_grad():
        for key, t in model.state_dict().items():
            if t.dtype.is_floating_point and t.numel() > 1:
                t.add_(0.01 * torch.randn(t.shape, generator=gen))
            if key.endswith(".initialized"):
                t.fill_(1)  # as in a trained checkpoint: ActNorm's data-dependent init (actnorm.py:21-27) has run
    state = model.state_dict()
    out = {key: t.numpy().copy() for key, t in state.items()}
    np.savez_compressed(os.path.join(HERE, "state_dict_glow_L2K2.npz"), __keys__=np.array(list(state.keys())), **out)
    print(len(out), "entries,", sum(v.size for v in out.values()), "values")
    for k, v in out.items():
        print(" ", k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
