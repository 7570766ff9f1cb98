"""Fixture for the train-step test (tests/test_hip_train.py): one step of a small if_glow_mnist-shaped model (two blocks of
two steps [ActNorm, 2x2 inverse-flow layer, per-element spline, Coupling] with a SplitPrior in between, the reference
experiment's module order, inf/experiments/if_glow_mnist.py:33-132) run in fp64 on the CPU from the REFERENCE'S OWN layers
where they import -- Normalization, LogitTransform, Squeeze, ActNorm, Coupling, SplitPrior, SplineActivation -- and, for the
inverse-flow layer (inf.layers.inv_conv needs the CUDA extension), the exact operator by a dense solve under autograd
(the reference's compute_expensive recipe, inf/layers/selfnorm.py:175-180).  The loss is experiment.py:160-195's
(-(log p + log-det), summed over the batch / len(x)), with every layer's log-det counted once.  Records the input batch
(already dequantised), the state_dict after the data-dependent ActNorm initialisation, the loss and every parameter's
gradient.  Run in the build container only:

    python tests/golden/make_golden_trainstep.py            # the MNIST-Glow shape
    python tests/golden/make_golden_trainstep.py --cifar    # the CIFAR / ImageNet-32 Glow shape (configs[3], configs[4])
"""
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.modules["wandb"] = types.ModuleType("wandb")

from inf.layers.actnorm import ActNorm  # noqa: E402
from inf.layers.activations import SplineActivation  # noqa: E402
from inf.layers.coupling import Coupling  # noqa: E402
from inf.layers.normalize import Normalization  # noqa: E402
from inf.layers.splitprior import SplitPrior  # noqa: E402
from inf.layers.squeeze import Squeeze  # noqa: E402
from inf.layers.transforms import LogitTransform  # noqa: E402


class StdNormal(nn.Module):
    """inf/train/losses.py:21-49 (MultivariateNormal(0, I).log_prob); the reference class pins device='cuda'"""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)

    def log_prob(self, x, context=None):
        return (-0.5 * (math.log(2 * math.pi) + x.pow(2))).flatten(1).sum(-1)


class ExactInvFlowNoPad(nn.Module):
    """z = A^-1 x for the TL-padded conv with the effective weight (solve_mc.py:105-109 semantics), dense, fp64.
    (inv_flow_with_pad(order='TL') and inv_flow_no_pad are the same operator: inf/layers/inv_conv.py:94-364 / 365-513.)"""

    def __init__(self, C, K, gen):
        super().__init__()
        w = nn.init.dirac_(torch.empty(C, C, K, K)).double() + 0.05 * torch.randn(C, C, K, K, generator=gen, dtype=torch.float64)
        w[:, -1, -1, -1] = 1.0
        self.weight_fwd = nn.Parameter(w)
        self.K = K

    def forward(self, x, context=None):
        B, C, H, W = x.shape
        K = self.K
        m = torch.ones(C, C, K, K, dtype=torch.float64)
        m[:, :, -1, -1] = torch.tril(torch.ones(C, C, dtype=torch.float64), -1)
        const = torch.zeros(C, C, K, K, dtype=torch.float64)
        const[:, :, -1, -1] = torch.eye(C, dtype=torch.float64)
        we = self.weight_fwd * m + const
        n = C * H * W
        eye = torch.eye(n, dtype=torch.float64).reshape(n, C, H, W)
        A = F.conv2d(F.pad(eye, (K - 1, 0, K - 1, 0)), we).reshape(n, n).T
        z = torch.linalg.solve(A, x.reshape(B, n).T).T.reshape(B, C, H, W)
        return z, 0.0


def main(cifar=False):
    """cifar=False: the if_glow_mnist shape (per-element splines, 2x2 layers, alpha 1e-7: if_glow_mnist.py:33-132);
    cifar=True: the if_glow_cifar / if_glow_imagenet32 shape (three colour channels, 3x3 layers, ONE shared spline of 10 bins
    and tail bound 20 per step, none behind the last step, alpha 1e-6: if_glow_cifar.py:23-100)."""
    torch.set_num_threads(4)
    gen = torch.Generator().manual_seed(78 if cifar else 77)
    torch.manual_seed(78 if cifar else 77)
    B, size = (4, (3, 8, 8)) if cifar else (6, (1, 8, 8))
    width, nb, tb = (16, 10, 20) if cifar else (16, 5, 20)
    alpha = 1e-6 if cifar else 1e-7
    K = 3 if cifar else 2
    layers = [Normalization(translation=0, scale=256), Normalization(translation=-alpha, scale=1 / (1 - 2 * alpha)), LogitTransform()]
    cur = size
    for block in range(2):
        layers.append(Squeeze())
        cur = (cur[0] * 4, cur[1] // 2, cur[2] // 2)
        for k in range(2):
            layers.append(ActNorm(cur[0]))
            layers.append(ExactInvFlowNoPad(cur[0], K, gen))
            if not (cifar and block == 1 and k == 1):
                layers.append(SplineActivation(cur, n_bins=nb, tail_bound=tb, individual_weights=not cifar))
            layers.append(Coupling(cur, width=width))
        if block == 0:
            layers.append(SplitPrior(cur, StdNormal, width=width))
            cur = (cur[0] // 2, cur[1], cur[2])
    base = StdNormal(cur)
    model = nn.ModuleDict({str(i): m for i, m in enumerate(layers)}).double()
    # the zero-initialised last convolution of every conditioner would make the couplings the identity: give them values
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("net.4.weight"):
                p.copy_(0.05 * torch.randn(p.shape, generator=gen, dtype=torch.float64))
            if "unnormalized" in name:
                p.copy_(0.3 * torch.randn(p.shape, generator=gen, dtype=torch.float64))
    x = (torch.randint(0, 256, (B, *size), generator=gen).double() + torch.rand(B, *size, generator=gen, dtype=torch.float64))

    def loss_of(x):
        h, logdet = x, 0.0
        for i in range(len(layers)):
            h, ld = model[str(i)](h)
            logdet = logdet + ld
        logp = base.log_prob(h) + logdet
        return (-logp).sum() / len(x)

    with torch.no_grad():
        loss_of(x)  # data-dependent ActNorm initialisation (actnorm.py:21-27)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.zero_grad()
    loss = loss_of(x)
    loss.backward()
    out = {"x": x.numpy(), "loss": float(loss), "config": np.array([B, size[0], size[1], size[2], width, nb, tb])}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    for k, p in model.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    path = os.path.join(HERE, "trainstep_glow_cifar_b4_8x8_L2K2.npz" if cifar else "trainstep_glow_b6_8x8_L2K2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "loss", float(loss), "params", sum(p.numel() for p in model.parameters()), "keys", len(out))


if __name__ == "__main__":
    main(cifar="--cifar" in sys.argv)
