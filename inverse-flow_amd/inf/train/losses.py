"""Diagonal-Gaussian base distribution (reference: inf/train/losses.py:21-49), the only piece of
the training harness FlowSequential needs."""
import math

import torch
import torch.nn as nn


class NegativeGaussianLoss(nn.Module):
    """Standard normal over a tensor of shape `size` (per-sample log-probability / sampling)."""

    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)
        self.dim = 1
        for s in self.size:
            self.dim *= int(s)
        self.register_buffer("_device_probe", torch.zeros(1), persistent=False)

    def forward(self, input, context=None):
        return -self.log_prob(input, context).sum(-1)

    def log_prob(self, input, context=None, sum=True):
        p = -0.5 * (math.log(2 * math.pi) + input.pow(2))
        return p.flatten(start_dim=1).sum(-1) if sum else p

    def sample(self, n_samples, context=None):
        x = torch.randn(n_samples, *self.size, device=self._device_probe.device)
        return x, self.log_prob(x)
