"""Uniform noise on [0, 1]^d for dequantisation (reference surface: inf/layers/distributions/uniform.py:6-39).
A caller of the path (BASELINE configs 3-5 dequantise their integer images): plain torch."""
import numpy as np
import torch
import torch.nn as nn


class UniformDistribution(nn.Module):
    def __init__(self, size):
        super().__init__()
        self.size = tuple(size)
        self.dim = int(np.prod(size))
        self.register_buffer("empty", torch.zeros(1))

    def forward(self, input, context=None):
        return self.log_prob(input, context)

    def log_prob(self, input, context=None):
        inside = (input >= 0) & (input <= 1.0)
        log_px = torch.where(inside, torch.zeros_like(input), torch.full_like(input, -1e30))
        return log_px.view(log_px.size(0), self.dim).sum(-1)

    def sample(self, n_samples, context=None):
        return torch.rand((n_samples, *self.size), device=self.empty.device), 0.0
