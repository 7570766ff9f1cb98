"""Input preprocessing of the Glow models -- Dequantization, Normalization, LogitTransform (reference surface:
inf/layers/dequantize.py:6-38, normalize.py:6-36, transforms.py:6-19).  Callers of the path (BASELINE configs 3-5), one
elementwise torch expression each, with the reference's log-dets."""
import torch

from .flowlayer import PreprocessingFlowLayer


class Dequantization(PreprocessingFlowLayer):
    """x + u, u ~ deq_distribution on [0, 1]^d; log-det term = -log q(u) (0 for the uniform)."""

    def __init__(self, deq_distribution):
        super().__init__()
        self.distribution = deq_distribution

    def forward(self, input, context=None):
        noise, log_qnoise = self.distribution.sample(input.size(0), input.float())
        if torch.is_tensor(log_qnoise):
            log_qnoise = log_qnoise.to(input.device)
        return input + noise.to(input.device), -log_qnoise

    def reverse(self, input, context=None):
        return input.floor()

    def logdet(self, input, context=None):
        raise NotImplementedError


class Normalization(PreprocessingFlowLayer):
    """(x - translation) / scale; log-det = -C H W log(scale) per image."""

    def __init__(self, translation, scale, learnable=False):
        super().__init__()
        if learnable:
            self.translation = torch.nn.Parameter(torch.Tensor([translation]))
            self.scale = torch.nn.Parameter(torch.Tensor([scale]))
        else:
            self.register_buffer("translation", torch.Tensor([translation]))
            self.register_buffer("scale", torch.Tensor([scale]))

    def forward(self, input, context=None):
        return (input - self.translation) / self.scale, self.logdet(input, context)

    def reverse(self, input, context=None):
        return input * self.scale + self.translation

    def logdet(self, input, context=None):
        N, C, H, W = input.size()
        return (-C * H * W * torch.log(self.scale)).expand(N)


class LogitTransform(PreprocessingFlowLayer):
    """log x - log(1 - x); log-det = sum(-log x - log(1 - x))."""

    def forward(self, input, context=None):
        return torch.log(input) - torch.log(1 - input), self.logdet(input, context)

    def reverse(self, input, context=None):
        return torch.sigmoid(input)

    def logdet(self, input, context=None):
        return (-torch.log(input) - torch.log(1 - input)).flatten(start_dim=1).sum(-1)
