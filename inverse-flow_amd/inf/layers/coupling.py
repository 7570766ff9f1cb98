"""Affine coupling on the HIP library (reference: inf/layers/coupling.py:9-102).

The conditioner `net` (3x3 conv -> ReLU -> 1x1 conv -> ReLU -> Conv2dZero) is three library convolutions and
stays in torch; everything that touches the activation elementwise -- the strided split of h into (h_s, t),
log_s = 2 tanh(h_s / 2), z2 = x2 exp(log_s) + t, the concatenation and the per-image sum of log_s -- is one pass
of libinvflow_hip (ifl_coupling_f32), and one more for the backward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

import invflow_hip as H

from .flowlayer import FlowLayer

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


class Conv2dZero(nn.Module):
    """coupling.py:9-45: zero-initialised convolution with a learned per-channel log-scale."""

    def __init__(self, in_channels, out_channels, bias=True, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                 dilation=1, groups=1, logscale_factor=3):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.logscale_factor = logscale_factor
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels, *self.kernel_size))
        zeros = torch.zeros(out_channels)
        self.bias = nn.Parameter(zeros) if bias else None
        self.logs = nn.Parameter(zeros)  # (the reference shares this tensor with the bias: coupling.py:33-37)

    def forward(self, input):
        output = F.conv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return output * torch.exp(self.logs * self.logscale_factor).view(1, -1, 1, 1)


class _AffineFn(torch.autograd.Function):
    """(x, h) -> (cat(x1, x2 exp(log_s) + t), sum log_s); the gradient w.r.t. x is the direct part only: autograd
    adds the part through h = net(x1) itself."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, h):
        x, h = x.contiguous(), h.contiguous()
        y, ld = H.coupling(x, h)
        ctx.save_for_backward(x, h)
        return y, ld

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, h = ctx.saved_tensors
        return H.coupling_backward(gy.contiguous(), None if gld is None else gld.contiguous(), x, h)


def _hip_ok(x):
    return x.dim() == 4 and x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16)


class Coupling(FlowLayer):
    def __init__(self, input_size, width=512, n_context=None):
        super().__init__()
        self.n_channels = n_channels = input_size[0]
        self.half_channels = n_channels // 2
        self.width = width
        if n_context is not None:
            in_channels = self.half_channels + n_context
            self.uses_context = True
        else:
            in_channels = self.half_channels
            self.uses_context = False
        self.net = nn.Sequential(nn.Conv2d(in_channels, width, kernel_size=(3, 3), padding=(1, 1), bias=False),
                                 nn.ReLU(),
                                 nn.Conv2d(width, n_channels, (1, 1), bias=False),
                                 nn.ReLU(),
                                 Conv2dZero(n_channels, n_channels))

    def _h(self, x, context=None):
        assert (context is not None) == self.uses_context
        x1 = x[:, :self.half_channels, :, :]
        if context is not None:
            return self.net(torch.cat([x1, context], dim=1))
        return self.net(x1)

    def get_xs_logs_t(self, x, context=None):
        """coupling.py:66-82 (torch expressions; the HIP path does not materialise these)."""
        h = self._h(x, context)
        x1 = x[:, :self.half_channels, :, :]
        x2 = x[:, self.half_channels:, :, :]
        h_s, t = h[:, ::2], h[:, 1::2]
        logs_range = 2.
        log_s = logs_range * torch.tanh(h_s / logs_range)
        return x1, x2, log_s, t

    def forward(self, input, context=None):
        if _hip_ok(input) and input.size(1) % 2 == 0:
            return _AffineFn.apply(input, self._h(input, context))
        x1, x2, log_s, t = self.get_xs_logs_t(input, context)
        z2 = x2 * torch.exp(log_s) + t
        return torch.cat([x1, z2], dim=1), log_s.flatten(start_dim=1).sum(-1)

    def reverse(self, input, context=None):
        if _hip_ok(input) and input.dtype == torch.float32 and input.size(1) % 2 == 0 and not torch.is_grad_enabled():
            return H.coupling(input.contiguous(), self._h(input, context).float().contiguous(), reverse=True)
        x1, x2, log_s, t = self.get_xs_logs_t(input, context)
        z2 = (x2 - t) * torch.exp(-log_s)
        return torch.cat([x1, z2], dim=1)

    def logdet(self, input, context=None):
        z, ldj = self.forward(input, context)
        return ldj
