"""Squeeze / UnSqueeze on the HIP library (reference: inf/layers/squeeze.py:5-52): the space-to-depth permutation
as one pass (ifl_squeeze_f32) instead of view + permute + contiguous; zero log-det."""
import torch

import invflow_hip as H

from .flowlayer import FlowLayer


class _S2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return H.space_to_depth(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return H.depth_to_space(g.contiguous())


class _D2S(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return H.depth_to_space(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return H.space_to_depth(g.contiguous())


def _hip_ok(x):
    return x.dim() == 4 and x.is_cuda and x.dtype == torch.float32


def space_to_depth(x):
    if _hip_ok(x):
        return _S2D.apply(x)
    xs = x.size()  # squeeze.py:5-13
    x = x.view(xs[0], xs[1], xs[2] // 2, 2, xs[3] // 2, 2)
    x = x.permute((0, 1, 3, 5, 2, 4)).contiguous()
    return x.view(xs[0], xs[1] * 4, xs[2] // 2, xs[3] // 2)


def depth_to_space(x):
    if _hip_ok(x):
        return _D2S.apply(x)
    xs = x.size()  # squeeze.py:16-25
    x = x.view(xs[0], xs[1] // 4, 2, 2, xs[2], xs[3])
    x = x.permute((0, 1, 4, 2, 5, 3)).contiguous()
    return x.view(xs[0], xs[1] // 4, xs[2] * 2, xs[3] * 2)


class Squeeze(FlowLayer):
    def forward(self, input, context=None):
        return space_to_depth(input), self.logdet(input, context)

    def reverse(self, input, context=None):
        return depth_to_space(input)

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))


class UnSqueeze(FlowLayer):
    def forward(self, input, context=None):
        return depth_to_space(input), self.logdet(input, context)

    def reverse(self, input, context=None):
        return space_to_depth(input)

    def logdet(self, input, context=None):
        return input.new_zeros(len(input))
