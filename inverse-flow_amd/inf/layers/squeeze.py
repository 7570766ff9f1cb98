"""Squeeze / UnSqueeze on the HIP library (reference surface: inf/layers/squeeze.py:5-52).

space_to_depth maps (B, C, H, W) to (B, 4C, H/2, W/2) with output channel 4c + 2dy + dx holding x[c, 2h+dy, 2w+dx]
(the ordering of squeeze.py:5-13) -- which is torch's pixel_unshuffle with factor 2; depth_to_space is its inverse
(pixel_shuffle).  CUDA fp32 / bf16 tensors take one pass of libinvflow_hip (ifl_squeeze_f32 / _bf16) in either direction, and the
gradient of one permutation is the other; everything else uses the torch primitives.  The log-det is zero."""
import torch
import torch.nn.functional as F

import invflow_hip as H

from .flowlayer import FlowLayer


class _Permute(torch.autograd.Function):
    """to_depth = True: space_to_depth; False: depth_to_space.  Backward: the opposite direction."""

    @staticmethod
    def forward(ctx, x, to_depth):
        ctx.to_depth = to_depth
        x = x.contiguous()
        return H.space_to_depth(x) if to_depth else H.depth_to_space(x)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (H.depth_to_space(g) if ctx.to_depth else H.space_to_depth(g)), None


def _on_library(x):
    return x.dim() == 4 and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)


def space_to_depth(x):
    return _Permute.apply(x, True) if _on_library(x) else F.pixel_unshuffle(x, 2)


def depth_to_space(x):
    return _Permute.apply(x, False) if _on_library(x) else F.pixel_shuffle(x, 2)


class _SqueezeBase(FlowLayer):
    _down = True  # forward squeezes space into depth

    def forward(self, input, context=None):
        out = space_to_depth(input) if self._down else depth_to_space(input)
        return out, self.logdet(input, context)

    def reverse(self, input, context=None):
        return depth_to_space(input) if self._down else space_to_depth(input)

    def logdet(self, input, context=None):
        return input.new_zeros(input.shape[0])


class Squeeze(_SqueezeBase):
    _down = True


class UnSqueeze(_SqueezeBase):
    _down = False
