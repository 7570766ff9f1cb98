"""Inverse-of-convolution flow layers on MI355X (reference: inf/layers/inv_conv.py).

Same operator surface as the reference -- `inv_conv_4d`, `inv_flow_with_pad`, `inv_flow_no_pad`,
parameter name `weight_fwd`, `forward(input, context, compute_expensive) -> (out, ldj)`,
`reverse`, `logdet`, `get_mask`, `reset_gradients`, `reset_parameters` -- but the arithmetic runs
in libinvflow_hip.so (hand-written HIP, gfx950) through the C ABI of include/invflow.h:

  layer forward  x -> z = A^-1 x     ifl_inverse_f32   (ref: inv_conv_.forward, inv_conv.py:46-60)
  autograd bwd   (dx, dW) fused      ifl_backward_f32  (ref: inv_conv_.backward, inv_conv.py:62-81)
  layer reverse  z -> x = A z        ifl_forward_f32   (ref: reverse, inv_conv.py:249-267,442-460)

Semantics are those of the reference's exact CPU solver (inf/utils/solve_mc.py:88-114).
Documented deviations from the reference *layer* code: the non-TL orders are
handled functionally by index reflection inside the kernels (the reference flips
`weight_fwd.data` in place on every call, inv_conv.py:200-212, and forgets the flips in
`reverse`); gradients are the true ones and already masked; log|det| is the exact 0.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

import invflow_hip as _h

from .flowlayer import FlowLayer, mark_expensive

_ORDER_FLIP_DIMS = {"TL": (), "TR": (3,), "BL": (2,), "BR": (2, 3)}


def flip_kernel(W):
    """Kernel of the transposed convolution (inv_conv.py:39-40)."""
    return torch.flip(W, (2, 3)).permute(1, 0, 2, 3).clone()


# Inside a torch.autocast region (the reference's bf16 training configs) the arithmetic of these layers stays fp32:
# tensor arguments are cast to float32 on the way in, gradients come back in float32.  Outside autocast the activation's
# storage format decides, as the reference's kernels are dispatched on the tensor's dtype: bf16 activations (weights stay
# fp32) take the library's bf16 entry points and stay bf16.
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


class inv_conv_(torch.autograd.Function):
    """z = A^-1 x with the true gradients; `order` / `flags` / recon settings are non-tensor args."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, W, order="TL", flags=0, recon_weight=0.0):
        x = x.contiguous()
        Wc = W.contiguous()
        need_bwd = any(ctx.needs_input_grad[:2])
        # side channel of this step: adjoint weights + max|z| from the forward, so the backward neither folds
        # again nor scans z and dx (ifl_carry_bytes, include/invflow.h)
        carry = _h.new_carry(Wc) if need_bwd else None
        z = _h.inverse(x, Wc, order, flags, carry=carry)
        ctx.order, ctx.flags, ctx.recon_weight = order, flags, float(recon_weight)
        ctx.carry = carry
        ctx.w_version = Wc._version
        if recon_weight != 0.0:
            ctx.save_for_backward(Wc, z, x)
        else:
            ctx.save_for_backward(Wc, z)
        ctx.recon_loss = None
        return z

    @staticmethod
    @_bwd32
    def backward(ctx, output_grad):
        saved = ctx.saved_tensors
        Wc, z = saved[0], saved[1]
        x = saved[2] if len(saved) > 2 else None
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx, dw, rl = _h.backward(output_grad.contiguous(), z, Wc, ctx.order, ctx.flags, x=x,
                                 recon_weight=ctx.recon_weight, need_dx=need_dx, need_dw=need_dw,
                                 carry=ctx.carry if Wc._version == ctx.w_version else None)
        ctx.recon_loss = rl
        return dx, dw, None, None, None

    @staticmethod
    def clip_gradients(module, clip_value):
        for p in module.parameters():
            if p.grad is not None:
                p.grad.data.clamp_(-clip_value, clip_value)


def inv_conv_4d(x, W, order="TL", flags=0, recon_weight=0.0):
    """Functional form (inv_conv.py:89-91); the extra arguments default to the reference call."""
    return inv_conv_.apply(x, W, order, flags, recon_weight)


def _init_weight(out_channels, in_channels, kernel_size, reference_init=False):
    """Identity + small noise, or a random orthogonal matrix for 1x1 (inv_conv.py:149-165).

    Where the identity sits is the one deliberate difference from the reference's recipe.  nn.init.dirac_ puts it at the
    kernel's CENTRE (inv_conv.py:154); the exact operator has its unit diagonal at the LAST tap (solve_mc.py:105-109), so
    for a 3x3 kernel the reference's initial layer is the identity PLUS a one-pixel diagonal shift: its inverse amplifies
    about 6x per layer and a Glow of 32 such layers without ActNorm (if_glow_cifar.py:147, if_multiGPU_imagenet32.py:284-345)
    leaves fp32 on its first batch.  Default here: the identity at the operator's diagonal tap (the layer starts as the
    identity map, like every other Glow layer); `reference_init=True` reproduces inv_conv.py:153-170 tap for tap.  For 2x2
    kernels the two coincide.  Shapes and state-dict keys are the same either way."""
    w_shape = (out_channels, in_channels, *kernel_size)
    if kernel_size[0] == 1 and kernel_size[1] == 1:
        q = np.linalg.qr(np.random.randn(out_channels, in_channels))[0]
        return torch.tensor(q).to(torch.float).view(w_shape)
    if reference_init:
        w = nn.init.dirac_(torch.empty(w_shape))
    else:
        w = torch.zeros(w_shape)
        for c in range(min(out_channels, in_channels)):
            w[c, c, -1, -1] = 1.0
    return w + nn.init.xavier_normal_(torch.empty(w_shape), gain=0.01)


class _InvFlowBase(FlowLayer):
    order = "TL"

    def __init__(self, in_channels, out_channels, kernel_size, sym_recon_grad=False, only_R_recon=False,
                 recon_loss_weight=1.0, recon_loss_lr=0.0, recon_alpha=0.9, reference_init=False):
        super().__init__()
        self.reference_init = reference_init  # (see _init_weight)
        assert len(kernel_size) == 2
        assert in_channels == out_channels, "an invertible convolution needs in_channels == out_channels"
        self.kernel_size = _pair(kernel_size)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.sym_recon_grad = sym_recon_grad
        self.only_R_recon = only_R_recon
        self.recon_loss_weight = recon_loss_weight
        self.recon_loss_lr = recon_loss_lr
        self.recon_loss_ema = None
        self.alpha = recon_alpha
        self.flags = 0
        self.reset_parameters()

    def reset_parameters(self):
        self.logabsdet_dirty = True
        w = _init_weight(self.out_channels, self.in_channels, self.kernel_size, self.reference_init)
        # the reference pins the last input channel of the diagonal tap (inv_conv.py:168-170); the
        # exact solver ignores it (unit diagonal, solve_mc.py:105-109) -- kept for state-dict parity
        w[:, -1, -1, -1] = 1.0
        dims = _ORDER_FLIP_DIMS[self.order]
        if dims:
            w = torch.flip(w, dims)  # stored pre-flipped for the order (inv_conv.py:172-179)
        self.weight_fwd = nn.Parameter(w.contiguous())

    # -- the flow -----------------------------------------------------------------------------
    def forward(self, input, context=None, compute_expensive=False):
        if self.training:
            self.logabsdet_dirty = True
        self.input = input
        self.output = inv_conv_4d(input, self.weight_fwd, self.order, self.flags)
        # log|det A| is exactly 0 (unit lower-triangular operator): same value on both branches
        ldj = self.logdet(input, context, compute_expensive) if compute_expensive else 0.0
        return self.output, ldj

    def reverse(self, input, context=None, compute_expensive=False):
        with torch.no_grad():
            return _h.forward(input.contiguous(), self.weight_fwd.detach().contiguous(), self.order, self.flags)

    @mark_expensive
    def logdet(self, input, context=None, compute_expensive=False):
        if compute_expensive:
            return input.new_zeros(len(input))
        return 0.0

    # -- gradient mask (inv_conv.py:223-248) -----------------------------------------------------
    def get_mask(self):
        mask = torch.ones_like(self.weight_fwd.data)
        keep_diag = 1 if (self.flags & _h.FLAG_GENERAL_DIAG) else 0
        for c_out in range(mask.shape[0]):
            mask[c_out, c_out + keep_diag:, -1, -1] = 0.0
        dims = _ORDER_FLIP_DIMS[self.order]
        return torch.flip(mask, dims) if dims else mask

    def reset_gradients(self):
        """Mask the gradient IN PLACE (inv_conv.py:223-230 assigns a new tensor: that would take .grad out of a flat gradient
        bucket -- inf/train/step.py -- and out of a captured graph's addresses).  The mask is cached per device, so the
        masking is one multiplication that captures into a graph."""
        g = self.weight_fwd.grad
        if g is not None:
            m = getattr(self, "_grad_mask", None)
            if m is None or m.device != g.device or m.dtype != g.dtype:
                m = self._grad_mask = self.get_mask().to(device=g.device, dtype=g.dtype)
            g.mul_(m)

    def add_recon_grad(self, recon_loss_weight_update=None):
        """||x - A A^-1 x||^2 is zero up to rounding for the exact inverse: the recon gradient of this
        layer is computed inside the fused backward when requested (ifl_backward_f32 `recon_weight`);
        this method exists for surface parity (inv_conv.py:269-320) and returns the last recon loss."""
        if recon_loss_weight_update is not None:
            self.recon_loss_weight = recon_loss_weight_update
        return torch.zeros((), device=self.weight_fwd.device)

    def extra_repr(self):
        return "{}, {}, kernel_size={}, order={}".format(self.in_channels, self.out_channels, self.kernel_size,
                                                         self.order)


class inv_flow_with_pad(_InvFlowBase):
    """Inverse-conv layer with an explicit padding corner (inv_conv.py:94-364)."""

    def __init__(self, in_channels, out_channels, kernel_size, order="TL", **kw):
        assert order in {"TL", "TR", "BL", "BR"}, "unknown order: {}".format(order)
        self.order = order
        K_H, K_W = kernel_size[0], kernel_size[1]
        # (left, right, top, bottom), as F.pad takes it (inv_conv.py:126-144)
        self.pad = {"TL": (K_W - 1, 0, K_H - 1, 0), "TR": (0, K_W - 1, K_H - 1, 0),
                    "BL": (K_W - 1, 0, 0, K_H - 1), "BR": (0, K_W - 1, 0, K_H - 1)}[order]
        super().__init__(in_channels, out_channels, kernel_size, **kw)
        self.mask = self.get_mask()


class inv_flow_no_pad(_InvFlowBase):
    """The TL-only variant used by the Glow experiments (inv_conv.py:365-513)."""
    order = "TL"
