"""Self-normalising convolution on MI355X (reference: inf/layers/selfnorm.py:24-334).

`SelfNormConvFunc` keeps the reference's self-normalised gradient
    dW = (dz x^T - flip(R) * multiple) / 2,   dR = (-dx (Wx)^T + flip(W) * flip(multiple)) / 2
(selfnorm.py:52-90) but its three dense contractions -- conv2d, backward_weight, backward_input,
which the reference routes to cuDNN through inf/utils/convbackward/conv2d_backward.cpp -- run in
libinvflow_hip.so (ifl_conv2d_f32 / ifl_conv2d_wgrad_f32 / ifl_conv2d_igrad_f32).
Only stride 1, dilation 1, groups 1 (all the reference experiments use) is supported.
"""
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

import invflow_hip as _h

from .flowlayer import ModifiedGradFlowLayer, mark_expensive


def flip_kernel(W):
    return torch.flip(W, (2, 3)).permute(1, 0, 2, 3).clone()


@lru_cache(maxsize=128)
def _weight_multiple_host(wshape, H, W, padding):
    """Number of output positions each tap touches = backward_weight(ones, ones)/B
    (selfnorm.py:24-32), in closed form: it depends on the tap only."""
    Co, Ci, KH, KW = wshape
    ph, pw = padding
    OH, OW = H + 2 * ph - KH + 1, W + 2 * pw - KW + 1
    ch = [sum(1 for oh in range(OH) if 0 <= oh - ph + kh < H) for kh in range(KH)]
    cw = [sum(1 for ow in range(OW) if 0 <= ow - pw + kw < W) for kw in range(KW)]
    m = torch.tensor(np.outer(ch, cw), dtype=torch.float32)
    return m.view(1, 1, KH, KW).expand(Co, Ci, KH, KW).contiguous()


def _compute_weight_multiple(wshape, x, padding):
    return _weight_multiple_host(tuple(wshape), x.shape[2], x.shape[3], tuple(padding)).to(x.device)


# Inside a torch.autocast region (the reference's bf16 training configs) the arithmetic of these layers stays fp32:
# tensor arguments are cast to float32 on the way in, gradients come back in float32.
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


class _Conv2dHip(torch.autograd.Function):
    """Plain conv2d with true gradients, all three pieces in HIP (used by add_recon_grad)."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, W, padding):
        x, W = x.contiguous(), W.contiguous()
        ctx.save_for_backward(x, W)
        ctx.padding = padding
        return _h.conv2d(x, W, None, padding)

    @staticmethod
    @_bwd32
    def backward(ctx, go):
        x, W = ctx.saved_tensors
        go = go.contiguous()
        gx = _h.conv2d_igrad(go, W, x.shape, ctx.padding) if ctx.needs_input_grad[0] else None
        gw = _h.conv2d_wgrad(go, x, W.shape, ctx.padding) if ctx.needs_input_grad[1] else None
        return gx, gw, None


def _conv(x, W, padding):
    return _Conv2dHip.apply(x, W, padding)


class SelfNormConvFunc(torch.autograd.Function):
    @staticmethod
    @_fwd32
    def forward(ctx, x, W, bw, R, stride, padding, dilation, groups):
        x, W, R = x.contiguous(), W.contiguous(), R.contiguous()
        z = _h.conv2d(x, W, bw.contiguous() if bw is not None else None, padding)
        ctx.save_for_backward(x, W, bw, R, z)
        ctx.padding = tuple(padding)
        return z

    @staticmethod
    @_bwd32
    def backward(ctx, output_grad):
        x, W, bw, R, output = ctx.saved_tensors
        p = ctx.padding
        go = output_grad.contiguous()
        multiple = _compute_weight_multiple(W.shape, x, p)
        delta_z_xt = _h.conv2d_wgrad(go, x, W.shape, p)
        weight_grad_fwd = (delta_z_xt - flip_kernel(R) * multiple) / 2.0
        input_grad = _h.conv2d_igrad(go, W, x.shape, p)
        Wx = output - bw.view(1, -1, 1, 1) if bw is not None else output
        # (the weight gradient is linear in its first operand: negate the C*C*K*K result, not the activation)
        neg_delta_x_Wxt = -_h.conv2d_wgrad(input_grad, Wx.contiguous(), R.shape, p)
        weight_grad_inv = (neg_delta_x_Wxt + flip_kernel(W) * flip_kernel(multiple)) / 2.0
        bw_grad = go.flatten(2).sum(-1).sum(0) if bw is not None else None
        return input_grad, weight_grad_fwd, bw_grad, weight_grad_inv, None, None, None, None


def selfnorm_conv_2d(x, W, bw, R, stride, padding, dilation=1, groups=1):
    return SelfNormConvFunc.apply(x, W, bw, R, stride, padding, dilation, groups)


class SelfNormConv(ModifiedGradFlowLayer):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, stride=1, padding=0, dilation=1,
                 groups=1, sym_recon_grad=False, only_R_recon=False, recon_loss_weight=1.0, recon_loss_lr=0.0,
                 recon_alpha=0.9):
        super().__init__()
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        if self.stride != (1, 1) or self.dilation != (1, 1) or groups != 1:
            raise NotImplementedError("SelfNormConv on MI355X supports stride=1, dilation=1, groups=1")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.sym_recon_grad = sym_recon_grad
        self.only_R_recon = only_R_recon
        self.recon_loss_weight = recon_loss_weight
        self.recon_loss_lr = recon_loss_lr
        self.recon_loss_ema = None
        self.alpha = recon_alpha
        self.use_bias = bias
        self.reset_parameters()

    def reset_parameters(self):
        self.logabsdet_dirty = True
        w_shape = (self.out_channels, self.in_channels, *self.kernel_size)
        w_noise = nn.init.xavier_normal_(torch.empty(w_shape), gain=0.01)
        if self.kernel_size == (1, 1):
            q = np.linalg.qr(np.random.randn(self.out_channels, self.in_channels))[0]
            w_init = torch.tensor(q).to(torch.float).view(w_shape)
        else:
            w_init = nn.init.dirac_(torch.empty(w_shape)) + w_noise
        self.weight_fwd = nn.Parameter(w_init)
        self.weight_inv = nn.Parameter(flip_kernel(w_init))
        b_small = torch.nn.init.normal_(torch.empty(self.out_channels), std=float(w_noise.std()))
        self.bias_fwd = nn.Parameter(b_small) if self.use_bias else None

    def forward(self, input, context=None, compute_expensive=False):
        if self.training:
            self.logabsdet_dirty = True
        self.input = input
        if compute_expensive:
            self.output = _conv(input, self.weight_fwd, self.padding)
            if self.bias_fwd is not None:
                self.output = self.output + self.bias_fwd.view(1, -1, 1, 1)
            ldj = self.logdet(input, context)
        else:
            self.output = selfnorm_conv_2d(input, self.weight_fwd, self.bias_fwd, self.weight_inv, self.stride,
                                           self.padding, self.dilation, self.groups)
            ldj = 0.0
        return self.output, ldj

    def reverse(self, input, context=None, compute_expensive=False):
        if self.bias_fwd is not None:
            input = input - self.bias_fwd.view(1, -1, 1, 1)
        if compute_expensive:
            T = self.dense_operator(input)
            rev = torch.linalg.solve(T, input.flatten(start_dim=1).double().cpu().T).T
            return rev.to(input.dtype).to(input.device).view(input.shape)
        return _conv(input, self.weight_inv, self.padding)

    def add_recon_grad(self, recon_loss_weight_update=None):
        """Reconstruction gradient ||x - R(Wx)||^2 (+ symmetric ||z - W(Rz)||^2), GECO update of the
        weight -- the procedure of selfnorm.py:187-229 with the convolutions and their gradients in HIP."""
        x = self.input.detach()
        z = _conv(x, self.weight_fwd, self.padding)
        if self.only_R_recon:
            z = z.detach()
        x_hat = _conv(z, self.weight_inv, self.padding)
        recon_loss = (x - x_hat).pow(2).flatten(start_dim=1).sum(-1)
        if self.sym_recon_grad:
            zsym = z.detach()
            xsym = _conv(z, self.weight_inv, self.padding)
            z_hat_sym = _conv(xsym, self.weight_fwd, self.padding)
            recon_loss = (recon_loss + (zsym - z_hat_sym).pow(2).flatten(start_dim=1).sum(-1)) / 2.0
        if recon_loss_weight_update is not None:
            self.recon_loss_weight = recon_loss_weight_update
        recon_loss = torch.where(torch.isnan(recon_loss), torch.zeros_like(recon_loss), recon_loss)
        recon_loss_weighted = self.recon_loss_weight * recon_loss.mean()
        recon_loss_weighted.backward()
        if self.recon_loss_lr > 0.0:  # GECO
            with torch.no_grad():
                m = recon_loss.mean()
                self.recon_loss_ema = m if self.recon_loss_ema is None else self.alpha * self.recon_loss_ema + (1 - self.alpha) * m
                C_t = m + (self.recon_loss_ema - m)
                self.recon_loss_weight = self.recon_loss_weight * torch.exp(self.recon_loss_lr * C_t)
        return recon_loss_weighted

    def dense_operator(self, input):
        """Dense matrix of the convolution (what the reference builds with inf/utils/toeplitz.py:9-44),
        obtained by pushing the identity through the HIP conv; fp64 on the CPU like selfnorm.py:175-180.
        Differentiable in weight_fwd while gradients are recorded (the reference's exact-gradient baseline trains
        through slogdet of this matrix, selfnorm.py:240-246 with experiment.py:161)."""
        C, H, W = input.shape[1:]
        n = C * H * W
        eye = torch.eye(n, device=input.device, dtype=torch.float32).view(n, C, H, W)
        if torch.is_grad_enabled() and self.weight_fwd.requires_grad:
            cols = _conv(eye, self.weight_fwd, self.padding)
        else:
            cols = _h.conv2d(eye, self.weight_fwd.detach().contiguous(), None, self.padding)
        return cols.flatten(start_dim=1).T.double().cpu()

    @mark_expensive
    def logdet(self, input, context=None, compute_expensive=True):
        if torch.is_grad_enabled() and self.weight_fwd.requires_grad:
            # part of the loss: a fresh graph per call, never the cached value
            logabsdet = torch.slogdet(self.dense_operator(input))[1].to(input.dtype).to(input.device)
            return logabsdet.view(1).expand(len(input))
        if self.logabsdet_dirty:
            self.logabsdet = torch.slogdet(self.dense_operator(input))[1].to(input.dtype).to(input.device)
            self.logabsdet_dirty = False
        return self.logabsdet.view(1).expand(len(input))


class SelfNormFC(SelfNormConv):
    """Fully-connected variant: a 1x1 SelfNormConv on (B, F, 1, 1) (selfnorm.py:280-334)."""

    def __init__(self, in_features, out_features, bias=True, **kwargs):
        super().__init__(in_features, out_features, (1, 1), bias, **kwargs)

    def reset_parameters(self):
        self.logabsdet_dirty = True
        w_shape = (self.out_channels, self.in_channels, 1, 1)
        sq = min(self.out_channels, self.in_channels)
        w_init = nn.init.xavier_normal_(torch.empty(w_shape), gain=0.01)
        std = float(w_init.std())
        w_init[:sq, :sq, 0, 0] = torch.eye(sq)
        self.weight_fwd = nn.Parameter(w_init)
        self.weight_inv = nn.Parameter(flip_kernel(w_init))
        self.bias_fwd = nn.Parameter(torch.nn.init.normal_(torch.empty(self.out_channels), std=std)) if self.use_bias else None

    def forward(self, input, context=None, compute_expensive=False):
        out, ldj = super().forward(input.reshape(-1, self.in_channels, 1, 1), context, compute_expensive)
        return out.view(-1, self.out_channels), ldj

    def reverse(self, input, context=None, compute_expensive=False):
        rev = super().reverse(input.reshape(-1, self.out_channels, 1, 1), context, compute_expensive)
        return rev.reshape(-1, self.in_channels)

    @mark_expensive
    def logdet(self, input, context=None, compute_expensive=True):
        if self.in_channels != self.out_channels:
            return torch.zeros(len(input), device=input.device)
        if self.logabsdet_dirty:
            self.logabsdet = torch.slogdet(self.weight_fwd[:, :, 0, 0].detach().double().cpu())[1].to(input.dtype).to(input.device)
            self.logabsdet_dirty = False
        return self.logabsdet.view(1).expand(len(input))
