"""Affine coupling on the HIP library (reference surface: inf/layers/coupling.py:9-102).

y = cat(x1, x2 * exp(log_s) + t) with (h_s, t) the even / odd channels of h = net(x1) and log_s = 2 tanh(h_s / 2);
log-det = sum log_s; reverse: x2 = (y2 - t) exp(-log_s).  The conditioner `net` -- 3x3 conv, ReLU, 1x1 conv, ReLU,
zero-initialised 3x3 conv with a learned per-channel log-scale (Conv2dZero) -- keeps the reference's module tree and
parameter names (a reference state_dict loads as is) and stays on library convolutions; everything that touches the
activation elementwise (the strided split of h, tanh / exp, the affine map, the concatenation, the per-image sum) is
one pass of libinvflow_hip each way (ifl_coupling_f32 / ifl_coupling_backward_f32).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import invflow_hip as H

from .flowlayer import FlowLayer

_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")
LOGS_RANGE = 2.0  # coupling.py:79


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2dZero(nn.Module):
    """A convolution that starts at zero with a per-channel output gain exp(logscale_factor * logs) (coupling.py:9-45).
    As in the reference, `bias` and `logs` are two Parameters over ONE zero tensor (they share storage)."""

    def __init__(self, in_channels, out_channels, bias=True, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                 dilation=1, groups=1, logscale_factor=3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.logscale_factor = groups, logscale_factor
        self.weight = nn.Parameter(torch.zeros(out_channels, in_channels // groups, *self.kernel_size))
        shared = torch.zeros(out_channels)
        self.bias = nn.Parameter(shared) if bias else None
        self.logs = nn.Parameter(shared)

    def forward(self, input):
        gain = torch.exp(self.logs * self.logscale_factor)
        out = F.conv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return out * gain[None, :, None, None]


class _Affine(torch.autograd.Function):
    """(x, h) -> (y, sum log_s).  The input gradient returned here is the direct part; the part through h = net(x1) is
    autograd's own business."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, h):
        # (outside autocast the activation's storage format decides: bf16 activations take the bf16 entry points)
        x, h = x.contiguous(), h.to(x.dtype).contiguous()
        ctx.save_for_backward(x, h)
        return H.coupling(x, h)

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, h = ctx.saved_tensors
        return H.coupling_backward(gy.contiguous(), gld.contiguous() if gld is not None else None, x, h)


class _CondAffine(torch.autograd.Function):
    """(x, the conditioner's parameters) -> (y, sum log_s): conditioner and affine map on the library, forward and backward
    (csrc/conditioner.hip: two launches for h = net(x1), six for its backward, next to the affine map's one each way).
    fp32 arithmetic throughout -- under autocast too: the net's arithmetic is small, its launch count is not; `lowp`
    (autocast on) only picks bf16 operand matrices for the three weight-gradient products."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, w1, w2, w3, b3, logs, logscale, lowp, wt=None):
        x = x.contiguous()
        w1, w2 = w1.contiguous(), w2.contiguous()
        if wt is None:  # (else: prepared for all couplings of the model in one launch, prepare_conditioners)
            wt = H.cond_prep(w1, w2, w3.contiguous(), logs.contiguous(), logscale)
        a2, h = H.cond_forward(x, wt, w1, w2, b3, w3.shape[0], w1.shape[0])
        ctx.save_for_backward(x, h, a2, wt, w1)
        ctx.width = w1.shape[0]
        ctx.logscale, ctx.has_bias, ctx.lowp = logscale, b3 is not None, lowp
        return H.coupling(x, h)

    @staticmethod
    @_bwd32
    def backward(ctx, gy, gld):
        x, h, a2, wt, w1 = ctx.saved_tensors
        gx, gh = H.coupling_backward(gy.contiguous(), gld.contiguous() if gld is not None else None, x, h)
        dw1, dw2, dw3, dlogs, db3 = H.cond_backward(x, gh, h, a2, wt, w1, gx, ctx.width, ctx.logscale, ctx.lowp)
        return gx, dw1, dw2, dw3, (db3 if ctx.has_bias else None), dlogs, None, None, None


class ConditionerPrep:
    """The weight images (transposed kernels + gain, csrc/conditioner.hip) of all fused couplings of a model in ONE launch.
    A training step calls `fill()` before its forward pass and `release()` after its backward: between the two every
    coupling takes its image from here instead of preparing it inside its own forward (a launch per layer: 30 a step for
    the configs[3] model, 144 for configs[4]).  Outside that window -- and for couplings the fused path does not cover --
    nothing changes.  The job table is rebuilt when a parameter has moved (e.g. re-homed into a flat optimizer buffer).
    Only for models whose images together stay below `max_bytes` (8 MB: they then wait in the L2s for their layers): the
    images of the configs[4] model (33 MB) prepared up front come back from memory, and its step got slower (30.0 ->
    30.2 ms), where the configs[3] step gains 1 % (5.41 -> 5.35 ms)."""

    def __init__(self, model, max_bytes=8 << 20):
        self.layers = [m for m in model.modules() if isinstance(m, Coupling)]
        self.table, self._active, self.max_bytes = None, [], max_bytes

    def _entries(self):
        act = []
        for m in self.layers:
            c1, c2, c3 = m.net[0], m.net[2], m.net[4]
            ok = (m.fused and not m.uses_context and m.n_channels % 2 == 0 and c1.weight.is_cuda and c1.weight.dtype == torch.float32
                  and H.cond_supported(m.n_channels, m.width) and isinstance(c3, Conv2dZero) and c3.kernel_size == (3, 3)
                  and all(t.is_contiguous() for t in (c1.weight, c2.weight, c3.weight, c3.logs)))
            if ok:
                act.append((m, (c1.weight.detach(), c2.weight.detach(), c3.weight.detach(), c3.logs.detach(), c3.logscale_factor)))
        return act

    def fill(self):
        act = self._entries()
        ent = [e for _, e in act]
        if self.table is None or self.table.stale(ent):
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                return  # (no table yet and a capture running: the couplings prepare their images themselves)
            self.table = H.CondPrepTable(ent) if ent else None
        if self.table is None or 4 * sum(w.numel() for w in self.table.wt) > self.max_bytes:
            return
        self.table.run()
        self._active = [m for m, _ in act]
        for m, wt in zip(self._active, self.table.wt):
            m._prepared_wt = wt

    def release(self):
        for m in self._active:
            m._prepared_wt = None
        self._active = []


def _on_library(x):
    return x.dim() == 4 and x.is_cuda and x.size(1) % 2 == 0 and x.dtype in (torch.float32, torch.float16, torch.bfloat16)


class Coupling(FlowLayer):
    channels_last = True  # the conditioner's 16-bit convolutions on NHWC operands (_net_channels_last)
    _prepared_wt = None  # the weight image of THIS step, when a ConditionerPrep filled it (else the forward prepares it)
    fused = True  # conditioner + affine map on csrc/conditioner.hip where its shapes are covered (_CondAffine)

    def __init__(self, input_size, width=512, n_context=None):
        super().__init__()
        channels = input_size[0]
        self.n_channels, self.half_channels, self.width = channels, channels // 2, width
        self.uses_context = n_context is not None
        conditioner_in = self.half_channels + (n_context if self.uses_context else 0)
        self.net = nn.Sequential(
            nn.Conv2d(conditioner_in, width, kernel_size=(3, 3), padding=(1, 1), bias=False), nn.ReLU(),
            nn.Conv2d(width, channels, (1, 1), bias=False), nn.ReLU(),
            Conv2dZero(channels, channels))

    def _fusable(self, x, context):
        c1, c2, c3 = self.net[0], self.net[2], self.net[4]
        return (self.fused and context is None and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32
                and self.n_channels % 2 == 0 and H.cond_supported(self.n_channels, self.width)
                and c3.stride == (1, 1) and c3.padding == (1, 1) and c3.dilation == (1, 1) and c3.groups == 1
                and c3.kernel_size == (3, 3) and c1.weight.dtype == torch.float32)

    def _fused_h(self, x):
        c1, c2, c3 = self.net[0], self.net[2], self.net[4]
        w1, w2 = c1.weight.detach().contiguous(), c2.weight.detach().contiguous()
        wt = H.cond_prep(w1, w2, c3.weight.detach().contiguous(), c3.logs.detach().contiguous(), c3.logscale_factor)
        return H.cond_forward(x, wt, w1, w2, None if c3.bias is None else c3.bias.detach(), self.n_channels, self.width)[1]

    def _conditioner(self, x, context):
        assert (context is not None) == self.uses_context
        x1 = x[:, :self.half_channels]
        if context is not None:
            x1 = torch.cat([x1, context], dim=1)
        if self.channels_last and x1.is_cuda and x1.dim() == 4 and torch.is_autocast_enabled("cuda"):
            return self._net_channels_last(x1, torch.get_autocast_dtype("cuda"))
        return self.net(x1)

    def _net_channels_last(self, x1, dtype):
        """`self.net` under autocast with the activations and the 16-bit copies of the kernels in channels-last memory.
        MIOpen's 16-bit convolutions are NHWC kernels: handed NCHW tensors it wraps every one of them in transposes and
        casts (20 launches for a forward + backward of the 3x3, 11 with NHWC operands, and 30 % less device time --
        tools/conv_layout_probe.py), and the training step of these models is bound by its launch count.  Same modules, same
        parameters, same casts as autocast makes (one per operand), only the memory format differs."""
        cl = torch.channels_last
        c1, c2, c3 = self.net[0], self.net[2], self.net[4]
        h = F.conv2d(x1.to(dtype, memory_format=cl), c1.weight.to(dtype, memory_format=cl), None, c1.stride, c1.padding)
        h = F.conv2d(F.relu(h), c2.weight.to(dtype, memory_format=cl), None, c2.stride, c2.padding)
        h = F.conv2d(F.relu(h), c3.weight.to(dtype, memory_format=cl), None if c3.bias is None else c3.bias.to(dtype),
                     c3.stride, c3.padding, c3.dilation, c3.groups)
        return h * torch.exp(c3.logs * c3.logscale_factor)[None, :, None, None]

    def get_xs_logs_t(self, x, context=None):
        """(x1, x2, log_s, t) as torch tensors (the library path never materialises them)."""
        h = self._conditioner(x, context)
        x1, x2 = x.split([self.half_channels, self.n_channels - self.half_channels], dim=1)
        return x1, x2, LOGS_RANGE * torch.tanh(h[:, 0::2] / LOGS_RANGE), h[:, 1::2]

    def forward(self, input, context=None):
        if self._fusable(input, context):
            c1, c2, c3 = self.net[0], self.net[2], self.net[4]
            # (under bf16 autocast the weight gradients' GEMMs take bf16 operands -- the step's precision; fp32 otherwise)
            return _CondAffine.apply(input, c1.weight, c2.weight, c3.weight, c3.bias, c3.logs, c3.logscale_factor,
                                     torch.is_autocast_enabled("cuda"), self._prepared_wt)
        if _on_library(input):
            return _Affine.apply(input, self._conditioner(input, context))
        x1, x2, log_s, t = self.get_xs_logs_t(input, context)
        return torch.cat([x1, torch.addcmul(t, x2, log_s.exp())], dim=1), log_s.sum(dim=(1, 2, 3))

    def reverse(self, input, context=None):
        if not torch.is_grad_enabled() and self._fusable(input, context):
            x = input.contiguous()
            return H.coupling(x, self._fused_h(x), reverse=True)
        if _on_library(input) and input.dtype in (torch.float32, torch.bfloat16) and not torch.is_grad_enabled():
            return H.coupling(input.contiguous(), self._conditioner(input, context).to(input.dtype).contiguous(), reverse=True)
        x1, y2, log_s, t = self.get_xs_logs_t(input, context)
        return torch.cat([x1, (y2 - t) * torch.exp(-log_s)], dim=1)

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
