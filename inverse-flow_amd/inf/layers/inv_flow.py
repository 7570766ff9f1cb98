"""The full "inverse flow" block: TL -> TR -> BL -> BR chain of inverse-conv layers
(reference: inf/layers/inv_flow.py:13-53).

The forward runs as ONE library call (ifl_unit_inverse_f32: a single fold launch for the four layers and their
adjoints, then the four scans back to back), the backward as one call as well (ifl_unit_backward_f32); `reverse`
walks the layers like the reference."""
import torch
import torch.nn as nn

import invflow_hip as _h

from .inv_conv import inv_flow_with_pad


# Inside a torch.autocast region (the reference's bf16 training configs) the arithmetic of these layers stays fp32:
# tensor arguments are cast to float32 on the way in, gradients come back in float32.
_fwd32 = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
_bwd32 = torch.amp.custom_bwd(device_type="cuda")


class _unit_fn(torch.autograd.Function):
    """z_BR = A_BR^-1 A_BL^-1 A_TR^-1 A_TL^-1 x with the true gradients of x and of the four kernels."""

    @staticmethod
    @_fwd32
    def forward(ctx, x, w_tl, w_tr, w_bl, w_br, flags=0):
        x = x.contiguous()
        ws4 = [w.contiguous() for w in (w_tl, w_tr, w_bl, w_br)]
        need_bwd = any(ctx.needs_input_grad[:5])
        carries = [_h.new_carry(w) for w in ws4] if need_bwd else None
        zs = _h.unit_inverse(x, ws4, flags, carries)
        ctx.flags, ctx.carries = flags, carries
        ctx.versions = [w._version for w in ws4]
        ctx.save_for_backward(*ws4, *zs)
        return zs[3]

    @staticmethod
    @_bwd32
    def backward(ctx, output_grad):
        saved = ctx.saved_tensors
        ws4, zs = list(saved[:4]), list(saved[4:])
        fresh = all(w._version == v for w, v in zip(ws4, ctx.versions))
        dx, dws = _h.unit_backward(output_grad.contiguous(), zs, ws4, ctx.flags, ctx.carries if fresh else None)
        return (dx, *dws, None)


class Inv_FlowUnit(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, reference_init=False):
        super().__init__()
        if isinstance(kernel_size, int) or len(kernel_size) == 1:
            k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
            kernel_size = (k, k)
        kw = dict(reference_init=reference_init)  # (inf/layers/inv_conv.py, _init_weight)
        self.conv_tl = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="TL", **kw)
        self.conv_tr = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="TR", **kw)
        self.conv_bl = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="BL", **kw)
        self.conv_br = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="BR", **kw)

    def _chain(self):
        return (self.conv_tl, self.conv_tr, self.conv_bl, self.conv_br)

    def forward(self, x, context=None):
        layers = self._chain()
        flags = layers[0].flags
        # one library call for the block when the four layers agree on their flags and have the unit diagonal
        # (log-det exactly 0, inv_conv.py:221); otherwise layer by layer like the reference
        if all(l.flags == flags for l in layers) and not (flags & _h.FLAG_GENERAL_DIAG) and x.is_cuda:
            out = _unit_fn.apply(x, *(l.weight_fwd for l in layers), flags)
            return out, 0.0
        logdet = 0.0
        for layer in layers:
            x, ld = layer(x, context)
            logdet = logdet + ld
        return x, logdet

    def reverse(self, x, context=None):
        for layer in reversed(self._chain()):
            x = layer.reverse(x, context)
        return x

    def logdet(self, input, context=None):
        return 0.0
