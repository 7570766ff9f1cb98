"""The FInC-flow convolution: a corner-padded masked convolution whose *reverse* is the inverse of convolution
(reference: inf/layers/conv.py:22-222 `PaddedConv2d`, inf/layers/fincflow.py:14-106 `Finc_FlowUnit`).

The forward direction is an ordinary padded convolution (autograd through torch); the reverse is the same triangular
solve as the inverse-flow layer's forward and runs on the HIP library (`ifl_inverse_f32`).  Where the reference flips
the input, the kernel and the result with torch.flip copies around a TL-only solver (conv.py:116-165,192-219), the
library takes the layer's `order` and folds the reflection into its addressing.

State-dict layout as in the reference: the kernel lives in an `nn.Conv2d` child, key `conv.weight` (conv.py:60), stored
pre-flipped for the order (conv.py:74-81), so a reference checkpoint (experiment.py:475-502) loads unchanged."""
import torch
import torch.nn as nn
import torch.nn.functional as F

import invflow_hip as _h

from .flowlayer import FlowLayer

_ORDER_FLIP_DIMS = {"TL": None, "TR": [3], "BL": [2], "BR": [2, 3]}


class PaddedConv2d(FlowLayer):
    """Conv2d padded on two sides only: TL top+left, TR top+right, BL bottom+left, BR bottom+right."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=False, order="TL"):
        super().__init__()
        assert len(kernel_size) == 2
        assert order in _ORDER_FLIP_DIMS, "unknown order: {}".format(order)
        assert in_channels == out_channels, "an invertible convolution needs in_channels == out_channels"
        self.kernel_size = kernel_size
        self.order = order
        kh, kw = kernel_size
        # (left, right, top, bottom) as F.pad takes them (conv.py:41-59)
        self.pad = ((kw - 1) if order in ("TL", "BL") else 0, (kw - 1) if order in ("TR", "BR") else 0,
                    (kh - 1) if order in ("TL", "TR") else 0, (kh - 1) if order in ("BL", "BR") else 0)
        # the reference ignores its `bias` argument (conv.py:60); the child keeps the `conv.weight` key
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=False)
        self.reset_parameters()

    def reset_parameters(self):
        w = self.conv.weight.data
        nn.init.normal_(w, mean=0.0, std=0.05)
        for c_out in range(w.shape[0]):  # unit lower-triangular diagonal tap (conv.py:68-70)
            w[c_out, c_out, -1, -1] = 1.0
            w[c_out, c_out + 1:, -1, -1] = 0.0
        dims = _ORDER_FLIP_DIMS[self.order]
        if dims:
            self.conv.weight.data = torch.flip(w, dims).contiguous()
        self.mask = self.get_mask()

    def get_mask(self):
        mask = torch.ones_like(self.conv.weight.data)
        for c_out in range(mask.shape[0]):
            mask[c_out, c_out:, -1, -1] = 0.0
        dims = _ORDER_FLIP_DIMS[self.order]
        return torch.flip(mask, dims) if dims else mask

    def reset_gradients(self):
        """in place (the reference assigns a new tensor, conv.py:99-100: that would take .grad out of a flat gradient bucket
        and out of a captured graph's addresses); the mask is cached per device"""
        g = self.conv.weight.grad
        if g is not None:
            m = getattr(self, "_grad_mask", None)
            if m is None or m.device != g.device or m.dtype != g.dtype:
                m = self._grad_mask = self.mask.to(device=g.device, dtype=g.dtype)
            g.mul_(m)

    def forward(self, x, context=None, compute_expensive=None):
        return self.conv(F.pad(x, self.pad)), 0.0

    def reverse(self, x, context=None, compute_expensive=None):
        """(y, 0): y with forward(y) == x, as the reference's reverse_cython / reverse_cuda return it."""
        if not x.is_cuda:
            raise RuntimeError("PaddedConv2d.reverse runs on the HIP library: x must be a CUDA tensor")
        with torch.no_grad():
            w = self.conv.weight.detach().to(torch.float32).contiguous()
            y = _h.inverse(x.to(torch.float32).contiguous(), w, self.order)
        return y.to(x.dtype), 0

    def logdet(self, x, context=None):
        return 0.0

    def extra_repr(self):
        return "kernel_size={}, order={}".format(tuple(self.kernel_size), self.order)


class Finc_FlowUnit(nn.Module):
    """Four PaddedConv2d, one per corner, each on a quarter of the channels (fincflow.py:14-50)."""

    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        if isinstance(kernel_size, int) or len(kernel_size) == 1:
            k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
            kernel_size = (k, k)
        assert in_channels % 4 == 0, "Input channels have to be a multiple of 4"
        c = in_channels // 4
        self.conv_tl = PaddedConv2d(c, c, kernel_size, order="TL")
        self.conv_tr = PaddedConv2d(c, c, kernel_size, order="TR")
        self.conv_bl = PaddedConv2d(c, c, kernel_size, order="BL")
        self.conv_br = PaddedConv2d(c, c, kernel_size, order="BR")

    def _convs(self):
        return (self.conv_tl, self.conv_tr, self.conv_bl, self.conv_br)

    def forward(self, x, context=None):
        outs, logdet = [], 0.0
        for conv, part in zip(self._convs(), torch.chunk(x, 4, dim=1)):
            o, ld = conv(part)
            outs.append(o)
            logdet = logdet + ld
        return torch.cat(outs, dim=1), logdet

    def reverse(self, x, context=None):
        # fincflow.py:79-106 concatenates flipped copies for one TL launch; here each quarter is one library call
        # on its own order, nothing is flipped
        outs = [conv.reverse(part.contiguous())[0] for conv, part in zip(self._convs(), torch.chunk(x, 4, dim=1))]
        return torch.cat(outs, dim=1)
