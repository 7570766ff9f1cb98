"""The full "inverse flow" block: TL -> TR -> BL -> BR chain of inverse-conv layers
(reference: inf/layers/inv_flow.py:13-53)."""
import torch.nn as nn

from .inv_conv import inv_flow_with_pad


class Inv_FlowUnit(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size):
        super().__init__()
        if isinstance(kernel_size, int) or len(kernel_size) == 1:
            k = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
            kernel_size = (k, k)
        self.conv_tl = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="TL")
        self.conv_tr = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="TR")
        self.conv_bl = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="BL")
        self.conv_br = inv_flow_with_pad(out_channels, out_channels, kernel_size, order="BR")

    def _chain(self):
        return (self.conv_tl, self.conv_tr, self.conv_bl, self.conv_br)

    def forward(self, x, context=None):
        logdet = 0.0
        for layer in self._chain():
            x, ld = layer(x, context)
            logdet = logdet + ld
        return x, logdet

    def reverse(self, x, context=None):
        for layer in reversed(self._chain()):
            x = layer.reverse(x, context)
        return x

    def logdet(self, input, context=None):
        return 0.0
