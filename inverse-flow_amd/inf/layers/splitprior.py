"""SplitPrior: half of the channels leave the flow through a coupling-conditioned Gaussian (reference surface:
inf/layers/splitprior.py:7-40).  A caller of the path (the multi-scale split of BASELINE configs 3-5); its coupling is the
package's Coupling (the affine part on the HIP library)."""
import torch

from .coupling import Coupling
from .flowlayer import FlowLayer


class SplitPrior(FlowLayer):
    def __init__(self, input_size, distribution, width=512):
        super().__init__()
        assert len(input_size) == 3
        self.n_channels = input_size[0]
        self.transform = Coupling(input_size, width=width)
        self.base = distribution((self.n_channels // 2, input_size[1], input_size[2]))

    def forward(self, input, context=None):
        x, ldj = self.transform(input, context)
        x1 = x[:, :self.n_channels // 2, :, :]
        x2 = x[:, self.n_channels // 2:, :, :]
        return x1.contiguous(), self.base.log_prob(x2) + ldj

    def reverse(self, input, context=None):
        x2, _ = self.base.sample(input.shape[0], context)
        return self.transform.reverse(torch.cat([input, x2], dim=1), context)

    def logdet(self, input, context=None):
        return self.forward(input, context)[1]
