"""Container that chains flow layers and accumulates log|det J| (reference:
inf/layers/flowsequential.py:8-141).  API-compatible; two reference defects are not reproduced
(SURVEY 2.3-9): each layer's log-det is added once (the reference adds it twice,
flowsequential.py:30-35) and nothing is printed per layer (flowsequential.py:36)."""
import torch
import torch.nn as nn

from .activations import FlowActivationLayer
from .flowlayer import ModifiedGradFlowLayer, PreprocessingFlowLayer
from .selfnorm import SelfNormConv


def _call(module, input, context, compute_expensive):
    if isinstance(module, ModifiedGradFlowLayer):
        return module(input, context, compute_expensive=compute_expensive)
    return module(input, context)


def _reverse(module, input, context, compute_expensive):
    if isinstance(module, ModifiedGradFlowLayer):
        return module.reverse(input, context, compute_expensive)
    return module.reverse(input, context)


class FlowSequential(nn.Module):
    def __init__(self, base_distribution, *modules):
        super().__init__()
        self.base_distribution = base_distribution
        for i, module in enumerate(modules):
            self.add_module(str(i), module)
        self.sequence_modules = modules

    def __iter__(self):
        yield from self.sequence_modules

    def forward(self, input, context=None, compute_expensive=False):
        logdet = 0.0
        output = input
        for module in self:
            output, layer_logdet = _call(module, output, context, compute_expensive)
            if isinstance(layer_logdet, torch.Tensor) and isinstance(logdet, torch.Tensor):
                logdet = logdet.to(layer_logdet.device) + layer_logdet
            else:
                logdet = logdet + layer_logdet
        logprob = self.base_distribution.log_prob(output)
        return output, logprob + logdet

    def log_prob(self, input, context=None, compute_expensive=True):
        return self.forward(input, context, compute_expensive)[1]

    def cheap_unnormed_log_prob(self, input, context=None):
        return self.log_prob(input, context=context, compute_expensive=False)

    def _of_type(self, cls, negate=False):
        for module in self.sequence_modules:
            if isinstance(module, cls) != negate:
                yield module

    def activation_modules(self):
        return self._of_type(FlowActivationLayer)

    def selfnorm_modules(self):
        return self._of_type(SelfNormConv)

    def preprocessing_modules(self):
        return self._of_type(PreprocessingFlowLayer)

    def non_preprocessing_modules(self):
        return self._of_type(PreprocessingFlowLayer, negate=True)

    def non_preprocessing_logdet(self, input, context=None, *, compute_expensive=False):
        logdet = 0.0
        for module in self.non_preprocessing_modules():
            input, layer_logdet = _call(module, input, context, compute_expensive)
            logdet = logdet + layer_logdet
        return self.base_distribution.log_prob(input) + logdet

    def add_recon_grad(self, recon_loss_weight_update=None):
        total = 0.0
        for conv in self.selfnorm_modules():
            total = total + conv.add_recon_grad(recon_loss_weight_update)
        return total

    def sample(self, n_samples, context=None, compute_expensive=False, also_true_inverse=False):
        z, _ = self.base_distribution.sample(n_samples, context)
        x = z
        for module in reversed(self.sequence_modules):
            x = _reverse(module, x, context, compute_expensive)
        x_true = x
        if not compute_expensive and also_true_inverse:
            x_true = z
            for module in reversed(self.sequence_modules):
                x_true = _reverse(module, x_true, context, True)
        return x, x_true

    def reconstruct(self, x, context=None, compute_expensive=False):
        h = x
        for module in self.sequence_modules:
            h, _ = _call(module, h, context, compute_expensive)
        for module in reversed(self.sequence_modules):
            h = _reverse(module, h, context, compute_expensive)
        return h
