"""Container that chains flow layers and accumulates log|det J| (reference surface: inf/layers/flowsequential.py:8-141).

API-compatible with the reference container.  Two of its defects are not reproduced: a layer's log-det
is added once (the reference adds it twice, flowsequential.py:30-35) and nothing is printed per layer
(flowsequential.py:36)."""
import torch
from torch import nn

from .activations import FlowActivationLayer
from .flowlayer import ModifiedGradFlowLayer, PreprocessingFlowLayer
from .selfnorm import SelfNormConv


def _apply(module, method, tensor, context, compute_expensive):
    """module.<method>(tensor, context[, compute_expensive]) -- the switch only for layers that know it."""
    fn = module if method == "forward" else getattr(module, method)
    if isinstance(module, ModifiedGradFlowLayer):
        return fn(tensor, context, compute_expensive=compute_expensive)
    return fn(tensor, context)


def _accumulate(total, term):
    if torch.is_tensor(total) and torch.is_tensor(term):
        return total.to(term.device) + term
    return total + term


def _sum_terms(terms):
    """the layers' log-dets added up: python numbers in python, tensors of one shape on one device by ONE stack + sum (a
    model of a hundred layers otherwise spends a hundred launches of a B-element add on it), anything else one by one"""
    total, same = 0.0, []
    for t in terms:
        if torch.is_tensor(t) and t.is_cuda and t.dim() == 1 and (not same or (t.shape == same[0].shape and t.device == same[0].device
                                                                             and t.dtype == same[0].dtype)):
            same.append(t)
        else:
            total = _accumulate(total, t)
    if len(same) > 2:
        return _accumulate(total, torch.stack(same).sum(0))
    for t in same:
        total = _accumulate(total, t)
    return total


class FlowSequential(nn.Module):
    def __init__(self, base_distribution, *modules):
        super().__init__()
        self.base_distribution = base_distribution
        self.sequence_modules = modules
        for index, module in enumerate(modules):
            self.add_module(str(index), module)

    def __iter__(self):
        return iter(self.sequence_modules)

    # ---- the two directions --------------------------------------------------------------------------------------
    def _push(self, modules, tensor, context, compute_expensive):
        """forward through `modules`: (output, summed log-det)"""
        terms = []
        for module in modules:
            tensor, term = _apply(module, "forward", tensor, context, compute_expensive)
            terms.append(term)
        return tensor, _sum_terms(terms)

    def _pull(self, tensor, context, compute_expensive):
        """reverse through all modules, last first"""
        for module in self.sequence_modules[::-1]:
            tensor = _apply(module, "reverse", tensor, context, compute_expensive)
        return tensor

    def forward(self, input, context=None, compute_expensive=False):
        output, logdet = self._push(self.sequence_modules, input, context, compute_expensive)
        return output, self.base_distribution.log_prob(output) + logdet

    def log_prob(self, input, context=None, compute_expensive=True):
        return self.forward(input, context, compute_expensive)[1]

    def cheap_unnormed_log_prob(self, input, context=None):
        return self.log_prob(input, context=context, compute_expensive=False)

    def non_preprocessing_logdet(self, input, context=None, *, compute_expensive=False):
        output, logdet = self._push(list(self.non_preprocessing_modules()), input, context, compute_expensive)
        return self.base_distribution.log_prob(output) + logdet

    def sample(self, n_samples, context=None, compute_expensive=False, also_true_inverse=False):
        z, _ = self.base_distribution.sample(n_samples, context)
        x = self._pull(z, context, compute_expensive)
        x_true = self._pull(z, context, True) if (also_true_inverse and not compute_expensive) else x
        return x, x_true

    def reconstruct(self, x, context=None, compute_expensive=False):
        latent, _ = self._push(self.sequence_modules, x, context, compute_expensive)
        return self._pull(latent, context, compute_expensive)

    # ---- module selections ----------------------------------------------------------------------------------------
    def _select(self, kind, keep=True):
        return (m for m in self.sequence_modules if isinstance(m, kind) == keep)

    def activation_modules(self):
        return self._select(FlowActivationLayer)

    def selfnorm_modules(self):
        return self._select(SelfNormConv)

    def preprocessing_modules(self):
        return self._select(PreprocessingFlowLayer)

    def non_preprocessing_modules(self):
        return self._select(PreprocessingFlowLayer, keep=False)

    def add_recon_grad(self, recon_loss_weight_update=None):
        return sum((conv.add_recon_grad(recon_loss_weight_update) for conv in self.selfnorm_modules()), 0.0)
